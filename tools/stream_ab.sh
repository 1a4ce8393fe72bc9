#!/bin/bash
# A/B of the persistent row-walking launch (row_stream.hpp) against the one-row-per-lane-group launch (row_gather.hpp): the same
# bench.py configuration from the tuning build of the library with MISPMM_STREAM=0 / 1, resident and HBM-streamed loops.
#   bash tools/stream_ab.sh <outdir> [extra bench args]
OUT=$1; shift
mkdir -p "$OUT"
export MISPMM_LIB=cuda-optimization-for-spmm_amd/libmispmm_tune.so
for spec in "k128:--config headline" "k256:--config headline --k-cols 256" "cfg3:--config 3" "cfg5:--config 5"; do
  tag=${spec%%:*}; args=${spec#*:}
  for s in 0 1; do
    MISPMM_STREAM=$s timeout -k 10 300 python3 bench.py $args --steps 20 --warmup 5 --no-extras --hbm-streaming on --cpu-seconds 1 --no-live-traffic "$@" \
        > "$OUT/stream_ab_${tag}_$s.json" 2> "$OUT/stream_ab_${tag}_$s.err" || { tail -5 "$OUT/stream_ab_${tag}_$s.err"; exit 1; }
    python3 - "$OUT/stream_ab_${tag}_$s.json" "$tag stream=$s" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{sys.argv[2]:16s} resident {d['ms_per_step'] * 1e3:7.3f} us ({d['roofline']['frac']:.3f})  streamed {d['hbm_streaming']['launch_us']:7.3f} us ({d['hbm_streaming']['frac']:.3f})  "
      f"{d['cpu_baseline']['gpu_parity']}  {d['config']['kernel_tag']}")
PY
  done
done
