#!/bin/bash
# the N = 256 grid is now a footprint rule (K x 256 B > 4 MiB -> 1 x 8, else 2 x 4): parity tests, then the new default against the forced grids
set -o pipefail
OUT=gpurun_out/r3s28
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests.log || exit 1
for m in g7jac010 delaunay_n12 ch7-6-b5 n4c6-b13; do
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix $m --k-cols 256 "default=$P/libmispmm_tune.so" "grid-1x8=$P/libmispmm_tune.so:MISPMM_CSR_TILING=1,8" "grid-2x4=$P/libmispmm_tune.so:MISPMM_CSR_TILING=2,4" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tiling_rule.log
done
for c in headline 3 5; do
timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-live-traffic > $OUT/bench_$c.json 2>/dev/null && python - <<PY
import json; r=json.loads(open("$OUT/bench_$c.json").read().strip().splitlines()[-1]); print("cfg $c", r["roofline"]["kernel_us"] if "kernel_us" in r["roofline"] else r["ms_per_step"]*1e3, r["roofline"]["frac"])
PY
done
echo done
