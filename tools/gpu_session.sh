#!/bin/bash
# One GPU session = one `gpurun` call: named steps run in order, everything they print or write goes under
# gpurun_out/<name>/ (scratch; copy what is to be kept into profiles/rN/).  Replaces the per-session scripts of round 3
# (tools/r3/session*.sh, in the history).
#
#   gpurun --timeout 1100 -- 'bash tools/gpu_session.sh <name> <step> [<step> ...]'
#
# steps (a failing step ends the session: no GPU step is started behind a failed or killed one)
#   parity            pytest -m "gpu and not perf" -x       (the curated suite, fuzz, sharded drivers, CLI, tools)
#   perf              pytest -m "gpu and perf"              (every assertion on a time; tests/test_zz_perf_gpu.py)
#   tests             pytest -m gpu -x                      (what the driver runs: parity first, perf last)
#   smoke             python __graft_entry__.py smoke
#   bench             bench.py at the driver's flags        -> bench_default.json
#   configs           bench.py --config 2 3 4 5             -> bench_cfg<N>.json
#   general           the general CSR entry point (MISPMM_NO_HINT=1) -> bench_general_entry.json
#   trace             rocprofv3 --kernel-trace --stats of the driver's command, digest of the timed kernel
#   pmc               the PMC passes of every configuration + traffic.json (tools/profile_r4.sh)
#   dist              bench.py --gpus: 1 rank over RCCL (torchrun form and plain), 2 ranks sharing the card (gloo + IPC)
#   stamps            per-wave stamps of the headline kernel (libmispmm_stamps.so)
#   cmd:<shell>       any other command, e.g. 'cmd:python tools/probe/lib_ab_probe.py a=... b=...'
set -o pipefail
NAME=$1; shift
OUT=gpurun_out/$NAME
mkdir -p "$OUT"
P=cuda-optimization-for-spmm_amd
export TMPDIR=/tmp
show() { python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
hs = d.get("hbm_streaming", {})
print(sys.argv[2], "us/step", round(d["ms_per_step"] * 1e3, 4), "frac", d["roofline"]["frac"], "placements", d.get("timing", {}).get("placements_us"),
      "| streamed", hs.get("launch_us"), hs.get("frac"), "| traffic", d["roofline"].get("traffic"), d["roofline"].get("traffic_source"),
      "|", d["config"].get("kernel_tag"))
PY
}
for step in "$@"; do
  echo "== $step"
  case "$step" in
    parity) timeout -k 10 1000 python3 -m pytest tests -m "gpu and not perf" -x -q 2>&1 | tail -15 | tee "$OUT/pytest_parity.log" || exit 1;;
    perf)   timeout -k 10 900 python3 -m pytest tests -m "gpu and perf" -q 2>&1 | tail -25 | tee "$OUT/pytest_perf.log" || exit 1;;
    tests)  timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log" || exit 1;;
    smoke)  timeout -k 10 300 python3 __graft_entry__.py smoke 2>&1 | tail -3 | tee "$OUT/smoke.log" || exit 1;;
    bench)  timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" || { tail -20 "$OUT/bench_default.err"; exit 1; }
            show "$OUT/bench_default.json" headline;;
    configs) for c in 2 3 4 5; do
              timeout -k 10 600 python3 bench.py --config $c --steps 20 --warmup 5 > "$OUT/bench_cfg$c.json" 2> "$OUT/bench_cfg$c.err" || { tail -20 "$OUT/bench_cfg$c.err"; exit 1; }
              show "$OUT/bench_cfg$c.json" "config $c"; done;;
    general) MISPMM_NO_HINT=1 timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_general_entry.json" 2> "$OUT/bench_general.err" || exit 1
             show "$OUT/bench_general_entry.json" "general entry";;
    dist)   # the one-process-per-GPU driver on the one card: 1 rank over RCCL (torchrun form, as the driver launches it) and 2 ranks sharing the card
            MISPMM_FORCE_DIST=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 \
                bench.py --gpus 1 --steps 64 --warmup 8 --config 5 > "$OUT/bench_torchrun_1rank_rccl_k512.json" 2> "$OUT/dist1.err" || { tail -20 "$OUT/dist1.err"; exit 1; }
            MISPMM_FORCE_DIST=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 500 --warmup 20 --bucket 500 > "$OUT/bench_dist_world1_k128.json" 2> "$OUT/dist2.err" || { tail -20 "$OUT/dist2.err"; exit 1; }
            MISPMM_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 2 --steps 64 --warmup 8 --exchange both > "$OUT/bench_dist_2ranks_one_card.json" 2> "$OUT/dist3.err" || { tail -20 "$OUT/dist3.err"; exit 1; }
            python3 - "$OUT" <<'PY'
import json, sys
for f in ("bench_torchrun_1rank_rccl_k512", "bench_dist_world1_k128", "bench_dist_2ranks_one_card"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json"))
    print(f, "n_gpus", d["n_gpus"], "us/step", round(d["ms_per_step"] * 1e3, 3), "kernel_only", round(d["kernel_only"]["ms_per_step"] * 1e3, 3),
          "batched", d.get("kernel_only_batched", {}).get("ms_per_step"), {m: r.get("ms_per_step", r) for m, r in d["exchange_modes"].items()}, d["cpu_baseline"]["gpu_parity"])
PY
            ;;
    trace)  timeout -k 10 600 bash tools/profile_r4.sh trace "$OUT" || exit 1;;
    pmc)    timeout -k 10 1100 bash tools/profile_r4.sh pmc "$OUT" || exit 1;;
    stamps) MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python3 tools/stamp_headline.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/stamps_headline.log" || exit 1;;
    cmd:*)  timeout -k 10 900 bash -c "${step#cmd:}" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/cmd.log" || exit 1;;
    *) echo "unknown step $step"; exit 2;;
  esac
done
echo "session $NAME done"
