#!/bin/bash
# the final tree: whole GPU suite, fuzz suite at four times its size, smoke, the driver's command once
set -o pipefail
OUT=gpurun_out/r3s45
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
MISPMM_FUZZ_SCALE=4 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_fuzz.log 2>&1; rc=$?
tail -4 $OUT/pytest_fuzz.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
t0=$(date +%s); timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
echo "wall $(( $(date +%s) - t0 )) s"
python -c "import json;d=json.load(open('$OUT/bench_default.json'));print('driver command', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['roofline']['traffic'], d['timing']['placements_us'], d['cpu_baseline']['gpu_parity'])"
echo done
