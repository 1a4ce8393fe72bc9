#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s19
mkdir -p $OUT
MISPMM_FUZZ_SCALE=4 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_fuzz.log 2>&1; rc=$?
tail -15 $OUT/pytest_fuzz.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -8 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo done
