#!/bin/bash
# span lists for short-row matrices with a few long rows (tols4000): parity, then the 12-directory CLI sweep
set -o pipefail
OUT=gpurun_out/r3s39
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/tests.log || exit 1
timeout -k 10 900 python -m pytest tests/test_cli.py tests/test_tools.py -m gpu -x -q 2>&1 | tail -6 | tee -a $OUT/tests.log || exit 1
timeout -k 10 900 python tools/sweep.py --iters 200 --out $OUT/sweep 2>&1 | tail -60 | tee $OUT/sweep.log
echo done
