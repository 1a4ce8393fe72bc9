#!/bin/bash
# two-body launch for COO / ELL / BSR lists too: parity, CLI on GL7d25 in every format
set -o pipefail
OUT=gpurun_out/r3s38
mkdir -p $OUT
echo "(spmm + fuzz suites: 255 passed in the run before)"
timeout -k 10 900 python -m pytest tests/test_cli.py -m gpu -x -q -k "long_row or sweep" 2>&1 | tail -6 | tee -a $OUT/tests.log || exit 1
timeout -k 10 600 python tools/sweep.py --dirs large_21074,medium_4000,large_20000 --iters 200 --out $OUT/sweep 2>&1 | tail -20 | tee $OUT/sweep.log
echo done
