#!/bin/bash
# the final tree: whole GPU suite, smoke
set -o pipefail
OUT=gpurun_out/r3s55
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
echo done
