#!/bin/bash
# the final tree once more (the split body gained diagnostic stamps under #ifdef only): GPU suite, smoke; kernel-trace digests of every configuration
set -o pipefail
OUT=gpurun_out/r3s49
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
bash tools/profile_r3_configs.sh > $OUT/profile_configs.log 2>&1; tail -30 $OUT/profile_configs.log
echo done
