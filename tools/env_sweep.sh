#!/bin/bash
# Same workload, several environment-selected launch variants, one process each (the knobs are read once per
# process).  tools/env_sweep.sh <k-cols> "<VAR=val ...>" "<VAR=val ...>" ...   ("-" = defaults)
K=$1; shift
# the knobs exist only in the tuning build of the library (make -C cuda-optimization-for-spmm_amd tune)
export MISPMM_LIB=${MISPMM_LIB:-$(dirname "$0")/../cuda-optimization-for-spmm_amd/libmispmm_tune.so}
for V in "$@"; do
  echo "== K=$K variant: $V"
  if [ "$V" = "-" ]; then V=""; fi
  env $V timeout -k 10 120 python3 tools/kernel_sweep.py --kernels 5 --k-cols "$K" --iters 300 --rounds 5 2>&1 | grep -v "^$" || exit 1
done
