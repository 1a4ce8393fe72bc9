#!/bin/bash
# tiling re-sweep at K = 256 / 512 with the round-2 kernels (tuning build of the library)
O=gpurun_out/r2i; mkdir -p $O
export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_tune.so
for k in 128 256 512; do
  for t in "-" "8,1" "4,2" "2,4" "1,8"; do
    if [ "$t" = "-" ]; then unset MISPMM_CSR_TILING; else export MISPMM_CSR_TILING=$t; fi
    python3 bench.py --k-cols $k --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$k tiling=$t', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'])
" | tee -a $O/tiling_sweep.log
  done
done
unset MISPMM_CSR_TILING
# ELL K=256 too
for t in "-" "2,4" "1,8"; do
  if [ "$t" = "-" ]; then unset MISPMM_CSR_TILING; else export MISPMM_CSR_TILING=$t; fi
  python3 bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ELL K=256 tiling=$t', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'])
" | tee -a $O/tiling_sweep.log
done
