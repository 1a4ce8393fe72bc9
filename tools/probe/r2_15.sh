#!/bin/bash
export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_tune.so
for k in 128 256 512; do
  for v in 1 2 4; do
    MISPMM_LONGROWS=1 MISPMM_LONGROWS_VEC=$v python3 bench.py --matrix GL7d25 --k-cols $k --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GL7d25 K=$k vec<=$v', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'], 'fast', d['other_acc_mode']['launch_us'])
"
  done
done | tee gpurun_out/longrows_vec.log
