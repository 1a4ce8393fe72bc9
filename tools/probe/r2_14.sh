#!/bin/bash
export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_tune.so
for k in 64 128 192 256 384 512; do
  for lr in 0 1; do
    MISPMM_LONGROWS=$lr python3 bench.py --matrix GL7d25 --k-cols $k --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GL7d25 K=$k longrows=$lr', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'])
"
  done
done | tee gpurun_out/longrows_k.log
