#!/bin/bash
for lib in libmispmm.so libmispmm_nt.so; do
  export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/$lib
  for c in headline 3 5; do
    python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib cfg $c', d['roofline']['launch_us'], d['roofline']['frac'])
"
  done
  python3 bench.py --k-cols 256 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib K=256', d['roofline']['launch_us'], d['roofline']['frac'])
"
done | tee gpurun_out/nt_a.log
