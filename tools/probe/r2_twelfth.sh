#!/bin/bash
O=gpurun_out/r2l; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -x -q -m gpu -k "long_row or csr_matches or ragged or fuzz or nonfinite" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_tune.so
for m in GL7d25 g7jac010 tols4000 ACTIVSg10K; do
  for lr in 0 1; do
    for k in 128 512; do
      MISPMM_LONGROWS=$lr python3 bench.py --matrix $m --k-cols $k --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m K=$k longrows=$lr', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'])
" | tee -a $O/longrows.log
    done
  done
done
