#!/bin/bash
set -x
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_multi.py -x -q -m gpu > $O/pytest_multi.log 2>&1
tail -5 $O/pytest_multi.log
MISPMM_ROWS2=1 timeout -k 10 900 python3 -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest_rows2.log 2>&1
tail -5 $O/pytest_rows2.log
for v in "X=0" "MISPMM_ROWS2=1"; do
  tag=$(echo "$v" | tr ' =,' '___')
  for c in headline 2 3 5; do
    env $v python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_cfg$c.json 2>> $O/bench.err
  done
  env $v MISPMM_NO_HINT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_nohint.json 2>> $O/bench.err
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix GL7d25 > $O/bench_${tag}_gl7d25.json 2>> $O/bench.err
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix ACTIVSg10K > $O/bench_${tag}_activ.json 2>> $O/bench.err
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --k-cols 256 > $O/bench_${tag}_k256.json 2>> $O/bench.err
done
grep -v amdgpu.ids $O/bench.err | tail -5
