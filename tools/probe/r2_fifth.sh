#!/bin/bash
set -x
O=gpurun_out/r2e; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -x -q -m gpu -k "bsr" > $O/pytest_bsr.log 2>&1
tail -15 $O/pytest_bsr.log
python3 bench.py --config 4 --steps 20 --warmup 5 --cpu-seconds 2 > $O/bench_cfg4_d8.json 2>> $O/bench.err
MISPMM_BSR_DEPTH=6 python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_d6.json 2>> $O/bench.err
MISPMM_BSR_DEPTH=4 python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_d4.json 2>> $O/bench.err
MISPMM_BSR_LDS=0 python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg4_old.json 2>> $O/bench.err
grep -v amdgpu.ids $O/bench.err | tail -5
python3 -c "
import json
for f in ('d8','d6','d4','old'):
    d=json.load(open('$O/bench_cfg4_%s.json' % f)); print(f, d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'], d.get('cpu_baseline',{}).get('gpu_parity'))
"
