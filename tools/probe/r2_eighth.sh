#!/bin/bash
set -x
O=gpurun_out/r2h; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_spmm.py -x -q -m gpu -k "bsrc or bsr" > $O/pytest_bsr.log 2>&1
tail -12 $O/pytest_bsr.log
python3 bench.py --config 4 --steps 20 --warmup 5 --cpu-seconds 2 > $O/bench_cfg4.json 2>> $O/bench.err
grep -v amdgpu.ids $O/bench.err | tail -5
python3 -c "
import json
d=json.load(open('$O/bench_cfg4.json')); print(d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'], d['cpu_baseline']['gpu_parity'], d['mfma'], d['other_bsr_kernel'])
"
