#!/bin/bash
set -x
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_spmm.py -x -q -m gpu -k "batched or csr_matches or ragged or lds_staged or uniform" > $O/pytest_sel.log 2>&1
tail -8 $O/pytest_sel.log
for c in headline 2 5; do python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cfg$c.json 2>> $O/bench.err; done
MISPMM_NO_HINT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nohint.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix ACTIVSg10K > $O/bench_activ.json 2>> $O/bench.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix ch7-6-b5 > $O/bench_ch7.json 2>> $O/bench.err
grep -v amdgpu.ids $O/bench.err | tail -5
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2f/bench_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'], d.get('batched'))
PY
