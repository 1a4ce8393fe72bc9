#!/bin/bash
O=gpurun_out/r2q; mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log
for m in delaunay_n12 qh1484 dw1024 g7jac010; do
  for h in 0 1; do
    MISPMM_NO_HINT=$h python3 bench.py --matrix $m --k-cols 128 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m no_hint=$h', d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'])
"
  done
done | tee $O/padded.log
python3 bench.py --config 2 --steps 20 --warmup 5 --cpu-seconds 2 > $O/bench_cfg2.json 2>/dev/null; python3 -c "
import json
d=json.load(open('$O/bench_cfg2.json')); print(d['value'], d['roofline']['launch_us'], d['roofline']['frac'], d['config']['kernel_tag'], d['cpu_baseline']['gpu_parity'], d['batched']['us_per_product'])"
