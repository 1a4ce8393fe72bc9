#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s23
mkdir -p $OUT
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_$i.json 2>> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
  python -c "import json;d=json.load(open('$OUT/bench_default_$i.json'));print('driver command run $i', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['timing']['placements_us'], d['cpu_baseline']['gpu_parity'])"
done
for cfg in 2 3 4 5; do
  timeout -k 10 400 python bench.py --config $cfg --steps 20 --warmup 5 > $OUT/bench_cfg$cfg.json 2>> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
  python -c "import json;d=json.load(open('$OUT/bench_cfg$cfg.json'));print('cfg $cfg', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['timing']['placements_us'], d['cpu_baseline']['gpu_parity'])"
done
timeout -k 10 900 python -m pytest tests/test_cli.py -m gpu -x -q -k "floors or batched" 2>&1 | tail -3
echo done
