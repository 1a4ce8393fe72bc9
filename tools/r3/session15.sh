#!/bin/bash
# Round-3 GPU session 15: the lines kept under profiles/r3/ (driver's flags), every BASELINE configuration with extras and the CPU leg.
set -o pipefail
OUT=gpurun_out/r3s15
mkdir -p $OUT
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config'].get('kernel_tag'), d.get('cpu_baseline',{}).get('value'))"; }
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2>> $OUT/err.log || exit 1
show $OUT/bench_default.json "driver's command"
for cfg in 2 3 4 5; do
  timeout -k 10 400 python bench.py --config $cfg --steps 20 --warmup 5 > $OUT/bench_cfg$cfg.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg$cfg.json "cfg $cfg"
done
timeout -k 10 400 python bench.py --steps 2000 --warmup 200 --no-extras --no-cpu-baseline > $OUT/bench_2000_200.json 2>> $OUT/err.log || exit 1
show $OUT/bench_2000_200.json "headline 2000/200"
MISPMM_NO_HINT=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_general_entry.json 2>> $OUT/err.log || exit 1
show $OUT/bench_general_entry.json "headline, general entry"
for acc in reference fast; do
  timeout -k 10 400 python bench.py --matrix GL7d25 --acc $acc --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_GL7d25_$acc.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_GL7d25_$acc.json "GL7d25 $acc"
done
MISPMM_FORCE_DIST=1 timeout -k 10 400 python bench.py --gpus 1 --config 5 --steps 64 --warmup 16 > $OUT/bench_dist_world1_k512.json 2>> $OUT/err.log || exit 1
python -c "import json;d=json.load(open('$OUT/bench_dist_world1_k512.json'));print('dist world 1 K=512', d['ms_per_step'], d['kernel_only'], d['roofline'], d['ranks_seen']['devices'], d['cpu_baseline']['value'])"
MISPMM_SHARE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --steps 64 --warmup 16 --no-cpu-baseline > $OUT/bench_dist_2ranks_one_card.json 2>> $OUT/err.log || exit 1
python -c "import json;d=json.load(open('$OUT/bench_dist_2ranks_one_card.json'));print('2 ranks one card', d['ms_per_step'], d['exchange_modes'], d['ranks_seen']['distinct_devices'])"
timeout -k 10 600 python tools/config_sweep.py > $OUT/config_sweep.log 2>&1 || exit 1
tail -30 $OUT/config_sweep.log | cut -c1-250
echo done
