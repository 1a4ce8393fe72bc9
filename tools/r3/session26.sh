#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s26
mkdir -p $OUT
for cfg in headline 4 5; do
  t0=$(date +%s); timeout -k 10 600 python bench.py --gpus 1 --config $cfg --steps 20 --warmup 5 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err || { tail -20 $OUT/bench_$cfg.err; exit 1; }
  echo "wall $(( $(date +%s) - t0 )) s"
  python -c "import json;d=json.load(open('$OUT/bench_$cfg.json'));r=d['roofline'];print('$cfg', round(d['ms_per_step']*1e3,4), r['frac'], r['traffic'], r['note'][-260:])"
done
echo done
