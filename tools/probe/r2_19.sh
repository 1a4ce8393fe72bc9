#!/bin/bash
O=gpurun_out/r2p; mkdir -p $O
for c in headline 2 3 4 5; do
  S=$(date +%s.%N); python3 bench.py --config $c > $O/bench_default_$c.json 2> $O/err_$c.log; echo "config $c wall $(echo "$(date +%s.%N) - $S" | bc) s"
  python3 -c "
import json
d=json.load(open('$O/bench_default_$c.json')); print('$c', d['steps'], d['warmup'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_parity'][:20])
"
done
