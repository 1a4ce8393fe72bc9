#!/bin/bash
O=gpurun_out/r2m; mkdir -p $O
for n in 2 4; do
  MISPMM_SHARE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 20 --warmup 5 --exchange both > $O/shared_$n.json 2> $O/shared_$n.err
  echo "rc=$?"; tail -c 600 $O/shared_$n.err | grep -v "CudaIPC\|amdgpu.ids" | tail -5
  python3 -c "
import json
d=json.load(open('$O/shared_$n.json')); print($n, d['value'], d['ms_per_step'], d['exchange_modes'])
"
done
