#!/bin/bash
O=gpurun_out/r2k; mkdir -p $O
MISPMM_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --exchange both > $O/bench_dist_world1_rccl.json 2> $O/err1.log
MISPMM_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --exchange both > $O/bench_dist_2ranks_one_card_peer.json 2> $O/err2.log
MISPMM_FORCE_DIST=1 python3 bench.py --gpus 1 --config 5 --steps 20 --warmup 5 --exchange both > $O/bench_dist_world1_rccl_k512.json 2> $O/err3.log
tail -c 300 $O/err1.log $O/err2.log $O/err3.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2k/*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['n_gpus'], d['value'], d['ms_per_step'], d['exchange_modes'], d['kernel_only'])
    except Exception as e: print(f, 'ERR', e)
PY
# single-process multi-device CLI on one card
python3 - <<'PY'
import sys, os, subprocess, tempfile, numpy as np
sys.path.insert(0,'cuda-optimization-for-spmm_amd')
from mispmm import datasets, formats
with tempfile.TemporaryDirectory() as t:
    d=os.path.join(t,'large_25605'); os.makedirs(d)
    formats.write_csr(os.path.join(d,'n4c6-b13.csr'), datasets.load_csr('n4c6-b13', dtype=np.float64), integer=True)
    for g in ('first','rccl','none'):
        p=subprocess.run(['cuda-optimization-for-spmm_amd/cuspmm','--csr','-k','512','--gpus','1','--gather',g,'--no-vendor','--iters','200','-d',d],capture_output=True,text=True)
        rec=[b for b in p.stdout.split('},') if 'ngpus' in b]
        open('gpurun_out/r2k/cli_gpus1_%s.txt'%g,'w').write(rec[0] if rec else p.stdout+p.stderr)
        print(g, rec[0].replace('\n',' ')[-330:] if rec else 'NO RECORD')
PY
