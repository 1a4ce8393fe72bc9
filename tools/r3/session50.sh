#!/bin/bash
# the driver's launch form for N > 1 (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE), rehearsed with two ranks on
# the one card (MISPMM_SHARE_GPU=1: gloo control, IPC-mapped buffers), and with one rank over RCCL
set -o pipefail
OUT=gpurun_out/r3s50
mkdir -p $OUT
MISPMM_SHARE_GPU=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/bench_torchrun_2ranks_one_card.json 2> $OUT/torchrun2.err || { tail -30 $OUT/torchrun2.err; exit 1; }
python -c "import json;d=json.loads(open('$OUT/bench_torchrun_2ranks_one_card.json').read().strip().splitlines()[-1]);print('2 ranks:', d['n_gpus'], d['value'], d['ms_per_step'], d['ranks_seen']['world_size'], d['ranks_seen']['backend'], {k:(v.get('value') or v) for k,v in d['exchange_modes'].items()}, d['cpu_baseline']['gpu_parity'])"
MISPMM_FORCE_DIST=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_torchrun_1rank_rccl.json 2> $OUT/torchrun1.err || { tail -30 $OUT/torchrun1.err; exit 1; }
python -c "import json;d=json.loads(open('$OUT/bench_torchrun_1rank_rccl.json').read().strip().splitlines()[-1]);print('1 rank RCCL:', d['n_gpus'], d['value'], d['ms_per_step'], d['ranks_seen']['backend'], {k:(v.get('value') or v) for k,v in d['exchange_modes'].items()}, d['kernel_only'])"
echo done
