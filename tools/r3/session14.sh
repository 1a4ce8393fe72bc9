#!/bin/bash
# Round-3 GPU session 14: the reference's integration sweeps through the cuspmm CLI (12 data directories x 4 formats) at K = 128 and K = 512
# (K = 512 exercises the clustered plan order of the host layer), and the sparsity sweep.
set -o pipefail
OUT=gpurun_out/r3s14
mkdir -p $OUT
timeout -k 10 900 python tools/sweep.py -k 128 --iters 200 --out $OUT/sweep_k128 > $OUT/sweep_k128.log 2>&1 || { tail -20 $OUT/sweep_k128.log; exit 1; }
tail -60 $OUT/sweep_k128.log
timeout -k 10 900 python tools/sweep.py --formats csr -k 512 --iters 100 --dirs large_25605,large_20000,large_15120 --out $OUT/sweep_k512 > $OUT/sweep_k512.log 2>&1 || { tail -20 $OUT/sweep_k512.log; exit 1; }
tail -30 $OUT/sweep_k512.log
timeout -k 10 600 python tools/sparsity_sweep.py > $OUT/sparsity.log 2>&1 || { tail -20 $OUT/sparsity.log; exit 1; }
tail -12 $OUT/sparsity.log
echo done
