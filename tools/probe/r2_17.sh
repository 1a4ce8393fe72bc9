#!/bin/bash
O=gpurun_out/r2n; mkdir -p $O
timeout -k 10 1000 python3 tools/sweep.py -k 128 --iters 200 --out $O/sweep > $O/sweep_stdout.log 2>&1; tail -60 $O/sweep_stdout.log
timeout -k 10 600 python3 tools/sparsity_sweep.py --iters 20 --out $O/sparsity > $O/sparsity_stdout.log 2>&1; tail -5 $O/sparsity_stdout.log
