#!/bin/bash
# the in-process sweep of every configuration / kernel and the sparsity sweep, final binary
set -o pipefail
OUT=gpurun_out/r3s52
mkdir -p $OUT
timeout -k 10 600 python tools/config_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/config_sweep.log; tail -5 $OUT/config_sweep.log | cut -c1-200
timeout -k 10 900 python tools/sparsity_sweep.py --out $OUT/sparsity 2>&1 | grep -v amdgpu.ids > $OUT/sparsity_sweep.log; tail -5 $OUT/sparsity_sweep.log | cut -c1-200
echo done
