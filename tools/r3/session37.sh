#!/bin/bash
# two-body launch: where to cut the span list (rows of more than T entries to the split body)
set -o pipefail
OUT=gpurun_out/r3s37
mkdir -p $OUT
for acc in reference fast; do
for t in 12 16 24 32 48 64; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc $acc --k-cols 128 --threshold $t 2>&1 | grep -v amdgpu.ids | grep -v "^split" | tee -a $OUT/threshold.log
done
done
for t in 16 24 48; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc reference --k-cols 256 --threshold $t 2>&1 | grep -v amdgpu.ids | grep -v "^split" | tee -a $OUT/threshold.log
done
echo done
