#!/bin/bash
# what REFERENCE mode pays on the long rows: uniform B (1-4 waves per launch fail the exactness test and walk their chunk in
# entry order) against a grid of values on which no test fails; and the split kernel alone after the revert
set -o pipefail
OUT=gpurun_out/r3s43
mkdir -p $OUT
for mode in uniform exact; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc reference --k-cols 128 --b-mode $mode 2>&1 | grep -v amdgpu.ids | tee -a $OUT/resum_cost.log
done
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc fast --k-cols 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/resum_cost.log
echo done
