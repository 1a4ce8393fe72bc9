#!/bin/bash
# feasibility of a two-body launch for long-row matrices (tools/probe/hybrid_longrows_probe.py)
set -o pipefail
OUT=gpurun_out/r3s34
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for acc in reference fast; do
MISPMM_SPLIT=0 MISPMM_LIB=$P/libmispmm_tune.so timeout -k 10 400 python tools/probe/hybrid_longrows_probe.py --acc $acc 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hybrid.log
done
echo done
