#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s25
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for rep in 1 2; do
timeout -k 10 400 python tools/probe/bsr_ab_probe.py "four-waves=$P/libmispmm_tune.so" "two-waves=$P/libmispmm_tune.so:MISPMM_BSR_WAVES=2" "production=$P/libmispmm.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_waves_ab.log
done
echo done
