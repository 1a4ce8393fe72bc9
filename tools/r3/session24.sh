#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s24
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for rep in 1 2; do
timeout -k 10 400 python tools/probe/bsr_ab_probe.py "tuning-build=$P/libmispmm_tune.so" "stagger-8=$P/libmispmm_x_stag8.so" "stagger-16=$P/libmispmm_x_stag16.so" "stagger-24=$P/libmispmm_x_stag24.so" "stagger-32=$P/libmispmm_x_stag32.so" "production=$P/libmispmm.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_ab.log
done
echo done
