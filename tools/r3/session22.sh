#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s22
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for rep in 1 2; do
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general "tuning-build=$P/libmispmm_tune.so" "72-vgpr-budget=$P/libmispmm_x_w7.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry uniform "tuning-build=$P/libmispmm_tune.so" "72-vgpr-budget=$P/libmispmm_x_w7.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
done
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix delaunay_n12 "tuning-build=$P/libmispmm_tune.so" "72-vgpr-budget=$P/libmispmm_x_w7.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix ACTIVSg10K "tuning-build=$P/libmispmm_tune.so" "72-vgpr-budget=$P/libmispmm_x_w7.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
echo done
