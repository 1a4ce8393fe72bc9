#!/bin/bash
# is "eight 32-column parts at N = 256" a rule or a constant of one matrix?  XCD grids at N = 256 on other matrices, in-process A/B
set -o pipefail
OUT=gpurun_out/r3s27
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for m in n4c6-b13 ACTIVSg10K delaunay_n12 ch7-6-b5 g7jac010; do
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix $m --k-cols 256 "default-1x8=$P/libmispmm_tune.so" "grid-2x4=$P/libmispmm_tune.so:MISPMM_CSR_TILING=2,4" "grid-4x2=$P/libmispmm_tune.so:MISPMM_CSR_TILING=4,2" "grid-8x1=$P/libmispmm_tune.so:MISPMM_CSR_TILING=8,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tiling_n256.log
done
for m in ACTIVSg10K delaunay_n12 ch7-6-b5; do
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix $m --k-cols 128 "default-4x2=$P/libmispmm_tune.so" "grid-2x4=$P/libmispmm_tune.so:MISPMM_CSR_TILING=2,4" "grid-8x1=$P/libmispmm_tune.so:MISPMM_CSR_TILING=8,1" "grid-1x8=$P/libmispmm_tune.so:MISPMM_CSR_TILING=1,8" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/tiling_n128.log
done
echo done
