#!/bin/bash
# stamps of the general CSR kernel on the ragged matrices (config 2 = delaunay_n12, ACTIVSg10K as CSR)
set -o pipefail
OUT=gpurun_out/r3s32
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for m in delaunay_n12 ACTIVSg10K; do
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_headline.py --matrix $m --graph 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_ragged.log
done
echo done
