#!/bin/bash
# stamps of the split body: where the long rows of GL7d25 spend their time, REFERENCE against FAST
set -o pipefail
OUT=gpurun_out/r3s46
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for acc in reference fast; do
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_split.py --acc $acc 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_split.log
done
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_split.py --acc reference --b-mode exact 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_split.log
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_split.py --acc reference --kernel 6 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_split.log
echo done
