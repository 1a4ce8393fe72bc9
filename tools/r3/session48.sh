#!/bin/bash
# 4-chunk rows: the whole row tested at once before the chunk-by-chunk walk -- parity (split, two-body, fuzz), stamps, A/B numbers
set -o pipefail
OUT=gpurun_out/r3s48
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/tests.log || exit 1
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_split.py --acc reference 2>&1 | grep -v amdgpu.ids | tee -a $OUT/stamps_split.log
for acc in reference fast; do
for n in 128 256 64; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc $acc --k-cols $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hybrid_ab.log
done
done
echo done
