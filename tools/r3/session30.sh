#!/bin/bash
# bf16 slots kernel with the shared reduce: bit-equality test, stamps, and two experiments (A tile read issued before the
# column list returns; LDS reads of two rows in flight)
set -o pipefail
OUT=gpurun_out/r3s30
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "slots" 2>&1 | tail -4 | tee $OUT/tests.log || exit 1
MISPMM_LIB=$P/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_bsr.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_share.log
for i in 1 2; do
timeout -k 10 300 python tools/probe/bsr_ab_probe.py "shared-reduce=$P/libmispmm_tune.so" "early-tile=$P/libmispmm_x_etile.so" "read-batch=$P/libmispmm_x_rbatch.so" "both=$P/libmispmm_x_both.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_share_variants.log
done
echo done
