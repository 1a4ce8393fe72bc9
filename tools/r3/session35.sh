#!/bin/bash
# two-body launch for long-row matrices (csr_hybrid): parity, then GL7d25 against the split kernel, then the headline
# (the row-gather kernel became a wrapper around a body function: same code expected) against the build before
set -o pipefail
OUT=gpurun_out/r3s35
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "split or two_body or long_row or coo_and_bsr" 2>&1 | tail -6 | tee $OUT/tests.log || exit 1
for acc in reference fast; do
for n in 128 64 256; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc $acc --k-cols $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hybrid_ab.log
done
done
timeout -k 10 300 python tools/probe/lib_ab_probe.py --entry uniform "refactored=$P/libmispmm_tune.so" "before=$P/libmispmm_x_base.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/refactor_ab.log
timeout -k 10 300 python tools/probe/lib_ab_probe.py --entry general --matrix delaunay_n12 "refactored=$P/libmispmm_tune.so" "before=$P/libmispmm_x_base.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/refactor_ab.log
echo done
