#!/bin/bash
# two-body launch: the row-gather body on the split body's XCD grid (aligned) against its own grid
set -o pipefail
OUT=gpurun_out/r3s36
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "split or two_body or long_row or coo_and_bsr" 2>&1 | tail -6 | tee $OUT/tests.log || exit 1
for align in 1 0; do
for acc in reference fast; do
for n in 128 64 256 32; do
echo "## MISPMM_HYBRID_ALIGN=$align" | tee -a $OUT/hybrid_ab.log
MISPMM_HYBRID_ALIGN=$align MISPMM_LIB=$P/libmispmm_tune.so timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc $acc --k-cols $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/hybrid_ab.log
done
done
done
echo done
