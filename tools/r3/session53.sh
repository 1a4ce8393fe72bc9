#!/bin/bash
# the shared reduce of the bf16 slots kernel with a bf16 C (the sweep read 4.55 us where round 3's earlier build read 4.24)
set -o pipefail
OUT=gpurun_out/r3s53
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for i in 1 2; do
timeout -k 10 300 python tools/probe/bsr_ab_probe.py --c-bf16 "wave0-reduces=$P/libmispmm_tune.so:MISPMM_BSR_SHARE=0" "shared-reduce=$P/libmispmm_tune.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_share_bf16c.log
done
timeout -k 10 300 python tools/probe/bsr_ab_probe.py "wave0-reduces=$P/libmispmm_tune.so:MISPMM_BSR_SHARE=0" "shared-reduce=$P/libmispmm_tune.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_share_bf16c.log
echo done
