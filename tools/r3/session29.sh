#!/bin/bash
# bf16 slots kernel: reduce + store dealt over the four waves (SHARE) against wave 0 doing both; parity first
set -o pipefail
OUT=gpurun_out/r3s29
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q -k "slots or bsr" 2>&1 | tail -4 | tee $OUT/tests.log || exit 1
for i in 1 2; do
timeout -k 10 300 python tools/probe/bsr_ab_probe.py "wave0-reduces=$P/libmispmm_tune.so:MISPMM_BSR_SHARE=0" "shared-reduce=$P/libmispmm_tune.so" "production=$P/libmispmm.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/bsr_share_ab.log
done
timeout -k 10 300 python tools/stamp_bsr.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_share.log
timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_cfg4.json 2>$OUT/bench_cfg4.err && tail -c 1500 $OUT/bench_cfg4.json
echo done
