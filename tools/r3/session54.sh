#!/bin/bash
# the shared reduce for an fp32 C only: the slots tests (incl. the three-way bit comparison), config 4 both C types, sweeps
set -o pipefail
OUT=gpurun_out/r3s54
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q -k "slots or bsr" 2>&1 | tail -3 | tee $OUT/tests.log || exit 1
timeout -k 10 300 python tools/probe/bsr_ab_probe.py --c-bf16 "production=$P/libmispmm.so" "forced-share=$P/libmispmm_tune.so:MISPMM_BSR_SHARE=2" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
timeout -k 10 300 python tools/probe/bsr_ab_probe.py "production=$P/libmispmm.so" "wave0-reduces=$P/libmispmm_tune.so:MISPMM_BSR_SHARE=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
timeout -k 10 600 python tools/config_sweep.py 2>&1 | grep -v amdgpu.ids > $OUT/config_sweep.log; grep "slots" $OUT/config_sweep.log | cut -c1-200
timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_cfg4.json 2>/dev/null && python -c "import json;d=json.load(open('$OUT/bench_cfg4.json'));print('cfg 4', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['config']['kernel_tag'])"
echo done
