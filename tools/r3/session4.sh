#!/bin/bash
# Round-3 GPU session 4: bsrc_slots with 24 KiB LDS / 80 VGPRs (6 workgroups per CU), new multi-GPU tests.
set -o pipefail
OUT=gpurun_out/r3s4
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
echo "== bsrc + multi tests"
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_multi.py -m gpu -x -q -k "agrees_with_the_plain or sentinel" > $OUT/pytest_sel.log 2>&1; rc=$?
tail -15 $OUT/pytest_sel.log
[ $rc -eq 0 ] || exit $rc
echo "== stamps bsrc_slots"
MISPMM_LIB=$PKG/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_bsr.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_bsrc_slots.log || exit 1
echo "== config 4"
for i in 1 2; do
  timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg4_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg4_$i.json "cfg 4 run $i"
done
python -c "import json;d=json.load(open('$OUT/bench_cfg4_1.json'));print(d['other_bsr_kernels'])"
for st in -1 16; do
  MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_BSR_STORE=$st timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg4_st$st.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg4_st$st.json "cfg4 store $st"
done
echo "== config 4 at K=256 and bf16 C through config_sweep"
timeout -k 10 600 python tools/config_sweep.py 2>&1 | tail -25
echo done
