#!/bin/bash
# Round-3 GPU session 5: bsrc_slots at 6 workgroups per CU; clustered row order (plan) at K = 128 / 256 / 512.
set -o pipefail
OUT=gpurun_out/r3s5
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'], d.get('batched',{}).get('us_per_product'))"; }
echo "== selected tests"
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_multi.py -m gpu -x -q -k "bsrc or plan or sentinel or strided or allgather_over" > $OUT/pytest_sel.log 2>&1; rc=$?
tail -8 $OUT/pytest_sel.log
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
echo "== clustered row order: plan (MIN_N=0) vs none (MIN_N=100000)"
for k in 128 256 512; do
  for mn in 0 100000; do
    MISPMM_PLAN_MIN_N=$mn timeout -k 10 300 python bench.py --k-cols $k --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k${k}_plan$mn.json 2>> $OUT/err.log || exit 1
    show $OUT/bench_k${k}_plan$mn.json "n4c6-b13 K=$k plan_min_n=$mn"
  done
done
for mn in 0 100000; do
  MISPMM_PLAN_MIN_N=$mn timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_cfg2_plan$mn.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg2_plan$mn.json "cfg 2 plan_min_n=$mn"
  MISPMM_PLAN_MIN_N=$mn timeout -k 10 300 python bench.py --matrix ACTIVSg10K --k-cols 256 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_activ_plan$mn.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_activ_plan$mn.json "ACTIVSg10K K=256 plan_min_n=$mn"
done
echo "== config sweep (BSR section)"
timeout -k 10 600 python tools/config_sweep.py 2>&1 | grep '"4' | cut -c1-260
echo done
