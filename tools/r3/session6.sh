#!/bin/bash
# Round-3 GPU session 6: bsrc_slots with the column list through the scalar cache.
set -o pipefail
OUT=gpurun_out/r3s6
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "bsrc or plan" > $OUT/pytest_sel.log 2>&1; rc=$?
tail -5 $OUT/pytest_sel.log
[ $rc -eq 0 ] || exit $rc
MISPMM_LIB=$PKG/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_bsr.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_bsrc_slots.log || exit 1
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg4_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg4_$i.json "cfg 4 run $i"
done
timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg5.json 2>> $OUT/err.log || exit 1
show $OUT/bench_cfg5.json "cfg 5 (plan rule)"
echo done
