#!/bin/bash
# Round-3 GPU session 3: stamps of the bf16 BSR kernel, run-to-run distribution of the headline with nt stores, full GPU suite.
set -o pipefail
OUT=gpurun_out/r3s3
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
echo "== stamps bsrc_slots"
MISPMM_LIB=$PKG/libmispmm_stamps.so timeout -k 10 300 python tools/stamp_bsr.py 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_bsrc_slots.log || exit 1
echo "== headline, production library, 6 processes"
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_head_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_head_$i.json "head run $i"
done
echo "== full GPU suite"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -5 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
echo "== every config, production library, driver flags"
for cfg in headline 2 3 4 5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 > $OUT/bench_cfg$cfg.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg$cfg.json "cfg $cfg"
done
echo done
