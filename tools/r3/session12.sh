#!/bin/bash
# Round-3 GPU session 12: the general entry point on the kernel specialised for its bet
set -o pipefail
OUT=gpurun_out/r3s12
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
run() { local tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/$tag.json 2>> $OUT/err.log || exit 1; show $OUT/$tag.json "$tag"; }
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_sel.log 2>&1; rc=$?
tail -5 $OUT/pytest_sel.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  run nohint_prod_$i MISPMM_NO_HINT=1 X=1
  run nohint_tune_bet0_$i MISPMM_NO_HINT=1 MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_ROW_GUESS=0
  run hint_prod_$i X=1
done
echo done
