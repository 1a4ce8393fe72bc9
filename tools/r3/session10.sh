#!/bin/bash
# Round-3 GPU session 10: clean A/B of the bet on uniform rows (same library, knob on / off); tuning vs production build;
# XCD tilings with non-temporal stores.
set -o pipefail
OUT=gpurun_out/r3s10
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
run() { # tag, env..., -- bench args
  local tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/$tag.json 2>> $OUT/err.log || exit 1
  show $OUT/$tag.json "$tag"
}
for i in 1 2 3; do
  run nohint_tune_bet1_$i MISPMM_NO_HINT=1 MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_ROW_GUESS=1
  run nohint_tune_bet0_$i MISPMM_NO_HINT=1 MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_ROW_GUESS=0
  run nohint_prod_$i MISPMM_NO_HINT=1 X=1
  run hint_prod_$i X=1
  run hint_tune_$i MISPMM_LIB=$PKG/libmispmm_tune.so
done
for t in 4,2 2,4 8,1 1,8; do
  run hint_tune_tiling_${t/,/x} MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_CSR_TILING=$t
done
echo done
