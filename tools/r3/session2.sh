#!/bin/bash
# Round-3 GPU session 2: C store cache policy (nt / sc1 / both) on every config; bsrc_slots with 32 KiB LDS.
set -o pipefail
OUT=gpurun_out/r3s2
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
echo "== bsrc tests"
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "bsrc" > $OUT/pytest_bsrc.log 2>&1 || { tail -30 $OUT/pytest_bsrc.log; exit 1; }
tail -2 $OUT/pytest_bsrc.log
echo "== config 4 store policies"
for pass in 1 2; do
for st in 2 -1 16 18; do
  MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_BSR_STORE=$st timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg4_st${st}_p$pass.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg4_st${st}_p$pass.json "cfg4 store $st pass $pass"
done
done
echo "== headline store policies"
for pass in 1 2; do
  for v in tune:128 x_stnt:128 x_st18:128 x_st3:128 x_stnt:64; do
    lib=${v%%:*}; blk=${v##*:}
    MISPMM_LIB=$PKG/libmispmm_$lib.so MISPMM_BLOCK=$blk timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_head_${lib}_b${blk}_p$pass.json 2>> $OUT/err.log || exit 1
    show $OUT/bench_head_${lib}_b${blk}_p$pass.json "head $lib block $blk pass $pass"
  done
done
echo "== other configs: sc1 (tune) vs nt (x_stnt) vs sc1+nt"
for cfg in 2 3 5; do
  for lib in tune x_stnt x_st18; do
    MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg${cfg}_$lib.json 2>> $OUT/err.log || exit 1
    show $OUT/bench_cfg${cfg}_$lib.json "cfg $cfg $lib"
  done
done
for lib in tune x_stnt; do
  MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --k-cols 256 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_k256_$lib.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_k256_$lib.json "K=256 $lib"
  MISPMM_NO_HINT=1 MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_$lib.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_nohint_$lib.json "headline no hint $lib"
done
echo done
