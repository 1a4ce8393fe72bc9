#!/bin/bash
# Round-3 GPU session 7: full GPU suite after the plan / guess / batch / CLI changes; general-entry (no hint) line.
set -o pipefail
OUT=gpurun_out/r3s7
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -15 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  MISPMM_NO_HINT=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_nohint_$i.json "headline, general entry (no hint), run $i"
  MISPMM_NO_HINT=1 MISPMM_ROW_GUESS=0 MISPMM_LIB=$PKG/libmispmm_tune.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_noguess_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_nohint_noguess_$i.json "headline, general entry, guess off, run $i"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_hint_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_hint_$i.json "headline, uniform hint, run $i"
done
MISPMM_NO_HINT=1 timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg2.json 2>> $OUT/err.log || exit 1
show $OUT/bench_cfg2.json "cfg 2"
echo done
