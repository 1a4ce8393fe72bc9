#!/bin/bash
# Round-3 GPU session 8: row mapping as a template parameter (headline back to its nt level?), the bet on uniform rows.
set -o pipefail
OUT=gpurun_out/r3s8
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
timeout -k 10 900 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "plan or bets or csr_matches or batch" > $OUT/pytest_sel.log 2>&1; rc=$?
tail -5 $OUT/pytest_sel.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_hint_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_hint_$i.json "headline, uniform hint, run $i"
  MISPMM_NO_HINT=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_nohint_$i.json "headline, general entry (bet on uniform rows), run $i"
  MISPMM_NO_HINT=1 MISPMM_ROW_GUESS=0 MISPMM_LIB=$PKG/libmispmm_tune.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_noguess_$i.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_nohint_noguess_$i.json "headline, general entry, bet off, run $i"
done
for cfg in 2 3 5; do
  timeout -k 10 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg$cfg.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg$cfg.json "cfg $cfg"
done
MISPMM_ROW_GUESS=0 MISPMM_LIB=$PKG/libmispmm_tune.so timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg2_noguess.json 2>> $OUT/err.log || exit 1
show $OUT/bench_cfg2_noguess.json "cfg 2 bet off (not a candidate: nnz % M != 0)"
echo done
