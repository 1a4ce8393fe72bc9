#!/bin/bash
# Round-3 GPU session 1: new bf16 BSR kernel (tests + bench) and single-launch experiments on the headline.
# Run from the repository root on the GPU box: bash tools/r3/session1.sh
set -o pipefail
OUT=gpurun_out/r3s1
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
echo "== bsrc_slots tests"
timeout -k 10 600 python -m pytest tests/test_gpu_spmm.py -m gpu -x -q -k "bsrc" > $OUT/pytest_bsrc.log 2>&1 || { tail -30 $OUT/pytest_bsrc.log; exit 1; }
tail -3 $OUT/pytest_bsrc.log
echo "== config 4"
timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err || { tail -20 $OUT/bench_cfg4.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3s1/bench_cfg4.json'))
print('cfg4', d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['kernel_tag'], {k:(v['launch_us'],v['roofline_frac']) for k,v in d.get('other_bsr_kernels',{}).items()})
PY
MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_BSR_SC1=0 timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg4_plain_stores.json 2>> $OUT/bench_cfg4.err || exit 1
python -c "import json;d=json.load(open('$OUT/bench_cfg4_plain_stores.json'));print('cfg4 plain stores',d['ms_per_step']*1e3,d['config']['kernel_tag'])"
echo "== headline variants (two interleaved passes)"
for pass in 1 2; do
  for v in tune:128 x_prio:128 x_stnt:128 x_stsc01:128 tune:64 tune:256; do
    lib=${v%%:*}; blk=${v##*:}
    MISPMM_LIB=$PKG/libmispmm_$lib.so MISPMM_BLOCK=$blk timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_head_${lib}_b${blk}_p$pass.json 2>> $OUT/bench_head.err || exit 1
    python -c "import json;d=json.load(open('$OUT/bench_head_${lib}_b${blk}_p$pass.json'));print('head $lib block $blk pass $pass',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"
  done
done
MISPMM_LIB=$PKG/libmispmm_tune.so MISPMM_STORE_SC1=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_head_plain.json 2>> $OUT/bench_head.err || exit 1
python -c "import json;d=json.load(open('$OUT/bench_head_plain.json'));print('head plain stores',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"
echo "== row clustering (time only: C rows come out permuted)"
export MISPMM_LIB=$PKG/libmispmm_tune.so
for k in 128 512; do
  for cl in 0 4 8 788; do
    echo "-- K=$k cluster parts $cl"
    timeout -k 10 300 python tools/kernel_sweep.py --kernels 5 --k-cols $k --iters 300 --rounds 5 --cluster $cl 2>&1 | grep '"graph"' | grep reference || exit 1
  done
done
echo "== K=512: 32-column groups (two passes per XCD part)"
MISPMM_GROUP=8 timeout -k 10 300 python tools/kernel_sweep.py --kernels 5 --k-cols 512 --iters 300 --rounds 5 2>&1 | grep '"graph"' || exit 1
echo done
