#!/bin/bash
# round-2 second GPU pass: launch pitch / gap / span from per-launch stamps, CU-tile mapping variants, bench methodology check
set -x
O=gpurun_out/r2b; mkdir -p $O
S=$PWD/cuda-optimization-for-spmm_amd/libmispmm_stamps.so
MISPMM_LIB=$S python3 tools/stamp_headline.py --graph --launches 64 > $O/stamps_default.log 2>&1
MISPMM_CUTILE=1 MISPMM_LIB=$S python3 tools/stamp_headline.py --graph --launches 64 > $O/stamps_cut1.log 2>&1
MISPMM_CUTILE=1 MISPMM_CUT_LDS=84000 MISPMM_LIB=$S python3 tools/stamp_headline.py --graph --launches 64 > $O/stamps_cut1_lds.log 2>&1
MISPMM_CUTILE=2 MISPMM_LIB=$S python3 tools/stamp_headline.py --graph --launches 64 > $O/stamps_cut2.log 2>&1
MISPMM_CUTILE=1 python3 -m pytest tests/test_gpu_spmm.py -x -q -m gpu -k "uniform or csr_matches or ell_matches or exact_grid" > $O/pytest_cut1.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/bench_2000_200.json 2>> $O/bench_20_5.err
for v in "MISPMM_CUTILE=1" "MISPMM_CUTILE=1 MISPMM_CUT_LDS=84000" "MISPMM_CUTILE=2" "MISPMM_CUTILE=3" "MISPMM_CUTILE=1 MISPMM_CSR_TILING=8,1" "MISPMM_CUTILE=1 MISPMM_CSR_TILING=2,4"; do
  tag=$(echo "$v" | tr ' =,' '___')
  env $v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$tag.json 2>> $O/bench_20_5.err
done
for c in 3 5; do MISPMM_CUTILE=1 python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_cut1_cfg$c.json 2>> $O/bench_20_5.err; done
tail -5 $O/pytest_cut1.log
