#!/bin/bash
# Round-3 GPU session 9: general-kernel code size experiment, then the rocprofv3 passes of tools/profile_r3.sh.
set -o pipefail
OUT=gpurun_out/r3s9
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
show() { python -c "import json,sys;d=json.load(open('$1'));print('$2',round(d['ms_per_step']*1e3,4),d['roofline']['frac'],d['config']['kernel_tag'])"; }
for i in 1 2; do
  for lib in tune x_fewbodies; do
    MISPMM_NO_HINT=1 MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_nohint_${lib}_$i.json 2>> $OUT/err.log || exit 1
    show $OUT/bench_nohint_${lib}_$i.json "general entry, $lib, run $i"
  done
done
for lib in tune x_fewbodies; do
  MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --config 2 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_cfg2_$lib.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_cfg2_$lib.json "cfg 2 $lib"
  MISPMM_LIB=$PKG/libmispmm_$lib.so timeout -k 10 300 python bench.py --matrix ACTIVSg10K --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_activ_$lib.json 2>> $OUT/err.log || exit 1
  show $OUT/bench_activ_$lib.json "ACTIVSg10K CSR K=128 $lib"
done
echo "== rocprofv3"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile_r3.sh > $OUT/profile_r3.log 2>&1; rc=$?
tail -40 $OUT/profile_r3.log
exit $rc
