#!/bin/bash
# Round-3 GPU session 18: in-process A/B (same operands) of the round's kernel decisions
set -o pipefail
OUT=gpurun_out/r3s18
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for rep in 1 2; do
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry uniform "production=$P/libmispmm.so" "tuning-build=$P/libmispmm_tune.so" "sc1-stores=$P/libmispmm_x_st16.so" \
  "plain-stores=$P/libmispmm_tune.so:MISPMM_STORE_SC1=0" "setprio=$P/libmispmm_x_prio.so" "block64=$P/libmispmm_tune.so:MISPMM_BLOCK=64" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general "bet-on=$P/libmispmm_tune.so:MISPMM_ROW_GUESS=1" "bet-off=$P/libmispmm_tune.so:MISPMM_ROW_GUESS=0" \
  "production=$P/libmispmm.so" "two-bodies=$P/libmispmm_x_fewbodies.so" "two-bodies-bet-off=$P/libmispmm_x_fewbodies.so:MISPMM_ROW_GUESS=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
done
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry general --matrix delaunay_n12 "five-bodies=$P/libmispmm_tune.so" "two-bodies=$P/libmispmm_x_fewbodies.so" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
timeout -k 10 400 python tools/probe/lib_ab_probe.py --entry uniform --k-cols 256 "nt=$P/libmispmm_tune.so" "sc1-stores=$P/libmispmm_x_st16.so" "plain-stores=$P/libmispmm_tune.so:MISPMM_STORE_SC1=0" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/lib_ab.log
echo done
