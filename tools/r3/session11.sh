#!/bin/bash
# Round-3 GPU session 11: per-wave stamps of the headline kernel, uniform-row entry point vs general entry point (bet on / off)
set -o pipefail
OUT=gpurun_out/r3s11
mkdir -p $OUT
PKG=cuda-optimization-for-spmm_amd
export MISPMM_LIB=$PKG/libmispmm_stamps.so
echo "== uniform-row entry point"; timeout -k 10 300 python tools/stamp_headline.py --graph 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_uniform.log
echo "== general entry point (bet on uniform rows)"; MISPMM_NO_HINT=1 timeout -k 10 300 python tools/stamp_headline.py --graph 2>&1 | grep -v amdgpu.ids | tee $OUT/stamps_general.log
echo done
