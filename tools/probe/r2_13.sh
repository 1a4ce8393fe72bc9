#!/bin/bash
export MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_tune.so
for lr in 1 0; do echo "== MISPMM_LONGROWS=$lr"; MISPMM_LONGROWS=$lr MISPMM_DEEP=$((1-lr)) python3 tools/probe/long_row_slope_probe.py 2>&1 | grep "^{"; done | tee gpurun_out/long_row_slope.log
