#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s21
mkdir -p $OUT
timeout -k 10 300 python tools/probe/finegrained_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/finegrained.log
echo done
