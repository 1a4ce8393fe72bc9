#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s20
mkdir -p $OUT
timeout -k 10 300 python tools/probe/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/overlap.log
timeout -k 10 300 python tools/probe/overlap_probe.py --k-cols 512 2>&1 | grep -v amdgpu.ids | tee -a $OUT/overlap.log
timeout -k 10 300 python tools/probe/overlap_probe.py --matrix delaunay_n12 2>&1 | grep -v amdgpu.ids | tee -a $OUT/overlap.log
echo done
