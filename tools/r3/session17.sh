#!/bin/bash
set -o pipefail
OUT=gpurun_out/r3s17
mkdir -p $OUT
for k in 128 256 512; do
  timeout -k 10 400 python tools/probe/plan_ab_probe.py --k-cols $k 2>&1 | grep -v amdgpu.ids | tee -a $OUT/plan_ab.log
done
timeout -k 10 400 python tools/probe/plan_ab_probe.py --k-cols 128 --acc fast 2>&1 | grep -v amdgpu.ids | tee -a $OUT/plan_ab.log
timeout -k 10 400 python tools/probe/plan_ab_probe.py --matrix ACTIVSg10K --k-cols 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/plan_ab.log
timeout -k 10 400 python tools/probe/plan_ab_probe.py --matrix delaunay_n12 --k-cols 128 2>&1 | grep -v amdgpu.ids | tee -a $OUT/plan_ab.log
echo done
