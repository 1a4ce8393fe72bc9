#!/bin/bash
# Round-3 GPU session 13: does operand placement explain the process-to-process spread of the headline (3.39 .. 3.58 us)?
set -o pipefail
OUT=gpurun_out/r3s13
mkdir -p $OUT
for i in 1 2 3; do
  echo "== process $i"; timeout -k 10 300 python tools/probe/placement_probe.py --copies 6 2>&1 | grep -v amdgpu.ids | tee -a $OUT/placement.log
done
echo "== with 7 MiB pads"; timeout -k 10 300 python tools/probe/placement_probe.py --copies 6 --pad-mib 7 2>&1 | grep -v amdgpu.ids | tee -a $OUT/placement.log
echo done
