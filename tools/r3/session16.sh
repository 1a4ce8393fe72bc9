#!/bin/bash
# Round-3 GPU session 16: which operand's placement moves the headline's time?
set -o pipefail
OUT=gpurun_out/r3s16
mkdir -p $OUT
for v in c b both; do
  echo "== vary $v"; timeout -k 10 300 python tools/probe/placement_probe.py --copies 8 --vary $v 2>&1 | grep -v amdgpu.ids | tee -a $OUT/placement_$v.log
done
echo done
