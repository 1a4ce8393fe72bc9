#!/bin/bash
# small matrices (config 2 and the medium_* directories): all of a short row's reads in flight at once (16 per lane, one batch)
# against the rolling body with 8 -- these launches have 8 waves per CU, registers are free
set -o pipefail
OUT=gpurun_out/r3s41
mkdir -p $OUT
P=cuda-optimization-for-spmm_amd
for m in delaunay_n12 ch7-6-b5 dw1024 qh1484 g7jac010 ACTIVSg10K n4c6-b13; do
timeout -k 10 300 python tools/probe/lib_ab_probe.py --entry general --matrix $m "rolling-8=$P/libmispmm_tune.so" "one-batch-16=$P/libmispmm_tune.so:MISPMM_UMAX=16;MISPMM_ROLL=0" "rolling-16-of-32=$P/libmispmm_tune.so:MISPMM_DEEP=1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/small_ab.log
done
echo done
