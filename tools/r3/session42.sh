#!/bin/bash
# ordered pass of the split body software-pipelined (reads of 8 terms in flight under the adds of the 8 before): parity of
# everything, the fuzz suite at four times its size, then GL7d25 (before: two-body 5.09-5.16 / split 6.99 us REFERENCE;
# CLI: CSR 5.14, COO / BSR / ELL 7.6 us) and tols4000
set -o pipefail
OUT=gpurun_out/r3s42
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
MISPMM_FUZZ_SCALE=4 timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_fuzz.log 2>&1; rc=$?
tail -4 $OUT/pytest_fuzz.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for n in 128 64 256; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --ab --acc reference --k-cols $n 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ordered_pipelined.log
done
timeout -k 10 600 python tools/sweep.py --dirs large_21074,medium_4000,medium_2880 --iters 200 --out $OUT/sweep 2>&1 | tail -14 | tee $OUT/sweep.log
echo done
