#!/bin/bash
# should matrices whose longest row is between 33 and 63 entries get a span list (two-body launch) too?
set -o pipefail
OUT=gpurun_out/r3s44
mkdir -p $OUT
for m in g7jac010 ACTIVSg10K tols4000 delaunay_n12; do
for acc in reference fast; do
timeout -k 10 300 python tools/probe/hybrid_longrows_probe.py --force --matrix $m --acc $acc 2>&1 | grep -v amdgpu.ids | tee -a $OUT/force_spans.log
done
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo done
