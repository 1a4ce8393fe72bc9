#!/bin/bash
set -x
O=gpurun_out/r2j; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log
bash tools/profile_r2.sh > $O/profile_stdout.log 2>&1; tail -5 $O/profile_stdout.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench.err; tail -c 400 $O/bench_20_5.json
