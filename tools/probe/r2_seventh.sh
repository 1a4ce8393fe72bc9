#!/bin/bash
set -x
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
tail -8 $O/pytest_all.log
python3 tools/config_sweep.py > $O/config_sweep.log 2>&1
tail -20 $O/config_sweep.log
