#!/bin/bash
O=gpurun_out/r2o; mkdir -p $O
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1; tail -4 $O/pytest_all.log
