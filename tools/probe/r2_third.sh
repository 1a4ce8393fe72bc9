#!/bin/bash
# round-2 third GPU pass: gather-shape microbenchmark, new multi-GPU / CLI / tools tests, then the whole gpu suite
set -x
O=gpurun_out/r2c; mkdir -p $O
timeout -k 10 300 tools/micro/gather_shape > $O/gather_shape.log 2>&1
timeout -k 10 900 python3 -m pytest tests/test_gpu_multi.py tests/test_tools.py tests/test_cli.py -x -q -m gpu > $O/pytest_new.log 2>&1
tail -15 $O/pytest_new.log
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -x > $O/pytest_all.log 2>&1
tail -5 $O/pytest_all.log
