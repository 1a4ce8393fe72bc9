#!/bin/bash
# round-2 first GPU pass: driver-flag bench, long bench, probes
set -x
mkdir -p gpurun_out/r2a
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a/bench_20_5.json 2> gpurun_out/r2a/bench_20_5.err
python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline > gpurun_out/r2a/bench_2000_200.json 2> gpurun_out/r2a/bench_2000_200.err
python3 tools/probe/l2_resident_probe.py > gpurun_out/r2a/l2_probe.log 2>&1
MISPMM_LIB=$PWD/cuda-optimization-for-spmm_amd/libmispmm_stamps.so python3 tools/stamp_headline.py --graph > gpurun_out/r2a/stamps.log 2>&1
for c in 2 3 4 5; do python3 bench.py --config $c --steps 20 --warmup 5 --cpu-seconds 3 > gpurun_out/r2a/bench_cfg$c.json 2> gpurun_out/r2a/bench_cfg$c.err; done
tail -c 600 gpurun_out/r2a/*.err
