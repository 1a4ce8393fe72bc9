#!/bin/bash
# evidence of the final binary: full GPU suite + smoke, the driver's command three times, every configuration, GL7d25, then the rocprofv3 passes
set -o pipefail
OUT=gpurun_out/r3s40
mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?
tail -6 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
for i in 1 2 3; do
  t0=$(date +%s); timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_$i.json 2>> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
  echo "wall $(( $(date +%s) - t0 )) s"
  python -c "import json;d=json.load(open('$OUT/bench_default_$i.json'));print('driver command run $i', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['roofline']['traffic'], d['timing']['placements_us'], d['cpu_baseline']['gpu_parity'])"
done
for cfg in 2 3 4 5; do
  timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 > $OUT/bench_cfg$cfg.json 2>> $OUT/err.log || { tail -20 $OUT/err.log; exit 1; }
  python -c "import json;d=json.load(open('$OUT/bench_cfg$cfg.json'));print('cfg $cfg', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['roofline']['traffic'], d['timing']['placements_us'], d['cpu_baseline']['gpu_parity'])"
done
MISPMM_NO_HINT=1 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-live-traffic > $OUT/bench_general_entry.json 2>> $OUT/err.log && python -c "import json;d=json.load(open('$OUT/bench_general_entry.json'));print('general entry', round(d['ms_per_step']*1e3,4), d['roofline']['frac'], d['timing']['placements_us'])"
for acc in reference fast; do
  timeout -k 10 600 python bench.py --matrix GL7d25 --acc $acc --steps 20 --warmup 5 --no-live-traffic > $OUT/bench_GL7d25_$acc.json 2>> $OUT/err.log && python -c "import json;d=json.load(open('$OUT/bench_GL7d25_$acc.json'));print('GL7d25 $acc', round(d['ms_per_step']*1e3,4), d['roofline']['frac'])"
done
bash tools/profile_r3.sh > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
echo done
