#!/bin/bash
# rocprofv3 --kernel-trace --stats of every other bench configuration (the headline's is made by tools/profile_r3.sh):
# one digest per configuration into gpurun_out/prof_r3/ -- which kernel ran, how often, its duration under the profiler.
set -u
OUT=gpurun_out/prof_r3
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --placements 1"
run() {  # <tag> <bench args>
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -- python3 bench.py --gpus 1 $COMMON "$@" > "$OUT/bench_under_rocprof_$tag.json" 2> "$OUT/trace_$tag.err"
  python3 - "$OUT" "$tag" "$*" <<'PY'
import csv, glob, sys, statistics as st
out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
rows = []
for f in glob.glob(f"{out}/trace_{tag}/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append(r)
with open(f"{out}/bench_kernel_trace_digest_{tag}.txt", "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --placements 1 {args}\n")
    for k, rs in sorted(by.items(), key=lambda kv: -len(kv[1]))[:4]:
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        tail = slice(len(rs) // 4, None)
        f.write(f"{k[:170]}\n  dispatches {len(rs)}  duration ns: mean {st.mean(dur[tail]):.0f} median {st.median(dur[tail]):.0f} min {min(dur)} max {max(dur)}\n")
print(open(f"{out}/bench_kernel_trace_digest_{tag}.txt").read())
PY
  rm -rf "$OUT/trace_$tag"
}
run cfg2 --config 2
run cfg3 --config 3
run cfg4 --config 4
run cfg5 --config 5
run GL7d25 --matrix GL7d25
echo done
