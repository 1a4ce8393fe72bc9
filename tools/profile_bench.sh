#!/bin/bash
# Profile bench.py on the GPU box with rocprofv3 and leave small summaries under gpurun_out/prof/.
#   tools/profile_bench.sh <tag> [bench.py args...]
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2..: PMC counters, each in its own
# run with nothing but --pmc (gpurun refuses --pmc combined with trace domains).
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 300 --warmup 50 --no-cpu-baseline --launch eager $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
for f in $(find "$OUT/trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/kernel_stats.csv"; done
# per-dispatch trace is large: keep a digest (count, mean/min/max duration, mean gap) per kernel
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = {}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = by.setdefault(r["Kernel_Name"], {"dur": [], "gap": [], "vgpr": r.get("VGPR_Count"), "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
    d["dur"].append(e - s)
    if prev_end is not None:
        d["gap"].append(s - prev_end)
    prev_end = e
with open(out + "/kernel_trace_digest.txt", "w") as f:
    for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1]["dur"])):
        dur = d["dur"]; gap = d["gap"] or [0]
        f.write(f"{k[:110]}\n  calls {len(dur)}  dur_ns mean {st.mean(dur):.0f} median {st.median(dur):.0f} min {min(dur)} max {max(dur)}"
                f"  gap_before_ns median {st.median(gap):.0f}  grid {d['grid']} wg {d['wg']} vgpr {d['vgpr']}\n")
print(open(out + "/kernel_trace_digest.txt").read())
PY
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager $* > /dev/null 2> "$OUT/pmc$i.err" || echo "pmc pass $i failed" >> "$OUT/pmc_errors.txt"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as f:
    for k, cs in acc.items():
        f.write(k[:110] + "\n")
        for c, v in sorted(cs.items()):
            v2 = v[len(v)//2:]   # later dispatches = steady state
            f.write(f"  {c}: mean_per_dispatch {sum(v2)/len(v2):.1f}  (n={len(v2)})\n")
print(open(out + "/pmc_summary.txt").read())
PY
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]*/
