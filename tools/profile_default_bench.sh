#!/bin/bash
# rocprofv3 --kernel-trace --stats of EXACTLY the default bench command (graph replay), summary only.
set -u
OUT=gpurun_out/prof/${1:-r1_bench_default}
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/trace.err"
for f in $(find "$OUT/trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/kernel_stats.csv"; done
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "row_gather" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
pitch = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rows, rows[1:])]
tail = slice(len(rows) // 2, None)   # timed region = the later dispatches
with open(out + "/kernel_trace_digest.txt", "w") as f:
    f.write(f"row_gather_kernel dispatches {len(rows)}\n")
    f.write(f"duration ns: mean {st.mean(dur[tail]):.0f} median {st.median(dur[tail]):.0f} min {min(dur)} max {max(dur)}\n")
    f.write(f"gap to previous end ns: median {st.median(gap[tail]):.0f}\n")
    f.write(f"start-to-start pitch ns: median {st.median(pitch[tail]):.0f} mean {st.mean(pitch[tail]):.0f}\n")
print(open(out + "/kernel_trace_digest.txt").read())
PY
cat "$OUT/bench.json" | tail -1
rm -rf "$OUT/trace"
