#!/bin/bash
# rocprofv3 profiles of bench.py on the GPU box, summaries only, into <outdir> (scratch: copy what is to be kept into
# profiles/r4/).  Replaces tools/profile_r2.sh / profile_r3.sh / profile_r3_configs.sh (in the history).
#   tools/profile_r4.sh trace <outdir>   --kernel-trace --stats of every configuration: the driver's command (graph replay) and
#                                         an EAGER run of the same steps; digest per configuration (dispatches, duration, pitch)
#   tools/profile_r4.sh pmc <outdir>     PMC passes, each counter group in a run of its own with nothing but --pmc (gpurun refuses
#                                         --pmc together with trace domains), eager launches, no extras; resident AND HBM-streamed
#                                         (--operand-sets 0) loops; pmc_summary.txt + traffic.json
# The program behind `--` is python3 itself (never a wrapper: the profiler's preloaded library has initialised the GPU).
set -u
MODE=$1
OUT=$2
mkdir -p "$OUT"
export TMPDIR=/tmp
# the footprint rule instead of the plan-order autotune: under the profiler (serialised launches) the two candidates time
# differently and the profiled run could take another kernel than the bench line it is evidence for
export MISPMM_AUTOTUNE=0
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --placements 1 --no-live-traffic"

digest() {  # <trace dir> <digest file> <header>
python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, sys, statistics as st
d, out, header = sys.argv[1:4]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append(r)
with open(out, "a") as f:
    f.write("# " + header + "\n")
    for k, rs in sorted(by.items(), key=lambda kv: -len(kv[1]))[:3]:
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        tail = slice(len(rs) // 4, None)
        line = f"{k[:170]}\n  dispatches {len(rs)}  duration ns: mean {st.mean(dur[tail]):.0f} median {st.median(dur[tail]):.0f} min {min(dur)} max {max(dur)}"
        if len(rs) > 8:
            p = sorted(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rs[len(rs) // 4:], rs[len(rs) // 4 + 1:]))
            g = sorted(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs[len(rs) // 4:], rs[len(rs) // 4 + 1:]))
            line += f"\n  start-to-start pitch ns: median {st.median(p):.0f} p10 {p[len(p) // 10]}   gap to previous end ns: median {st.median(g):.0f}"
        line += f"\n  grid {rs[0].get('Grid_Size')} workgroup {rs[0].get('Workgroup_Size')} vgpr {rs[0].get('VGPR_Count')} sgpr {rs[0].get('SGPR_Count')} lds {rs[0].get('LDS_Block_Size')}\n"
        f.write(line)
PY
}

if [ "$MODE" = trace ]; then
  for spec in "headline:--config headline" "cfg2:--config 2" "cfg3:--config 3" "cfg4:--config 4" "cfg5:--config 5"; do
    tag=${spec%%:*}; args=${spec#*:}
    D="$OUT/bench_kernel_trace_digest_$tag.txt"; : > "$D"
    for launch in graph eager; do
      rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -- python3 bench.py --gpus 1 $COMMON $args --launch $launch \
          > "$OUT/bench_under_rocprof_${tag}_$launch.json" 2> "$OUT/trace_$tag.err" || { tail -5 "$OUT/trace_$tag.err"; exit 1; }
      if [ $tag = headline ] && [ $launch = graph ]; then for f in $(find "$OUT/trace_$tag" -name '*kernel_stats.csv'); do cp "$f" "$OUT/bench_kernel_stats.csv"; done; fi
      digest "$OUT/trace_$tag" "$D" "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 $COMMON $args --launch $launch   (bench.py's own figure under the profiler: $(python3 -c "import json;d=json.load(open('$OUT/bench_under_rocprof_${tag}_$launch.json'));print(round(d['ms_per_step']*1e3,3),'us per step')"))"
      rm -rf "$OUT/trace_$tag"
    done
    cat "$D"
  done
  exit 0
fi

pmc_pass() {  # <dir tag> <bench args> -- <counters...>
  local tag=$1; shift
  local args=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 bench.py $args --launch eager > "$OUT/$tag.json" 2> "$OUT/$tag.err" || echo "pmc pass $tag failed" | tee -a "$OUT/pmc_errors.txt"
}
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); pmc_pass "pmc_headline_$i" "--config headline $COMMON" $PMC
done
for c in headline 2 3 4 5; do
  [ $c = headline ] || pmc_pass "pmc_cfg${c}_1" "--config $c $COMMON" FETCH_SIZE
  [ $c = headline ] || pmc_pass "pmc_cfg${c}_2" "--config $c $COMMON" WRITE_SIZE TCC_REQ_sum
  pmc_pass "pmc_stream${c}_1" "--config $c $COMMON --operand-sets 0" FETCH_SIZE
  pmc_pass "pmc_stream${c}_2" "--config $c $COMMON --operand-sets 0" WRITE_SIZE TCC_REQ_sum
done
export MISPMM_NO_HINT=1
pmc_pass "pmc_nohint_1" "--config headline $COMMON" FETCH_SIZE
pmc_pass "pmc_nohint_2" "--config headline $COMMON" WRITE_SIZE TCC_REQ_sum
unset MISPMM_NO_HINT
pmc_pass "pmc_cfg4_3" "--config 4 $COMMON" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA

python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
summary, traffic = {}, {"_comment": "L2<->fabric bytes per launch from rocprofv3 PMC passes of `bench.py --config C --steps 20 --warmup 5 "
                                    "--no-extras --launch eager [--operand-sets 0]` (tools/profile_r4.sh pmc), corrected as MI355X_MICROARCH.md "
                                    "prescribes: FETCH_SIZE is in KiB and reads exactly 1/2 of wide coalesced reads on gfx950, WRITE_SIZE (KiB) is "
                                    "exact.  The counters sit on the L2's fabric side: Infinity-Cache hits are included, so this is an upper "
                                    "bound on HBM traffic.  Keys ending in /streamed: the loop rotates over 512 MiB of (B, C) pairs.  `kernel_tag` is "
                                    "what mispmm_last_kernel() reported in the profiled run; bench.py prints `traffic` only when the tag of its own run matches."}
GROUPS = {"headline": "pmc_headline", "2": "pmc_cfg2", "3": "pmc_cfg3", "4": "pmc_cfg4", "5": "pmc_cfg5",
          "headline streamed": "pmc_streamheadline", "2 streamed": "pmc_stream2", "3 streamed": "pmc_stream3", "4 streamed": "pmc_stream4",
          "5 streamed": "pmc_stream5", "headline, general entry (MISPMM_NO_HINT=1)": "pmc_nohint"}
for cfg, prefix in GROUPS.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    tag = key = None
    for d in sorted(glob.glob(f"{out}/{prefix}_*")):
        if not os.path.isdir(d):
            continue
        try:
            line = json.loads(open(d + ".json").read().strip().splitlines()[-1])
            tag, key = line["config"]["kernel_tag"], line
        except Exception:
            continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc or key is None:
        continue
    name = max(acc, key=lambda k: max(len(v) for v in acc[k].values()))        # the workload's kernel = the one with the most dispatches
    mean = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in acc[name].items()}
    summary[cfg] = {"kernel": name[:200], "kernel_tag": tag, "dispatches": {c: len(v) for c, v in acc[name].items()}, "mean_per_dispatch": mean}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        w = key["config"]["workload"]
        matrix, n, accm = w.split(" ")[0], key["metric"].split("K=")[1].split(" ")[0], key["config"]["acc_mode"]
        accm = accm if accm in ("reference", "fast") else "reference"
        fetch, write = int(mean["FETCH_SIZE"] * 1024 * 2), int(mean["WRITE_SIZE"] * 1024)
        tkey = f"{key['config']['baseline_config']}:{matrix}/{n}/{accm}" + ("/nohint" if "general entry" in cfg else "") + ("/streamed" if "streamed" in cfg else "")
        traffic[tkey] = {"fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write, "kernel_tag": tag, "kernel": name[:160],
                         "algorithmic_bytes": key["roofline"]["algorithmic_bytes_per_launch"], "source": "profiles/r4/pmc_summary.txt"}
with open(out + "/pmc_summary.txt", "w") as f:
    for cfg, sm in summary.items():
        f.write(f"== bench.py --config {cfg}: {sm['kernel']}\n   kernel_tag {sm['kernel_tag']}\n")
        for c, v in sorted(sm["mean_per_dispatch"].items()):
            f.write(f"  {c}: mean_per_dispatch {v:.1f}  (n={sm['dispatches'][c]}, first quarter dropped)\n")
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
print(open(out + "/pmc_summary.txt").read())
print(json.dumps(traffic, indent=1))
PY
rm -rf "$OUT"/pmc_*/
