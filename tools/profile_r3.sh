#!/bin/bash
# Round-3 profiles of bench.py on the GPU box (rocprofv3), summaries only, into gpurun_out/prof_r3/ -- copy what
# is to be kept into profiles/r3/.
#   tools/profile_r3.sh
# Pass 1: --kernel-trace --stats of the driver's own command (graph replay), digest of the timed kernel.
# Pass 2..: PMC counters, each group in a run of its own with nothing but --pmc (gpurun refuses --pmc together with
# trace domains), eager launches of the same steps, no cold shots / other accumulate mode (--no-extras), so every
# dispatch of the kernel is a steady-state one; the first quarter of the dispatches is dropped anyway.
# Last: traffic.json, L2<->fabric bytes per launch keyed by workload AND by the kernel tag bench.py reported.
set -u
OUT=gpurun_out/prof_r3
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-extras --placements 1"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --gpus 1 $COMMON > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
for f in $(find "$OUT/trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/bench_kernel_stats.csv"; done
python3 - "$OUT" <<'PY'
import csv, glob, sys, statistics as st
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"], []).append(r)
with open(out + "/bench_kernel_trace_digest.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras\n")
    for k, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        pitch = [int(b["Start_Timestamp"]) - int(a["Start_Timestamp"]) for a, b in zip(rs, rs[1:])]
        gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rs, rs[1:])]
        tail = slice(len(rs) // 4, None)
        line = f"{k[:150]}\n  dispatches {len(rs)}  duration ns: mean {st.mean(dur[tail]):.0f} median {st.median(dur[tail]):.0f} min {min(dur)} max {max(dur)}"
        if len(rs) > 8:
            p = sorted(pitch[tail]); g = sorted(gap[tail])
            line += f"\n  start-to-start pitch ns: median {st.median(p):.0f} p10 {p[len(p) // 10]}   gap to previous end ns: median {st.median(g):.0f}"
        line += f"\n  grid {rs[0].get('Grid_Size')} workgroup {rs[0].get('Workgroup_Size')} vgpr {rs[0].get('VGPR_Count')} sgpr {rs[0].get('SGPR_Count')} lds {rs[0].get('LDS_Block_Size')}\n"
        f.write(line)
print(open(out + "/bench_kernel_trace_digest.txt").read())
PY
rm -rf "$OUT/trace"

pmc_pass() {  # <dir tag> <bench args> -- <counters...>
  local tag=$1; shift
  local args=$1; shift
  echo "pmc pass $tag: $*"
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$tag" -- python3 bench.py $args --launch eager > "$OUT/$tag.json" 2> "$OUT/$tag.err" || echo "pmc pass $tag failed" | tee -a "$OUT/pmc_errors.txt"
}
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1)); pmc_pass "pmc_headline_$i" "--config headline $COMMON" $PMC
done
for c in 2 3 4 5; do
  pmc_pass "pmc_cfg${c}_1" "--config $c $COMMON" FETCH_SIZE
  pmc_pass "pmc_cfg${c}_2" "--config $c $COMMON" WRITE_SIZE TCC_REQ_sum
done
# kernels without an entry in round 2: the split kernel on GL7d25 (REFERENCE and FAST) and the general CSR entry point
pmc_pass "pmc_gl7ref_1" "--config headline --matrix GL7d25 $COMMON" FETCH_SIZE
pmc_pass "pmc_gl7ref_2" "--config headline --matrix GL7d25 $COMMON" WRITE_SIZE TCC_REQ_sum
pmc_pass "pmc_gl7fast_1" "--config headline --matrix GL7d25 --acc fast $COMMON" FETCH_SIZE
pmc_pass "pmc_gl7fast_2" "--config headline --matrix GL7d25 --acc fast $COMMON" WRITE_SIZE TCC_REQ_sum
export MISPMM_NO_HINT=1
pmc_pass "pmc_nohint_1" "--config headline $COMMON" FETCH_SIZE
pmc_pass "pmc_nohint_2" "--config headline $COMMON" WRITE_SIZE TCC_REQ_sum
unset MISPMM_NO_HINT
pmc_pass "pmc_cfg4_3" "--config 4 $COMMON" SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA

python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
summary, traffic = {}, {"_comment": "L2<->fabric bytes per launch from rocprofv3 PMC passes of `bench.py --config C --steps 20 --warmup 5 "
                                    "--no-extras --launch eager` (tools/profile_r3.sh), corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE "
                                    "is in KiB and reads exactly 1/2 of wide coalesced reads on gfx950, WRITE_SIZE (KiB) is exact.  The counters "
                                    "sit on the L2's fabric side: Infinity-Cache hits are included, so this is an upper bound on HBM traffic.  "
                                    "`kernel_tag` is what mispmm_last_kernel() reported in the profiled run; bench.py prints `traffic` only "
                                    "when the tag of its own run matches."}
GROUPS = {"headline": "pmc_headline", "2": "pmc_cfg2", "3": "pmc_cfg3", "4": "pmc_cfg4", "5": "pmc_cfg5",
          "GL7d25 reference": "pmc_gl7ref", "GL7d25 fast": "pmc_gl7fast", "headline, general entry (MISPMM_NO_HINT=1)": "pmc_nohint"}
for cfg, prefix in GROUPS.items():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    tag = None
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
    if not acc:
        continue
    # the workload's kernel = the one with the most dispatches
    name = max(acc, key=lambda k: max(len(v) for v in acc[k].values()))
    mean = {c: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for c, v in acc[name].items()}
    summary[cfg] = {"kernel": name[:200], "kernel_tag": tag, "dispatches": {c: len(v) for c, v in acc[name].items()}, "mean_per_dispatch": mean}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        w = key["config"]["workload"]
        matrix, n, accm = w.split(" ")[0], key["metric"].split("K=")[1].split(" ")[0], key["config"]["acc_mode"]
        accm = accm if accm in ("reference", "fast") else "reference"
        fetch, write = int(mean["FETCH_SIZE"] * 1024 * 2), int(mean["WRITE_SIZE"] * 1024)
        bcfg = key["config"]["baseline_config"]
        tkey = f"{bcfg}:{matrix}/{n}/{accm}" + ("/nohint" if "general entry" in cfg else "")
        traffic[tkey] = {"fetch_bytes": fetch, "write_bytes": write, "total_bytes": fetch + write, "kernel_tag": tag,
                                                 "kernel": name[:160], "algorithmic_bytes": key["roofline"]["algorithmic_bytes_per_launch"],
                                                 "source": "profiles/r3/pmc_summary.txt"}
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
