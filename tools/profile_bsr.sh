#!/bin/bash
# rocprofv3 on the bf16 BSR kernel: kernel trace + MFMA / traffic counters -> gpurun_out/prof/<tag>/
set -u
TAG=${1:-r1_bsr_bf16}
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_INSTS_VALU|SQ_ACTIVE_INST_VALU|SQ_INSTS_VMEM_RD|TA_BUSY_avr|TA_TA_BUSY_sum)\b" | sort -u > "$OUT/available_counters.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/run_bsr_bf16.py 200 > "$OUT/run.log" 2> "$OUT/trace.err"
for f in $(find "$OUT/trace" -name '*kernel_stats.csv'); do cp "$f" "$OUT/kernel_stats.csv"; done
i=0
for PMC in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 tools/run_bsr_bf16.py 20 > /dev/null 2> "$OUT/pmc$i.err" || echo "pmc pass $i ($PMC) failed" >> "$OUT/pmc_errors.txt"
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
        if "bsr" not in k: continue
        f.write(k[:120] + "\n")
        for c, v in sorted(cs.items()):
            v2 = v[len(v)//2:]
            f.write(f"  {c}: mean_per_dispatch {sum(v2)/len(v2):.1f}  (n={len(v2)})\n")
print(open(out + "/pmc_summary.txt").read())
PY
grep bsr "$OUT/kernel_stats.csv"
cat "$OUT/available_counters.txt" | tr '\n' ' '
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]*/
