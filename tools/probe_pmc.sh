#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the forced-occupancy probe's launches (tools/probe_stability.py under rocprofv3 --pmc), to tell a degraded burst's cause apart:
# waves resident for the whole launch but issuing slowly (wave-cycles grow with the launch time) or waves running one after the other (wave-cycles per wave unchanged).
# usage: tools/probe_pmc.sh <out-subdir-of-gpurun_out>
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-probe_pmc}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc" -o t -- python3 tools/probe_stability.py 4 > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tr = glob.glob(out + "/pmc/**/t_kernel_trace.csv", recursive=True)
cc = glob.glob(out + "/pmc/**/t_counter_collection.csv", recursive=True)
if not tr or not cc:
    print("no csv"); sys.exit(0)
dur = {}
name = {}
for r in csv.DictReader(open(tr[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    name[r["Dispatch_Id"]] = r["Kernel_Name"]
ctr = collections.defaultdict(dict)
for r in csv.DictReader(open(cc[0])):
    ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for d, c in ctr.items():
    n = name.get(d, "")
    if "probe_valu_kernel<2" not in n and "probe_valu_kernelILi2" not in n:
        continue
    rows.append((dur.get(d, 0.0), c))
rows.sort(key=lambda x: x[0])
print("two-wave probe launches: %d; duration ms min %.3f median %.3f max %.3f" % (len(rows), rows[0][0], rows[len(rows) // 2][0], rows[-1][0]))
def mean(sel, key):
    v = [c.get(key, 0.0) for _, c in sel]
    return sum(v) / max(len(v), 1)
fast = [r for r in rows if r[0] < rows[0][0] * 1.08]
slow = [r for r in rows if r[0] > rows[0][0] * 1.2]
for label, sel in (("fast (within 8 % of the best)", fast), ("slow (> 20 % above the best)", slow)):
    if not sel:
        print(label, ": none"); continue
    print("%s: %d launches, mean %.3f ms;  %s" % (label, len(sel), sum(r[0] for r in sel) / len(sel),
          "  ".join("%s %.4g" % (k, mean(sel, k)) for k in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"))))
PY
tail -6 "$OUT/run.log"
