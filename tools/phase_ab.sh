#!/bin/bash
# Runs ON THE GPU BOX: the balanced schedule's chunk list with T images interleaved (tuning variant 100 + T; 7 = the plain list of round 5, 6 = plan()'s
# choice) against the strips (3) and the default (0) -- kernel time interleaved per shape (tools/ab.py), and the L2 -> fabric read traffic (FETCH_SIZE)
# of every variant in ONE rocprofv3 pass per shape (tools/profile_target.py takes a list).     usage: tools/phase_ab.sh <out-subdir-of-gpurun_out> [mode=0]
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-phase_ab}; MODE=${2:-0}
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python3 tools/balanced_check.py > "$OUT/balanced_check.txt" 2>&1
TS="1 2 3 4 5 6 8 9 12 14 16 19"
VARS="0,3,7,6"; for T in $TS; do VARS="$VARS,$((100 + T))"; done
{
for P in 24 32 40 48 64 96 128; do timeout 300 python3 tools/ab.py $P 1920 $MODE 0 0,3,7,6 5 0 1080; done
for S in "3 4096 4096" "12 3840 2160" "16 5120 2880" "8 3840 2160" "48 1280 720"; do set -- $S; timeout 300 python3 tools/ab.py $1 $2 $MODE 0 0,3,7,6 5 0 $3; done
for P in 32 64 128; do timeout 300 python3 tools/ab.py $P 1920 $MODE 0 $VARS 3 0 1080; done
} > "$OUT/ab.txt" 2>&1
for S in "128 1920 1080" "96 1920 1080" "64 1920 1080" "48 1920 1080" "32 1920 1080" "24 1920 1080" "12 3840 2160" "16 5120 2880"; do
  set -- $S
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$1x$2x$3" -o t -- python3 tools/profile_target.py $1 3 $MODE 0 $2 $3 0 $VARS > "$OUT/fetch_$1x$2x$3.log" 2>&1
done
python3 - "$OUT" "$VARS" <<'PY' > "$OUT/fetch_summary.txt"
import csv, sys, glob, os
out, variants = sys.argv[1], sys.argv[2].split(",")
for d in sorted(glob.glob(out + "/fetch_*x*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/t_counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    rows = [(int(r["Dispatch_Id"]), float(r["Counter_Value"])) for r in csv.DictReader(open(f[0])) if "ssim_strip" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort()
    pairs, w, h = [int(x) for x in os.path.basename(d)[6:].split("x")]
    alg = 2.0 * w * h * pairs
    per = len(rows) // len(variants)
    line = []
    for i, v in enumerate(variants):
        vals = [x for _, x in rows[i * per:(i + 1) * per]]
        kib = sum(vals) / max(len(vals), 1)
        line.append("%s:%.3f" % (v, 2 * kib * 1024 / alg))
    print("%-22s fetch ratio (2 x FETCH_SIZE / algorithmic) per variant, %d launches each:  %s" % (os.path.basename(d)[6:], per, "  ".join(line)))
PY
cat "$OUT/fetch_summary.txt"; tail -3 "$OUT/balanced_check.txt"
