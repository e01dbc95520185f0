#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh.

usage: tools/make_traffic_json.py <collect dir> <round tag> [out=profiles/traffic.json]
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE reads half of the true bytes on gfx950: profiles/r01_fetch_size_calibration.md), averaged over
the strip-kernel launches of each pass and expressed per image pair.  The file names the sha256 of the ssim_kernels.hip in the tree -- the source the passes ran:
bench.py quotes the figures only for a library compiled from exactly that source (rmgr_ssim_hip_get_kernel_source_id).
"""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, R = sys.argv[1], sys.argv[2]
OUT = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "traffic.json")


def ctr(tag, name):
    rows = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/%s/t_counter_collection.csv" % (P, tag))) if "ssim_strip" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(rows) / len(rows)


def entry(tag, pairs, alg):
    f, w = ctr(tag + "_fetch", "FETCH_SIZE"), ctr(tag + "_write", "WRITE_SIZE")
    b = (2 * f + w) * 1024 / pairs
    return {"pairs": pairs, "fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "bytes_per_pair": round(b, 1), "algorithmic_bytes_per_pair": alg, "ratio": round(b / alg, 3)}


t = {"kernel_source_sha256": hashlib.sha256(open(os.path.join(ROOT, "ssim_amd", "csrc", "ssim_kernels.hip"), "rb").read()).hexdigest(),
     "_comment": "kernel_source_sha256 = the ssim_kernels.hip these passes ran (bench.py quotes the figures for that version only: rmgr_ssim_hip_get_kernel_source_id). "
                 "HBM bytes per launch measured with rocprofv3 --pmc (separate passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled per profiles/r01_fetch_size_calibration.md), "
                 "expressed per image pair; the 4096^2 entry is the headline batch itself (32 pairs) and the 1080p entry configs[3]'s per-GPU share itself (128 pairs): nothing scaled (round 5 measured 32 pairs and scaled -- a different plan); exact_1080p_32pairs_nomap: 32 pairs, for the record. Sources: profiles/%s_final_exact_4k_pmc.md, "
                 "%s_final_exact_8k_map_pmc.md, %s_final_exact_1080p_pmc.md" % (R, R, R),
     "exact_4096_nomap": entry("4k", 32, 2 * 4096 * 4096), "exact_8192_map": entry("8kmap", 2, 6 * 8192 * 8192), "exact_1080p_nomap": entry("1080p", 128, 2 * 1920 * 1080)}
if os.path.isdir("%s/1080p32_fetch" % P):
    t["exact_1080p_32pairs_nomap"] = entry("1080p32", 32, 2 * 1920 * 1080)
json.dump(t, open(OUT, "w"), indent=1)
print(json.dumps({k: v["ratio"] for k, v in t.items() if isinstance(v, dict)}))
