#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

usage: tools/summarize_prof.py <label> <kernel_trace.csv> [<counter_collection.csv> ...] > profiles/<label>.md
Groups dispatches of our kernels by (kernel, grid) -- and by duration where one grid serves different workloads -- so that batch launches and
single-pair launches are not averaged together, and averages each PMC counter per launch of the strip kernel.
"""
import collections
import csv
import re
import sys


def main():
    label, trace = sys.argv[1], sys.argv[2]
    print("# %s\n" % label)
    rows = list(csv.DictReader(open(trace)))
    groups = collections.OrderedDict()
    for r in rows:
        name = r["Kernel_Name"]
        if "ssim_hip" not in name:
            continue
        m = re.search(r"(ssim_\w+_kernel(?:<[^>]*>)?)", name)
        short = m.group(1) if m else name[:60]
        grid = "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
        key = (short, grid, r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
        groups.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("## kernel trace: %s\n" % trace.split("gpurun_out/")[-1])
    print("| kernel | grid (threads) | VGPRs | LDS B | launches | avg us | min us | max us |")
    print("|---|---|---|---|---|---|---|---|")
    for (short, grid, vg, lds), d in groups.items():
        # The balanced form of the strip kernel is launched with one wavefront per wave slot whatever the batch, so launches of different
        # workloads share a grid: split a group wherever its sorted durations jump by more than 2.5x (a 32-pair batch against a single pair).
        d = sorted(d)
        clusters, cur = [], [d[0]]
        for x in d[1:]:
            if x > 2.5 * cur[-1]:
                clusters.append(cur)
                cur = []
            cur.append(x)
        clusters.append(cur)
        for c in clusters:
            g = grid if len(clusters) == 1 else "%s (the launches of %.2f-%.2f ms)" % (grid, min(c) / 1e6, max(c) / 1e6)
            print("| %s | %s | %s | %s | %d | %.1f | %.1f | %.1f |" % (short, g, vg, lds, len(c), sum(c) / len(c) / 1e3, min(c) / 1e3, max(c) / 1e3))
    others = collections.Counter()
    for r in rows:
        if "ssim_hip" not in r["Kernel_Name"]:
            others[r["Kernel_Name"][:70]] += 1
    if others:
        print("\nother kernels in the trace (synthetic-input generation etc.): %d dispatches of %d distinct kernels" % (sum(others.values()), len(others)))
    for path in sys.argv[3:]:
        crow = list(csv.DictReader(open(path)))
        agg = collections.OrderedDict()
        for r in crow:
            if re.search(r"ssim_strip\d?_kernel", r["Kernel_Name"]):
                agg.setdefault((r["Counter_Name"], r["Grid_Size"]), []).append(float(r["Counter_Value"]))
        print("\n## counters: %s (per launch of the strip kernel)\n" % path.split("gpurun_out/")[-1])
        print("| counter | grid | launches | mean value |")
        print("|---|---|---|---|")
        for (c, g), v in agg.items():
            print("| %s | %s | %d | %.1f |" % (c, g, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
