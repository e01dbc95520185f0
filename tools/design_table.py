#!/usr/bin/env python3
"""Rewrites DESIGN.md's "Measured (round 6 ...)" table from the committed bench lines (profiles/r06_final_bench.json -- the MEDIAN box of the round's
collections, tools/pick_median_box.py --, the rocprofv3 trace summary and profiles/traffic.json), so that the table is whatever the last
tools/collect_profiles.sh + tools/publish_profiles.sh + tools/pick_median_box.py produced.
usage: python3 tools/design_table.py [--check]     (--check: exit 1 if DESIGN.md would change)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, "r06_final_bench.json")))
traffic = json.load(open(os.path.join(P, "traffic.json")))
cfg, sp, cpu = d["configs"], d["single_pair"], d["cpu_baseline"]
pc = cpu["per_config"]


def k(v):          # Mpix/s -> "207.1 k"
    return "%.1f k" % (v / 1e3)


def short(name):   # ssim_strip2_kernel<0, 0, true, true> -> <0, 0, true, true>
    return "`" + name[name.index("<"):] + "`"


def trace_row(kernel, near_ms):
    """(launches, average us) of the trace summary's row for `kernel` whose average is nearest to near_ms (one kernel + grid can serve several workloads)."""
    best = (None, None)
    for line in open(os.path.join(P, "r06_final_bench_kernel_trace.md")):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 7 and c[1] == kernel:
            n, avg = int(c[5]), float(c[6])
            if best[1] is None or abs(avg / 1e3 - near_ms) < abs(best[1] / 1e3 - near_ms):
                best = (n, avg)
    return best


head = d["roofline"]
launches, avg_us = trace_row(head["kernel"], head["kernel_avg_ms"])
cold = sp["cold_start"]
rows = []
v = d["valu"]
rows.append("| 32 × 4096² (`value` **%s**, %.3f ms per step; sustained 6 s: %s) | exact | `%s` | %.3f ms | %s | %.0f (%.2f %%); measured HBM traffic %.3f GB per launch = %.3f× algorithmic | %.1f T (%.1f %%; **%.1f %% of this box's %.1f T two-wave peak**, %.1f %% of its %.1f T eight-wave peak, both probed in the same run) | %s best / %s median (4096², one pair); 1 thread %.0f |" % (
    k(d["value"]), d["ms_per_step"], k(d["sustained"]["mpix_s"]), head["kernel"], head["kernel_avg_ms"],
    k(32 * 4096 * 4096 / head["kernel_avg_ms"] / 1e3), head["achieved"], head["frac"] * 100, head["traffic"] / 1e9, head["traffic"] / head["algorithmic_bytes_per_launch"],
    v["achieved"], v["frac"] * 100, v["frac_of_box_peak_at_kernel_occupancy"] * 100, v["box_peak_2wave"], v["frac_of_box_peak"] * 100, v["box_peak_8wave"],
    k(cpu["value"]), k(cpu["median"]), cpu["one_thread_mpix_s"]))
f, s = d["fast_mode"], d["separable_mode"]
rows.append("| 32 × 4096² | fast = hybrid / separable | %s / %s | %.3f / **%.3f ms** | %s / **%s** | %.0f / %.0f (%.1f %%) | %.1f T (%.1f %% of the box's two-wave peak) / %.1f T (%.1f %% of 78.6 T; %.1f %% of the box's three-wave peak) | |" % (
    short(f["kernel"]), short(s["kernel"]), f["kernel_avg_ms"], s["kernel_avg_ms"], k(f["mpix_s"]), k(s["mpix_s"]),
    f["roofline_frac"] * 8000, s["roofline_frac"] * 8000, s["roofline_frac"] * 100, f["valu_frac"] * 78.6, f["valu_frac_of_box_peak_at_kernel_occupancy"] * 100,
    s["valu_frac"] * 78.6, s["valu_frac"] * 100, s["valu_frac_of_box_peak_at_kernel_occupancy"] * 100))
x1 = cfg["4k x1 exact"]
rows.append("| 1 × 4096² (configs[1] literally) | exact (EARLY) | %s | %.4f ms | %s (enqueued back to back %s; blocking call %.3f ms = %s) | %.0f (%.1f %%) | %.1f T | %s |" % (
    short(x1["kernel"]), x1["kernel_avg_ms"], k(x1["mpix_s"]), k(sp["enqueued_mpix_s"]), sp["blocking_call_ms"], k(sp["blocking_call_mpix_s"]),
    x1["roofline"]["achieved"], x1["roofline"]["frac"] * 100, x1["valu"]["achieved"], k(pc["4k"]["threads_all_mpix_s"])))
e8, f8, s8 = cfg["8k-map exact"], cfg["8k-map fast"], cfg["8k-map separable"]
rows.append("| 2 × 8192² + map (configs[2]) | exact / fast / separable | %s / %s / %s | %.3f / %.3f / **%.3f ms** | %s / %s / **%s** | %.0f (%.1f %%) / %.0f (%.1f %%) / **%.0f (%.1f %%)**; traffic %.3f× | %.1f T (%.1f %% of the box's peak at its occupancy) / %.1f T (%.1f %%) / %.1f T (%.1f %%) | %s (8192² + map) |" % (
    short(e8["kernel"]), short(f8["kernel"]), short(s8["kernel"]), e8["kernel_avg_ms"], f8["kernel_avg_ms"], s8["kernel_avg_ms"], k(e8["mpix_s"]), k(f8["mpix_s"]), k(s8["mpix_s"]),
    e8["roofline"]["achieved"], e8["roofline"]["frac"] * 100, f8["roofline"]["achieved"], f8["roofline"]["frac"] * 100, s8["roofline"]["achieved"], s8["roofline"]["frac"] * 100,
    traffic["exact_8192_map"]["ratio"], e8["valu"]["achieved"], e8["valu"]["frac_of_box_peak_at_kernel_occupancy"] * 100, f8["valu"]["achieved"], f8["valu"]["frac_of_box_peak_at_kernel_occupancy"] * 100,
    s8["valu"]["achieved"], s8["valu"]["frac_of_box_peak_at_kernel_occupancy"] * 100, k(pc["8k-map"]["threads_all_mpix_s"])))
e1, f1, s1 = cfg["1080p x128 exact"], cfg["1080p x128 fast"], cfg["1080p x128 separable"]
bal = lambda c: "**balanced schedule**" if c["kernel"].endswith("true>") else "strips"
rows.append("| 128 × 1080p (configs[3] per-GPU share) | exact (%s) / fast (%s) / separable (%s) | %s / %s / %s | %.3f / %.3f / %.3f ms | **%s** / %s / %s | %.0f (%.1f %%) / %.0f / %.0f; traffic %.3f× (round 5: 3.03×) | %.1f T (%.1f %% of the box's two-wave peak) / %.1f T / %.1f T | **%s** as BASELINE.md §3 defines it (pairs looped serially, OpenMP inside each call; %d of 1024 timed) |" % (
    bal(e1), bal(f1), bal(s1), short(e1["kernel"]), short(f1["kernel"]), short(s1["kernel"]), e1["kernel_avg_ms"], f1["kernel_avg_ms"], s1["kernel_avg_ms"], k(e1["mpix_s"]), k(f1["mpix_s"]), k(s1["mpix_s"]),
    e1["roofline"]["achieved"], e1["roofline"]["frac"] * 100, f1["roofline"]["achieved"], s1["roofline"]["achieved"], traffic["exact_1080p_nomap"]["ratio"],
    e1["valu"]["achieved"], e1["valu"]["frac_of_box_peak_at_kernel_occupancy"] * 100, f1["valu"]["achieved"], s1["valu"]["achieved"], k(pc["1080p-batch"]["threads_all_mpix_s"]), pc["1080p-batch"]["pairs_timed"]))
dd = cfg["4k double + map"]
rows.append("| 4 × 4096² + map, fp64 internals (configs[4]) | double | `%s` | %.3f ms | %s | %.0f (%.1f %%) | %.1f T issue slots (%.1f %% of 39.3 T); fp64 arithmetic alone %.1f T (%.1f %%) | %s: the reference's `RMGR_SSIM_USE_DOUBLE` build with the map (%s without) |" % (
    dd["kernel"], dd["kernel_avg_ms"], k(dd["mpix_s"]), dd["roofline"]["achieved"], dd["roofline"]["frac"] * 100, dd["valu"]["achieved"], dd["valu"]["frac"] * 100,
    dd["valu"]["fp64_math_achieved"], dd["valu"]["fp64_math_frac"] * 100, k(pc["4k-double"]["with_map_threads_all_mpix_s"]), k(pc["4k-double"]["threads_all_mpix_s"])))
rows.append("| unchanged host-pointer call, 4096² (PCIe included) | exact | | %.3f ms / %.2f ms with map | %s / %s | | | |" % (
    sp["host_pointer_call_ms"], sp["host_pointer_call_with_map_ms"], k(sp["host_pointer_call_mpix_s"]), k(sp["host_pointer_call_with_map_mpix_s"])))
rows.append("| first call of a fresh process (1080p pair) | | | best plain run %.0f ms (104…270 across runs and boxes of rounds 5-6); the instrumented run: `hipInit` %.0f, first queue %.0f, `dlopen` %.1f, code object **%.1f**, first launch %.1f | | | | |" % (
    cold["plain"]["total_ms"], cold["split"]["runtime_init_ms"], cold["split"]["context_ms"], cold["split"]["dlopen_ms"], cold["split"]["code_object_ms"], cold["split"]["first_ssim_ms"]))

pr = d.get("plan_regret", {})
regret = "; ".join("%s %.3f" % (key, pr[key]["regret"]) for key in ("headline", "1080p x128", "8k-map x2", "4k x1") if key in pr)
intro = ("`profiles/r06_final_*`: `tools/collect_profiles.sh` on one box after the last kernel change — the PMC passes first, `profiles/traffic.json`\n"
         "regenerated from them on the box (stamped with the sha256 of the kernel source they ran), then the bench lines (which therefore carry\n"
         "`roofline.traffic`), the same command under `rocprofv3 --kernel-trace --stats` (`%s`: %.3f ms average over %d\n"
         "launches against %.3f ms from the published line's HIP events), the exchange / host / latency probes, cold start, concurrent callers, the 30 000-case soak, the\n"
         "full-size and balanced-schedule checks.  **The published line `profiles/r06_final_bench.json` is the MEDIAN box** by `value` of the round's bench runs of the\n"
         "final library (`profiles/r06_box_spread.md`; round 5 published its fastest of seven), and every VALU fraction in it divides by the SAME box's packed-fp32\n"
         "peak, probed in-process before the clock-settle loop and right after the timed steps (`rmgr_ssim_hip_probe_valu`, the better sample; rounds 4-5 divided by a constant from\n"
         "another box).  The timed launches ran at %.0f MHz (slowest XCD %.0f; `rmgr_ssim_hip_get_profile_clock`: one workgroup per XCD counts shader cycles per reference tick), the probe at %.0f MHz:\n"
         "the SSIM kernel draws more power than a pure FMA stream and is clocked lower for it; per CLOCK the kernel issues %.1f %% of the 32768 lane-operations the chip can, the probe %.1f %% --\n"
         "**the kernel runs at %.1f %% of the probe's per-clock rate**.\n"
         "`plan_regret` of that line (default plan / best candidate of `rmgr_ssim_hip_tune`, same run): %s. This table is generated: `tools/design_table.py`.\n\n"
         % (head["kernel"], avg_us / 1e3, launches, head["kernel_avg_ms"], v["shader_mhz_during_timed_launches"], v["slowest_xcd_mhz_during_timed_launches"], v["shader_mhz_during_probe"],
            v["frac_of_issue_peak_per_clock"] * 100, v["probe_frac_of_issue_peak_per_clock"] * 100, v["frac_of_box_peak_per_clock"] * 100, regret))
table = (intro + "| config | mode | kernel | kernel time | Mpix/s | algorithmic GB/s (% of 8 TB/s) | VALU lane-ops/s (% of the 78.6 T data sheet; % of the box's own probed peak) | CPU beside it (64 pinned threads, best run) |\n"
         "|---|---|---|---|---|---|---|---|\n" + "\n".join(rows) + "\n\n")

path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
m = re.search(r"(### Measured \(round 6[^\n]*\n)(.*?)(### Where each mode sits)", text, re.S)
assert m, "DESIGN.md: the Measured (round 6) section was not found"
new = text[:m.start(2)] + table + text[m.start(3):]
if "--check" in sys.argv:
    sys.exit(0 if new == text else 1)
open(path, "w").write(new)
print("DESIGN.md: Measured (round 6) table rewritten from profiles/r06_final_bench.json")
