#!/usr/bin/env python3
"""Publishes the MEDIAN box: among the bench lines of the final library collected on several boxes (gpurun_out/<dir>/bench_4k.json, one gpurun call = one
fresh box each), the one whose `value` is the median becomes profiles/<tag>_final_bench.json, and profiles/<tag>_box_spread.md lists them all -- `value`,
the headline kernel's time, the box's own probed VALU peaks and the kernel's fraction of them (which is what should agree between boxes), plan_regret.
Round 5 published its fastest of seven boxes; the driver's own run then read 3.5 % below the published line.

usage: tools/pick_median_box.py <tag, e.g. r06> <dir> [<dir> ...]      (dirs under gpurun_out/; lines whose kernel source id differs from the first's are refused)"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, dirs = sys.argv[1], sys.argv[2:]
    lines = []
    for d in dirs:
        path = os.path.join(ROOT, "gpurun_out", d, "bench_4k.json")
        try:
            line = json.loads(open(path).read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            print("skipped (no bench line): %s" % path)
            continue
        lines.append((d, path, line))
    assert len(lines) >= 3, "need at least three boxes for a median"
    lines.sort(key=lambda x: x[2]["value"])
    med = lines[(len(lines) - 1) // 2]                      # the lower median for an even count: never the flattering one
    shutil.copyfile(med[1], os.path.join(ROOT, "profiles", "%s_final_bench.json" % tag))
    out = ["# Round %s: `python bench.py` of the final library on %d boxes (one gpurun call each = a fresh box).  The published line, `profiles/%s_final_bench.json`, is the MEDIAN by `value`" % (tag[1:].lstrip("0"), len(lines), tag),
           "# (marked); round 5 published its fastest of seven.  `valu.frac_of_box_peak_at_kernel_occupancy` divides the kernel's lane-operations per second by the SAME box's forced-occupancy",
           "# v_pk_fma_f32 rate (rmgr_ssim_hip_probe_valu, probed in-process before the warm-up and after the timed steps): boxes differ in `value`, the fraction should not.", "",
           "| collection | `value` Mpix/s | headline kernel ms | box peak 2 waves / 8 waves (T lane-ops/s) | shader MHz under the kernel (slowest XCD) / under the probe | kernel / box two-wave peak | the same PER CLOCK | / 78.6 T data sheet | 128 x 1080p exact Mpix/s (fraction of box peak; per clock) | separable 32 x 4096^2 Mpix/s | plan_regret headline / 1080p x128 |",
           "|---|---|---|---|---|---|---|---|---|---|---|"]
    for d, _, l in lines:
        v, c = l["valu"], l.get("configs", {})
        p1080 = c.get("1080p x128 exact", {})
        pr = l.get("plan_regret", {})
        out.append("| %s%s | %.1f | %.4f | %.2f / %.2f | %.0f (%.0f) / %.0f | %.4f | **%.4f** | %.4f | %s | %s | %s / %s |" % (
            d, " **(median: published)**" if d == med[0] else "", l["value"], l["roofline"]["kernel_avg_ms"], v["box_peak_2wave"], v["box_peak_8wave"],
            v["shader_mhz_during_timed_launches"], v["slowest_xcd_mhz_during_timed_launches"], v["shader_mhz_during_probe"],
            v["frac_of_box_peak_at_kernel_occupancy"], v["frac_of_box_peak_per_clock"], v["frac"],
            ("%.1f (%.4f; %.4f)" % (p1080["mpix_s"], p1080["valu"]["frac_of_box_peak_at_kernel_occupancy"], p1080["valu"].get("frac_of_box_peak_per_clock", 0))) if p1080 else "-",
            ("%.1f" % l["separable_mode"]["mpix_s"]) if l.get("separable_mode") else "-",
            pr.get("headline", {}).get("regret", "-"), pr.get("1080p x128", {}).get("regret", "-")))
    vals = [l["value"] for _, _, l in lines]
    fr = [l["valu"]["frac_of_box_peak_at_kernel_occupancy"] for _, _, l in lines]
    fc = [l["valu"]["frac_of_box_peak_per_clock"] for _, _, l in lines]
    mh = [l["valu"]["shader_mhz_during_timed_launches"] for _, _, l in lines]
    out += ["", "`value` spread: %.1f ... %.1f k (%.1f %%).  Fraction of the box's own two-wave peak: %.4f ... %.4f (%.1f %% spread) -- NOT box-invariant, because the boxes differ in the clock they hold"
            % (min(vals) / 1e3, max(vals) / 1e3, 100.0 * (max(vals) - min(vals)) / min(vals), min(fr), max(fr), 100.0 * (max(fr) - min(fr)) / min(fr)),
            "under the SSIM kernel's load (%.0f ... %.0f MHz, %.1f %%) far more than under the probe's.  PER CLOCK -- the kernel's lane-operations per shader cycle over the probe's -- the fraction is"
            % (min(mh), max(mh), 100.0 * (max(mh) - min(mh)) / min(mh)),
            "%.4f ... %.4f (%.2f %% spread): that is the figure two boxes of different speed agree on, and the one a kernel change moves." % (min(fc), max(fc), 100.0 * (max(fc) - min(fc)) / min(fc))]
    open(os.path.join(ROOT, "profiles", "%s_box_spread.md" % tag), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
