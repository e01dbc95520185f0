#!/usr/bin/env python3
"""Reference-style accuracy and speed tables (tests/rmgr-ssim-tests.cpp:165-222 prints the
same two tables for its CPU implementations; README.md:89-92 publishes the accuracy one).

For every arithmetic mode of the engine, over the committed fixture pairs (einstein set + BBB crops,
tests/golden): average / maximum error of the global SSIM and of the per-pixel map against the
naive double-precision oracle (tests/ssim_naive.h semantics), and Mpix/s through the drop-in call
(host pointers) with and without a map.  Runs on the GPU box; uses the oracle as the checker.

usage: python tests/tools/error_table.py > profiles/<round>_error_table.md
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (checker)
import ssim_amd  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODES = [("exact (FMA order)", ssim_amd.MODE_EXACT), ("unfused (AVX/SSE/generic order)", ssim_amd.MODE_UNFUSED),
         ("fast (separable fp32)", ssim_amd.MODE_FAST), ("double (RMGR_SSIM_USE_DOUBLE)", ssim_amd.MODE_DOUBLE)]


def main():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    names = sorted(k for k in man if not k.startswith("_"))
    pairs = []
    for n in names:
        e = man[n]
        a = np.fromfile(os.path.join(GOLDEN, e["a"]), np.uint8).reshape(e["height"], e["width"])
        b = np.fromfile(os.path.join(GOLDEN, e["b"]), np.uint8).reshape(e["height"], e["width"])
        nv, _, nm = oracle.ssim_naive_f64(a, b, want_map=True, threads=8)
        pairs.append((n, a, b, nv, nm))
    ctx = ssim_amd.Context(0)
    print("# Accuracy and speed per arithmetic mode (%d fixture pairs: einstein set + BBB 255x63 / 257x65 crops)\n" % len(pairs))
    print("Device: %s\n" % ctx.describe())
    print("Errors are against the naive double-precision oracle (the reference's own test oracle, tests/ssim_naive.h).")
    print("For scale, the reference README.md:89-92 reports for its single-precision paths: global avg 1.75e-7 / max 1.49e-6,")
    print("per-pixel avg 5.61e-6 / max 6.22e-4; double-precision: 1.31e-7 / 4.75e-7 / 7.35e-8 / 9.21e-6.\n")
    print("| mode | global err avg | global err max | per-pixel err avg | per-pixel err max | identical to reference FMA maps |")
    print("|---|---|---|---|---|---|")
    for label, mode in MODES:
        ctx.set_mode(mode)
        ge, pe_sum, pe_max, npx, same = [], 0.0, 0.0, 0, 0
        for (n, a, b, nv, nm) in pairs:
            v, m = ctx.ssim_planes(a, b, want_map=True)
            ge.append(abs(float(v) - nv))
            d = np.abs(m.astype(np.float64) - nm)
            pe_sum += d.sum(); pe_max = max(pe_max, d.max()); npx += d.size
            if "map_sha256" in man[n]["fma"]:
                import hashlib
                same += hashlib.sha256(np.ascontiguousarray(m).tobytes()).hexdigest() == man[n]["fma"]["map_sha256"]
        print("| %s | %.3g | %.3g | %.3g | %.3g | %d / %d |" % (label, np.mean(ge), np.max(ge), pe_sum / npx, pe_max, same, len(pairs)))
    ctx.close()

    print("\n## Speed through the unchanged drop-in call (host pointers, PCIe staging included), Mpix/s\n")
    print("| input | no map | map | openmp entry, no map |")
    print("|---|---|---|---|")
    for (w, h) in ((256, 256), (1920, 1080), (4096, 4096)):
        a, b = oracle.synth_pair(w, h, 0x5EED)
        row = []
        reuse = np.zeros((h, w), np.float32)       # a caller-owned map buffer, touched once
        for kw in (dict(want_map=False), dict(out_map=reuse), dict(want_map=False, openmp=True)):
            ssim_amd.compute_ssim(a, b, **kw)
            t0 = time.perf_counter()
            reps = 0
            while time.perf_counter() - t0 < 0.5:
                ssim_amd.compute_ssim(a, b, **kw)
                reps += 1
            row.append(w * h * reps / (time.perf_counter() - t0) / 1e6)
        print("| %dx%d | %.0f | %.0f | %.0f |" % (w, h, row[0], row[1], row[2]))


if __name__ == "__main__":
    main()
