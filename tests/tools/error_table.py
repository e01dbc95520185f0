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
         ("fast (reference-order E planes, separable mu)", ssim_amd.MODE_FAST), ("separable (all planes, centred)", ssim_amd.MODE_SEPARABLE),
         ("double (RMGR_SSIM_USE_DOUBLE)", ssim_amd.MODE_DOUBLE)]
SETS = ["einstein", "bbb255", "bbb257", "bbb360", "bbb1080"]


def load_sets():
    """{set: [(name, a, b, fma value, fma map, naive value, naive map)]}: the reference's five test image sets
    (tests/rmgr-ssim-tests.cpp:338-465), reference values from the oracle (pinned to the real kernels by refsets.json)"""
    import hashlib
    from PIL import Image
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    ref = json.load(open(os.path.join(GOLDEN, "refsets.json")))["sets"]
    threads = oracle.oracle_lib().oracle_max_threads()
    out = {}
    for s in SETS:
        rows = []
        if s == "einstein":
            items = [(n, man[n]) for n in sorted(man) if n.startswith("einstein_")]
        else:
            items = [(s + "_" + k, ref[s]["pairs"][k]) for k in sorted(ref[s]["pairs"])]
        cache = {}
        for n, e in items:
            if s == "einstein":
                a = np.fromfile(os.path.join(GOLDEN, e["a"]), np.uint8).reshape(e["height"], e["width"])
                b = np.fromfile(os.path.join(GOLDEN, e["b"]), np.uint8).reshape(e["height"], e["width"])
            else:
                for f in (e["a_file"], e["b_file"]):
                    if f not in cache:
                        cache[f] = np.array(Image.open(os.path.join(GOLDEN, "images", f)).convert("RGB"))
                a = np.ascontiguousarray(cache[e["a_file"]][:e["height"], :e["width"], e["channel"]])
                b = np.ascontiguousarray(cache[e["b_file"]][:e["height"], :e["width"], e["channel"]])
                assert hashlib.sha256(a.tobytes()).hexdigest() == e["a_sha256"] and hashlib.sha256(b.tobytes()).hexdigest() == e["b_sha256"], n
            fv, _, fm = oracle.ssim_f32(a, b, want_map=True, threads=threads)
            assert "0x%08x" % np.float32(fv).view(np.uint32) == e["fma"]["ssim_hex"], n
            nv, _, nm = oracle.ssim_naive_f64(a, b, want_map=True, threads=threads)
            rows.append((n, a, b, float(fv), fm, nv, nm))
        out[s] = rows
    return out


def main():
    sets = load_sets()
    ctx = ssim_amd.Context(0)
    print("# Accuracy per arithmetic mode and image set (the reference's five test sets, %d pairs), and speed of the drop-in call\n" % sum(len(v) for v in sets.values()))
    print("Device: %s\n" % ctx.describe())
    print("`vs naive` = against the naive double-precision oracle (the reference's own test oracle, tests/ssim_naive.h; its test")
    print("tolerances are 2e-6 global / 1e-3 per pixel, tests/rmgr-ssim-tests.cpp:98-104); `vs FMA` = against the reference's FMA path")
    print("(north_star's tolerance: 1.5e-6 global / 6.3e-4 per pixel).  For scale, the reference README.md:89-92 reports for its")
    print("single-precision paths vs a quad-precision reference: global avg 1.75e-7 / max 1.49e-6, per-pixel avg 5.61e-6 / max 6.22e-4.\n")
    print("| mode | set (pairs) | global vs naive avg | max | pixel vs naive avg | max | global vs FMA max | pixel vs FMA max | maps identical to the FMA path's |")
    print("|---|---|---|---|---|---|---|---|---|")
    for label, mode in MODES:
        ctx.set_mode(mode)
        for s in SETS:
            ge, gf, pe_sum, pe_max, pf_max, npx, same = [], [], 0.0, 0.0, 0.0, 0, 0
            for (n, a, b, fv, fm, nv, nm) in sets[s]:
                v, m = ctx.ssim_planes(a, b, want_map=True)
                ge.append(abs(float(v) - nv)); gf.append(abs(float(v) - fv))
                d = np.abs(m.astype(np.float64) - nm)
                pe_sum += d.sum(); pe_max = max(pe_max, d.max()); npx += d.size
                pf_max = max(pf_max, float(np.abs(m.astype(np.float64) - fm.astype(np.float64)).max()))
                same += bool(np.array_equal(m.view(np.uint32), fm.view(np.uint32)))
            print("| %s | %s (%d) | %.3g | %.3g | %.3g | %.3g | %.3g | %.3g | %d / %d |" % (label, s, len(sets[s]), np.mean(ge), np.max(ge), pe_sum / npx, pe_max,
                  np.max(gf), pf_max, same, len(sets[s])))
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
