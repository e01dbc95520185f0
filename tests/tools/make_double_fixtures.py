#!/usr/bin/env python3
"""Expected outputs of the reference's OWN fp64 build (RMGR_SSIM_USE_DOUBLE, CMakeLists.txt:53 / src/ssim_internal.h:33-37)
for BASELINE.json configs[4]: tests/golden/ref_double.json (+ two full maps and a strided sample of the 4096^2 map).

Runs ONLY in the build container: needs oracle/_ref/libssim_ref_double.so = src/ssim_fma.cpp + src/ssim_avx.cpp of
/root/reference compiled with -DRMGR_SSIM_USE_DOUBLE=1 (oracle/Makefile).  Inputs are the pairs tests/golden/manifest.json
already holds (the reference's einstein set, the BBB crops) and the synthetic 4096^2 pair of SURVEY.md 8(d).  What is stored
is data: per pair the float the double build returns (hex bits), its fp64 sum (serial tile order), sha256 of its float map;
full maps for two pairs; for the 4096^2 pair every 997th map element.  No reference source text is stored.

What the numbers document: the double build's kernels keep FLOAT-typed tap literals, so it is NOT the exact
value: up to ~9e-7 per pixel and ~5e-7 globally away from tests/ssim_naive.h<double> (README.md:92 of the reference:
4.75e-7 / 9.21e-6 maxima) -- which is why MODE_DOUBLE, whose contract is naive<double> to 1e-7, is compared with it here
under a stated bound instead of bit for bit.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SAMPLE_STEP = 997


def f32_hex(v):
    return "0x%08x" % np.float32(v).view(np.uint32)


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def entry(a, b, threads=1):
    v, s, m = oracle.ref_ssim(a, b, want_map=True, impl=5, threads=threads, double=True)
    v0, s0, _ = oracle.ref_ssim(a, b, want_map=False, impl=5, threads=1, double=True)      # SIMD sum_tile (no map): same value
    assert f32_hex(v) == f32_hex(v0), (v, v0)
    nv, _, nm = oracle.ssim_naive_f64(a, b, want_map=True, threads=8)
    h, w = a.shape
    return {"ssim_hex": f32_hex(v), "ssim": repr(float(v)), "sum_serial_nomap": repr(s0), "mean": repr(s0 / (w * h)),
            "map_sha256": sha(m), "naive_f64": repr(nv),
            "abs_err_vs_naive": {"global": float(abs(s0 / (w * h) - nv)), "pixel": float(np.abs(m.astype(np.float64) - nm).max())}}, m


def main():
    assert oracle.have_ref_double(), "make -C oracle first"
    with open(os.path.join(OUT, "manifest.json")) as f:
        manifest = json.load(f)
    out = {"_about": "reference fp64 build (RMGR_SSIM_USE_DOUBLE=1) outputs; tests/tools/make_double_fixtures.py", "pairs": {}}
    for name in sorted(k for k in manifest if not k.startswith("_")):
        ent = manifest[name]
        w, h = ent["width"], ent["height"]
        a = np.fromfile(os.path.join(OUT, ent["a"]), np.uint8).reshape(h, w)
        b = np.fromfile(os.path.join(OUT, ent["b"]), np.uint8).reshape(h, w)
        e, m = entry(a, b)
        if name in ("einstein_blur", "einstein_jpg"):
            np.save(os.path.join(OUT, name + ".ref_double_map.npy"), m)
            e["map"] = name + ".ref_double_map.npy"
        out["pairs"][name] = e
        print("%-22s %s  vs naive: global %.2e pixel %.2e" % (name, e["ssim"], e["abs_err_vs_naive"]["global"], e["abs_err_vs_naive"]["pixel"]))
    a, b = oracle.synth_pair(4096, 4096, 0x5EED)
    e, m = entry(a, b, threads=8)
    sample = np.ascontiguousarray(m.reshape(-1)[::SAMPLE_STEP])
    np.save(os.path.join(OUT, "synth4096_5eed.ref_double_map_sample.npy"), sample)
    e.update({"width": 4096, "height": 4096, "seed": 0x5EED, "map_sample": "synth4096_5eed.ref_double_map_sample.npy", "map_sample_step": SAMPLE_STEP})
    out["synth4096_5eed"] = e
    print("synth 4096^2: %s  vs naive: global %.2e pixel %.2e" % (e["ssim"], e["abs_err_vs_naive"]["global"], e["abs_err_vs_naive"]["pixel"]))
    out["worst_vs_naive"] = {"global": max([p["abs_err_vs_naive"]["global"] for p in out["pairs"].values()] + [e["abs_err_vs_naive"]["global"]]),
                             "pixel": max([p["abs_err_vs_naive"]["pixel"] for p in out["pairs"].values()] + [e["abs_err_vs_naive"]["pixel"]])}
    with open(os.path.join(OUT, "ref_double.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("worst vs naive<double>:", out["worst_vs_naive"])


if __name__ == "__main__":
    main()
