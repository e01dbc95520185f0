#!/usr/bin/env python3
"""Generate tests/golden/ from the reference's own test images and the REAL reference kernels.

Runs ONLY in the build container (needs /root/reference, PIL and oracle/_ref/libssim_ref.so).
What it commits is data: decoded raw 8-bit pixels (inputs) and the numbers the reference
produces on them (expected outputs).  No reference source text is stored.

Inputs
  einstein set   tests/images/{einstein,meanshift,contrast,impulse,blur,jpg}.png  256x256 gray
                 (tests/rmgr-ssim-tests.cpp:338-370)
  bbb crops      big_buck_bunny_360_07806.png vs _00.jpg / _50.jpg, 3 channels, cropped to
                 255x63 and 257x65 as tests/rmgr-ssim-tests.cpp:428-465 do (one below / one above
                 the 256x64 tile).  JPEGs are decoded by PIL here, NOT stb_image, so the
                 hard-coded BBB constants of the reference tests do not apply;
                 the pin for these is the reference library's output on these exact pixels.
Outputs per pair (golden/manifest.json)
  fma / avx      global float (hex bits + repr), fp64 sum, sha256 of the float map
  naive_f64      tests/ssim_naive.h global (repr, 17 digits), sha256 of the double map
  maps           full FMA float map and naive double map for a few pairs (.npy)
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF_IMAGES = "/root/reference/tests/images/"
OUT = os.path.join(ROOT, "tests", "golden")

# tests/rmgr-ssim-tests.cpp:352-360 -- quad-precision constants the reference pins its oracle to
EINSTEIN_GOLDENS = {
    "einstein": "1.000000000000000000000000000000000",
    "meanshift": "0.987345868581455342542598819456431",
    "contrast": "0.901217091012390185892926336265424",
    "impulse": "0.839533769204009687363862456348761",
    "blur": "0.702192033056262932311859850040160",
    "jpg": "0.669938383706498006524758818118705",
}


def f32_hex(v):
    return "0x%08x" % np.float32(v).view(np.uint32)


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def record(name, a, b, keep_maps, manifest, golden=None):
    h, w = a.shape
    fma, fma_sum, fma_map = oracle.ref_ssim(a, b, want_map=True, impl=5)
    avx, avx_sum, avx_map = oracle.ref_ssim(a, b, want_map=True, impl=4)
    fma_nomap, _, _ = oracle.ref_ssim(a, b, want_map=False, impl=5)
    fma_omp, _, _ = oracle.ref_ssim(a, b, want_map=False, impl=5, threads=4)
    assert f32_hex(fma) == f32_hex(fma_nomap) == f32_hex(fma_omp)
    nv, nmap = oracle.ref_naive_f64(a, b, want_map=True)
    ent = {
        "width": w, "height": h,
        "a": name + ".a.u8", "b": name + ".b.u8",
        "fma": {"ssim_hex": f32_hex(fma), "ssim": repr(float(fma)), "sum": repr(fma_sum), "map_sha256": sha(fma_map)},
        "avx": {"ssim_hex": f32_hex(avx), "ssim": repr(float(avx)), "sum": repr(avx_sum), "map_sha256": sha(avx_map)},
        "naive_f64": {"ssim": repr(nv), "map_sha256": sha(nmap)},
    }
    if golden is not None:
        ent["reference_test_golden"] = golden
        assert abs(float(golden) - nv) < 1e-13, (name, golden, nv)  # REF_TOLERANCE, tests/rmgr-ssim-tests.cpp:72
    a.tofile(os.path.join(OUT, ent["a"]))
    b.tofile(os.path.join(OUT, ent["b"]))
    if keep_maps:
        np.save(os.path.join(OUT, name + ".fma_map.npy"), fma_map)
        np.save(os.path.join(OUT, name + ".naive_map.npy"), nmap)
        ent["fma"]["map"] = name + ".fma_map.npy"
        ent["naive_f64"]["map"] = name + ".naive_map.npy"
    manifest[name] = ent
    print("%-22s %dx%d fma=%s avx=%s naive=%.15f" % (name, w, h, ent["fma"]["ssim"], ent["avx"]["ssim"], nv))


def main():
    assert oracle.have_ref(), "build oracle/_ref first (make -C oracle)"
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    ein = np.array(Image.open(REF_IMAGES + "einstein.png"))
    assert ein.shape == (256, 256) and ein.dtype == np.uint8
    for n, g in EINSTEIN_GOLDENS.items():
        img = np.array(Image.open(REF_IMAGES + n + ".png"))
        assert img.shape == (256, 256)
        record("einstein_" + n, ein, img, keep_maps=n in ("blur", "jpg"), manifest=manifest, golden=g)

    png = np.array(Image.open(REF_IMAGES + "big_buck_bunny_360_07806.png").convert("RGB"))
    for q in ("00", "50"):
        jpg = np.array(Image.open(REF_IMAGES + "big_buck_bunny_360_07806_%s.jpg" % q).convert("RGB"))
        assert jpg.shape == png.shape == (360, 640, 3)
        for (cw, ch) in ((255, 63), (257, 65)):
            for c in range(3):
                a = np.ascontiguousarray(png[:ch, :cw, c])
                b = np.ascontiguousarray(jpg[:ch, :cw, c])
                record("bbb%dx%d_q%s_ch%d" % (cw, ch, q, c), a, b, keep_maps=(c == 1), manifest=manifest)

    # interleaved RGB crop kept as ONE buffer to exercise step=3 addressing (init_interleaved,
    # src/ssim.cpp:156-178) -- expected values are the per-channel entries above.
    np.ascontiguousarray(png[:65, :257, :]).tofile(os.path.join(OUT, "bbb257x65_png.rgb.u8"))
    jpg50 = np.array(Image.open(REF_IMAGES + "big_buck_bunny_360_07806_50.jpg").convert("RGB"))
    np.ascontiguousarray(jpg50[:65, :257, :]).tofile(os.path.join(OUT, "bbb257x65_q50.rgb.u8"))
    manifest["_interleaved"] = {"a": "bbb257x65_png.rgb.u8", "b": "bbb257x65_q50.rgb.u8", "width": 257, "height": 65,
                                "channels": 3, "per_channel": ["bbb257x65_q50_ch%d" % c for c in range(3)]}

    # Synthetic-generator known answers, recomputed here with the real kernels.
    synth = {}
    for (w, h, seed) in ((256, 256, 0x5EED), (1920, 1080, 0x5EED), (1920, 1080, 0x5EEE), (1920, 1080, 0x5EEF),
                         (4096, 4096, 0x5EED), (8192, 8192, 0x5EED)):
        a, b = oracle.synth_pair(w, h, seed)
        fma, fsum, _ = oracle.ref_ssim(a, b, impl=5, threads=8)
        key = "%dx%d_%x" % (w, h, seed)
        synth[key] = {"width": w, "height": h, "seed": seed, "sumA": int(a.sum(dtype=np.int64)), "sumB": int(b.sum(dtype=np.int64)),
                      "first4A": a[0, :4].tolist(), "first4B": b[0, :4].tolist(),
                      "fma": {"ssim_hex": f32_hex(fma), "ssim": repr(float(fma))}}
        if w * h <= 1920 * 1080:
            nv, _ = oracle.ref_naive_f64(a, b)
            synth[key]["naive_f64"] = repr(nv)
        print(key, synth[key])
    manifest["_synthetic"] = synth
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest), "entries to", OUT)


if __name__ == "__main__":
    main()
