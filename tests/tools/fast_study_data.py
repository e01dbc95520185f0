#!/usr/bin/env python3
"""Decode the reference's five test image sets (tests/rmgr-ssim-tests.cpp:338-465) and run the REAL reference kernels
(oracle/_ref) plus the naive double oracle over them; cache inputs and maps under /tmp/faststudy for
tests/tools/fast_mode_model.py.  Build-container only (needs /root/reference + PIL); nothing here is committed data.
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF_IMAGES = "/root/reference/tests/images/"
CACHE = "/tmp/faststudy"
QUALITIES = ("00", "10", "20", "30", "40", "50", "60", "70", "80", "90", "100")


def pairs():
    """yield (set, name, a, b): every (image pair, channel) the reference's tests run"""
    ein = np.array(Image.open(REF_IMAGES + "einstein.png"))
    for n in ("einstein", "meanshift", "contrast", "impulse", "blur", "jpg"):
        yield "einstein", "einstein_" + n, ein, np.array(Image.open(REF_IMAGES + n + ".png"))
    for res in ("360", "1080"):
        png = np.array(Image.open(REF_IMAGES + "big_buck_bunny_%s_07806.png" % res).convert("RGB"))
        for q in QUALITIES:
            jpg = np.array(Image.open(REF_IMAGES + "big_buck_bunny_%s_07806_%s.jpg" % (res, q)).convert("RGB"))
            for c in range(3):
                yield "bbb" + res, "bbb%s_q%s_ch%d" % (res, q, c), np.ascontiguousarray(png[:, :, c]), np.ascontiguousarray(jpg[:, :, c])
            if res == "360":
                for (cw, ch) in ((255, 63), (257, 65)):
                    for c in range(3):
                        yield "bbb%d" % cw, "bbb%dx%d_q%s_ch%d" % (cw, ch, q, c), np.ascontiguousarray(png[:ch, :cw, c]), np.ascontiguousarray(jpg[:ch, :cw, c])


def main():
    os.makedirs(CACHE, exist_ok=True)
    index = []
    for s, name, a, b in pairs():
        p = os.path.join(CACHE, name + ".npz")
        if not os.path.exists(p):
            fma, fsum, fmap = oracle.ref_ssim(a, b, want_map=True, impl=5, threads=8)
            nv, nmap = oracle.ref_naive_f64(a, b, want_map=True)
            np.savez(p, a=a, b=b, fma=np.float32(fma), fma_map=fmap, naive=np.float64(nv), naive_map=nmap.astype(np.float64))
            print(name, a.shape, float(fma), nv, "ref-vs-naive px %.3e glob %.3e" % (np.abs(fmap.astype(np.float64) - nmap).max(), abs(float(fma) - nv)))
        index.append((s, name))
    with open(os.path.join(CACHE, "index.txt"), "w") as f:
        for s, n in index:
            f.write("%s %s\n" % (s, n))
    print(len(index), "pairs cached in", CACHE)


if __name__ == "__main__":
    main()
