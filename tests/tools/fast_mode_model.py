#!/usr/bin/env python3
"""CPU model of MODE_FAST (the separable fp32 blur of ssim_kernels.hip blur_separable + ssim_px) in numpy, to study
its rounding bias against the naive double oracle on the committed fixtures.  fma(a,b,c) is modelled as
float32(float64(a)*float64(b) + float64(c)) (the product of two floats is exact in double).

usage: python tests/tools/fast_mode_model.py        (CPU only; reads tests/golden)
"""
import itertools
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLDEN = os.path.join(ROOT, "tests", "golden")
f32, f64 = np.float32, np.float64


def fma(a, b, c):
    return (a.astype(f64) * f64(b) + c.astype(f64)).astype(f32)


def true_taps():
    g = np.exp(-(np.arange(6, dtype=f64) ** 2) / (2 * 1.5 * 1.5))
    return g / (g[0] + 2 * g[1:].sum())


def nearest_taps():
    return true_taps().astype(f32)


def balanced_taps():
    """float taps, each the rounding of the true tap up or down, whose 11-tap sum is closest to 1."""
    g = true_taps()
    lo = np.array([np.nextafter(f32(v), f32(0)) if f32(v) > v else f32(v) for v in g], f32)
    hi = np.array([np.nextafter(v, f32(1)) for v in lo], f32)
    best = None
    for pick in itertools.product((0, 1), repeat=6):
        t = np.where(np.array(pick) == 1, hi, lo).astype(f64)
        err = abs(t[0] + 2 * t[1:].sum() - 1.0)
        dev = np.abs(t - g).max()
        if best is None or (err, dev) < best[:2]:
            best = (err, dev, t.astype(f32))
    return best[2]


def blur(P, g, order="center_first"):
    """P: float32 H x W plane (edge-replicated outside)."""
    H, W = P.shape
    pad = np.pad(P, 5, mode="edge")
    # horizontal on every padded row
    c = pad[:, 5:5 + W]
    s = [c] + [pad[:, 5 + i:5 + i + W] + pad[:, 5 - i:5 - i + W] for i in range(1, 6)]
    if order == "center_first":
        h = (s[0] * g[0]).astype(f32)
        for i in range(1, 6):
            h = fma(s[i], g[i], h)
    else:   # smallest taps first
        h = (s[5] * g[5]).astype(f32)
        for i in (4, 3, 2, 1, 0):
            h = fma(s[i], g[i], h)
    # vertical: ring order, top row first
    acc = (h[0:H] * g[5]).astype(f32)
    for k in range(1, 11):
        acc = fma(h[k:k + H], g[abs(k - 5)], acc)
    return acc


def ssim_fast(a, b, g, order="center_first"):
    a = a.astype(f32); b = b.astype(f32)
    muA, muB = blur(a, g, order), blur(b, g, order)
    eAA, eBB, eAB = blur(a * a, g, order), blur(b * b, g, order), blur(a * b, g, order)
    c1 = f32((0.01 * 255.0) ** 2); c2 = f32((0.03 * 255.0) ** 2)
    muA2, muB2, muAB = muA * muA, muB * muB, muA * muB
    sA2, sB2, sAB = eAA - muA2, eBB - muB2, eAB - muAB
    num = (f32(2) * muAB + c1) * (f32(2) * sAB + c2)
    den = ((muA2 + muB2) + c1) * ((sA2 + sB2) + c2)
    m = (num / den).astype(f32)
    return f32(m.astype(f64).sum() / f64(m.size)), m


def main():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    names = sorted(k for k in man if not k.startswith("_"))
    variants = {"nearest taps": (nearest_taps(), "center_first"), "balanced taps": (balanced_taps(), "center_first"),
                "balanced, small first": (balanced_taps(), "small_first")}
    for label, (g, order) in variants.items():
        t = g.astype(f64)
        print("%-24s taps %s  sum-1 = %.3e" % (label, " ".join("%.9g" % v for v in g), t[0] + 2 * t[1:].sum() - 1))
    print("%-22s %12s | %s" % ("fixture", "fma-ref err", " | ".join("%-22s" % k for k in variants)))
    worst = {k: 0.0 for k in variants}
    for n in names:
        e = man[n]
        w, h = e["width"], e["height"]
        a = np.fromfile(os.path.join(GOLDEN, e["a"]), np.uint8).reshape(h, w)
        b = np.fromfile(os.path.join(GOLDEN, e["b"]), np.uint8).reshape(h, w)
        naive = float(e["naive_f64"]["ssim"])
        row = []
        for k, (g, order) in variants.items():
            v, _ = ssim_fast(a, b, g, order)
            d = abs(float(v) - naive)
            worst[k] = max(worst[k], d)
            row.append("%-22.3e" % d)
        print("%-22s %12.3e | %s" % (n, abs(float(e["fma"]["ssim"]) - naive), " | ".join(row)))
    print("worst:", {k: "%.3e" % v for k, v in worst.items()})


if __name__ == "__main__":
    main()
