#!/usr/bin/env python3
"""Throughput of host-resident pairs: a loop over the single-pair drop-in call vs the pipelined batch entry point."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd
from ssim_amd import synth
for (w, h, n) in [(256, 256, 400), (640, 480, 200), (1920, 1080, 64), (4096, 4096, 12)]:
    ps = [synth.pair_numpy(w, h, synth.BASE_SEED + i) for i in range(min(n, 16))]
    ps = [ps[i % len(ps)] for i in range(n)]
    for a, b in ps[:3]: ssim_amd.compute_ssim(a, b)
    t = time.perf_counter()
    singles = [ssim_amd.compute_ssim(a, b)[0] for a, b in ps]
    t_loop = time.perf_counter() - t
    ssim_amd.compute_ssim_batch(ps[:4])
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        got = ssim_amd.compute_ssim_batch(ps)
        best = min(best, time.perf_counter() - t)
    assert all(np.float32(x).view(np.uint32) == np.float32(y).view(np.uint32) for x, y in zip(singles, got))
    px = float(w) * h * n
    print("%dx%d x %d host pairs: loop of single calls %.1f us/pair (%.2f Gpix/s) | pipelined batch %.1f us/pair (%.2f Gpix/s) | %.2fx"
          % (w, h, n, t_loop / n * 1e6, px / t_loop / 1e9, best / n * 1e6, px / best / 1e9, t_loop / best))
