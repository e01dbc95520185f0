#!/usr/bin/env python3
"""Kernel-time sweep over modes / variants / strip heights on the GPU box (events on the launch
stream, rmgr_ssim_hip_set_profiling).  Prints one line per configuration.

usage: python tools/sweep.py [quick]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ctx = ssim_amd.Context(0)
    print(ctx.describe())
    cases = [("4096x4096 x1", 4096, 4096, 1, False), ("4096x4096 x8", 4096, 4096, 8, False),
             ("8192x8192 x1 +map", 8192, 8192, 1, True), ("8192x8192 x1", 8192, 8192, 1, False),
             ("1920x1080 x64", 1920, 1080, 64, False)]
    if quick:
        cases = cases[:2]
    for (label, w, h, n, want_map) in cases:
        bufs = []
        params = (ssim_amd.Params * n)()
        a, b = synth.pair_numpy(w, h, synth.BASE_SEED)
        for i in range(n):
            if i and n <= 8:
                a, b = synth.pair_numpy(w, h, synth.BASE_SEED + i)
            da, db = ctx.upload(a), ctx.upload(b)
            bufs += [da, db]
            dm = None
            if want_map:
                dm = ctx.alloc(4 * w * h)
                bufs.append(dm)
            params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w, dm.ptr if dm else None, 1, w)
        sums = ctx.alloc(8 * n)
        bufs.append(sums)
        for mode in (0, 1, 2):
            if mode == 2 and (quick or n > 1 or w > 4096):
                continue
            for variant in ((0, 1) if mode != 2 else (0,)):
                for rows in (0, 16, 32, 64, 128, 256):
                    ctx.set_mode(mode)
                    ctx.set_tuning(rows, variant)
                    t_warm = time.perf_counter()          # >= 30 ms of work: clocks settle after the uploads
                    while time.perf_counter() - t_warm < 0.03:
                        ctx.enqueue_batch(params, n, sums.ptr)
                        ctx.synchronize()
                    ctx.set_profiling(True)
                    reps = 10
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        ctx.enqueue_batch(params, n, sums.ptr)
                    ctx.synchronize()
                    wall = (time.perf_counter() - t0) / reps
                    launches, ms = ctx.get_profile()
                    ctx.set_profiling(False)
                    k = ms / launches
                    px = float(w) * h * n
                    res = ssim_amd.finalize(sums.download(np.float64, (n,)), w, h)
                    print("%-20s mode %d variant %d rows %3d: kernel %8.4f ms  %9.1f Mpix/s  %7.1f GB/s(alg)  wall/step %8.4f ms  ssim[0]=%.9f"
                          % (label, mode, variant, rows, k, px / k / 1e3, px * (6 if want_map else 2) / k / 1e6, wall * 1e3, res[0]))
                    sys.stdout.flush()
        for d in bufs:
            d.free()
    ctx.close()


if __name__ == "__main__":
    main()
