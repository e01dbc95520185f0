#!/usr/bin/env python3
"""Kernel time for interleaved RGB input (step 3), all three channels of every pair in one launch.

usage: python3 tools/ab_rgb.py [pairs=4] [size=4096] [mode=0] [rounds=5]     (RMGR_SSIM_LIB selects the build)
"""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
    pairs, size, mode, rounds = arg(1, 4), arg(2, 4096), arg(3, 0), arg(4, 5)
    ctx = ssim_amd.Context(0, mode=mode)
    n = 3 * pairs
    params = (ssim_amd.Params * n)()
    keep = []
    for i in range(pairs):
        planes = [synth.pair_numpy(size, size, synth.BASE_SEED + 3 * i + c) for c in range(3)]
        a = np.ascontiguousarray(np.stack([p[0] for p in planes], axis=-1))
        b = np.ascontiguousarray(np.stack([p[1] for p in planes], axis=-1))
        da, db = ctx.upload(a), ctx.upload(b)
        keep += [da, db]
        for c in range(3):
            params[3 * i + c] = ssim_amd.make_params(size, size, da.ptr + c, 3, 3 * size, db.ptr + c, 3, 3 * size)
    sums = ctx.alloc(8 * n)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        ctx.enqueue_batch(params, n, sums.ptr)
        ctx.synchronize()
    res = []
    for _ in range(rounds):
        ctx.set_profiling(True)
        for _ in range(5):
            ctx.enqueue_batch(params, n, sums.ptr)
        ctx.synchronize()
        k, ms = ctx.get_profile()
        ctx.set_profiling(False)
        res.append(ms / k)
    v = ssim_amd.finalize(sums.download(np.float64, (n,)), size, size)
    med = statistics.median(res)
    print("lib %s | %d RGB pairs %dx%d mode %d: median %.4f ms (%.1f Gpix/s per channel result)  ssim[0..2] = %.9f %.9f %.9f"
          % (os.path.basename(ssim_amd.LIB_PATH), pairs, size, size, mode, med, float(size) * size * n / med / 1e6, v[0], v[1], v[2]))
    ctx.close()


if __name__ == "__main__":
    main()
