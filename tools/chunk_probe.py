#!/usr/bin/env python3
"""One launch of N pairs against N/k back-to-back launches of k pairs (wall time per step, HIP-event kernel time): is a
launch that fills the wave slots exactly once more efficient than one that needs several rounds?
usage: python tools/chunk_probe.py [pairs=32] [size=4096] [mode=0] [chunks=1,2,4,8] [height=size]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    arg = lambda i, d: sys.argv[i] if len(sys.argv) > i else d
    pairs, size, mode = int(arg(1, 32)), int(arg(2, 4096)), int(arg(3, 0))
    chunks = [int(v) for v in arg(4, "1,2,4,8").split(",")]
    height = int(arg(5, size))
    ctx = ssim_amd.Context(0, mode=mode)
    params = (ssim_amd.Params * pairs)()
    keep = []
    for i in range(pairs):
        da, db = ctx.alloc(size * height), ctx.alloc(size * height)
        ctx.synth_pair(da.ptr, size, db.ptr, size, size, height, synth.BASE_SEED + i)
        keep += [da, db]
        params[i] = ssim_amd.make_params(size, height, da.ptr, 1, size, db.ptr, 1, size)
    sums = ctx.alloc(8 * pairs)
    psize = ctypes.sizeof(ssim_amd.Params)

    def step(k):
        n = pairs // k
        for c in range(k):
            sub = (ssim_amd.Params * n).from_buffer(params, c * n * psize)
            ctx.enqueue_batch(sub, n, sums.ptr + 8 * c * n)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        step(1)
        ctx.synchronize()
    ref = None
    for rnd in range(3):
        for k in chunks:
            if pairs % k:
                continue
            step(k); ctx.synchronize()
            ctx.get_profile(); ctx.set_profiling(True)
            t0 = time.perf_counter()
            for _ in range(10):
                step(k)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 10
            n, ms = ctx.get_profile(); ctx.set_profiling(False)
            got = sums.download(np.float64, (pairs,))
            if ref is None:
                ref = got.copy()
            assert np.array_equal(got.view(np.uint64), ref.view(np.uint64)), k
            px = float(size) * height * pairs
            print("%d x %dx%d mode %d as %d launch(es) of %d: wall %.4f ms/step (%.1f Gpix/s), kernels %.4f ms/step (%.1f Gpix/s)"
                  % (pairs, size, height, mode, k, pairs // k, wall * 1e3, px / wall / 1e9, ms / 10, px / (ms / 10) / 1e6))
    ctx.close()


if __name__ == "__main__":
    main()
