#!/usr/bin/env python3
"""Interleaved A/B of kernel variants on one resident batch (within-process, several rounds).

usage: python tools/ab.py [pairs=8] [size=4096] [mode=0] [rows=256] [variants=0,1] [rounds=5] [map=0] [height=size]
Set RMGR_SSIM_LIB=<path> to test an alternative build of the library in a separate run.
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
    arg = lambda i, d: sys.argv[i] if len(sys.argv) > i else d
    pairs, size, mode, rows = int(arg(1, 8)), int(arg(2, 4096)), int(arg(3, 0)), int(arg(4, 256))
    variants = [int(v) for v in arg(5, "0,1").split(",")]
    rounds, want_map = int(arg(6, 5)), int(arg(7, 0))
    height = int(arg(8, size))
    ctx = ssim_amd.Context(0, mode=mode)
    params = (ssim_amd.Params * pairs)()
    keep = []
    native = hasattr(ctx.lib, "rmgr_ssim_hip_synth_pair_device")     # older builds: generate on the host
    for i in range(pairs):
        if native:
            da, db = ctx.alloc(size * height), ctx.alloc(size * height)
            ctx.synth_pair(da.ptr, size, db.ptr, size, size, height, synth.BASE_SEED + i)
        else:
            a, b = synth.pair_numpy(size, height, synth.BASE_SEED + i)
            da, db = ctx.upload(a), ctx.upload(b)
        dm = ctx.alloc(4 * size * height) if want_map else None
        keep += [da, db, dm]
        params[i] = ssim_amd.make_params(size, height, da.ptr, 1, size, db.ptr, 1, size, dm.ptr if dm else None, 1, size)
    sums = ctx.alloc(8 * pairs)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        ctx.enqueue_batch(params, pairs, sums.ptr)
        ctx.synchronize()
    res = {v: [] for v in variants}
    vals = {}
    for _ in range(rounds):
        for v in variants:
            ctx.set_tuning(rows, v)
            ctx.enqueue_batch(params, pairs, sums.ptr)
            ctx.synchronize()
            ctx.set_profiling(True)
            for _ in range(5):
                ctx.enqueue_batch(params, pairs, sums.ptr)
            ctx.synchronize()
            n, ms = ctx.get_profile()
            ctx.set_profiling(False)
            res[v].append(ms / n)
            vals[v] = ssim_amd.finalize(sums.download(np.float64, (pairs,)), size, height)[0]
    px = float(size) * height * pairs
    print("lib %s | pairs %d size %dx%d mode %d rows %d map %d" % (os.path.basename(ssim_amd.LIB_PATH), pairs, size, height, mode, rows, want_map))
    for v in variants:
        med, best = statistics.median(res[v]), min(res[v])
        print("  variant %d: median %.4f ms (%.1f Gpix/s)  best %.4f ms (%.1f Gpix/s)  ssim[0]=%.9f"
              % (v, med, px / med / 1e6, best, px / best / 1e6, vals[v]))
    ctx.close()


if __name__ == "__main__":
    main()
