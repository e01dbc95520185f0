#!/usr/bin/env python3
"""Minimal torch-free workload for rocprofv3 counter passes: the bench.py batch (N distinct 4096^2
pairs resident in HBM, global SSIM only, or with the map) enqueued K times through the C ABI.

usage: python3 tools/profile_target.py [pairs=8] [steps=5] [mode=0] [map=0] [width=4096] [height=width] [rgb=0] [variant=0[,variant...]]
rgb=1: every pair is an interleaved RGB image pair (step 3) and all three channels go into the launch (3 x pairs results)
variant: a comma-separated list runs `steps` launches of each tuning variant in turn (the counter rows of one rocprofv3 pass, in dispatch order)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
    pairs, steps, mode, want_map, w = arg(1, 8), arg(2, 5), arg(3, 0), arg(4, 0), arg(5, 4096)
    h = arg(6, w)
    rgb = arg(7, 0)
    ctx = ssim_amd.Context(0, mode=mode)
    variants = [int(v) for v in (sys.argv[8] if len(sys.argv) > 8 else "0").split(",")]
    n = pairs * (3 if rgb else 1)
    params = (ssim_amd.Params * n)()
    keep = []
    for i in range(pairs):
        if rgb:
            planes = [synth.pair_numpy(w, h, synth.BASE_SEED + 3 * i + c) for c in range(3)]
            a = np.ascontiguousarray(np.stack([p[0] for p in planes], axis=-1))
            b = np.ascontiguousarray(np.stack([p[1] for p in planes], axis=-1))
            da, db = ctx.upload(a), ctx.upload(b)
            keep += [da, db]
            for c in range(3):
                params[3 * i + c] = ssim_amd.make_params(w, h, da.ptr + c, 3, 3 * w, db.ptr + c, 3, 3 * w)
            continue
        da, db = ctx.alloc(w * h), ctx.alloc(w * h)
        ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
        dm = ctx.alloc(4 * w * h) if want_map else None
        keep += [da, db, dm]
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w, dm.ptr if dm else None, 1, w)
    pairs = n
    sums = ctx.alloc(8 * pairs)
    for v in variants:
        ctx.set_tuning(0, v)
        for _ in range(steps):
            ctx.enqueue_batch(params, pairs, sums.ptr)
        ctx.synchronize()
    res = ssim_amd.finalize(sums.download(np.float64, (pairs,)), w, h)
    print("pairs %d steps %d mode %d map %d size %dx%d: ssim[0] = %.9f (0x%08x)" % (pairs, steps, mode, want_map, w, h, res[0], res[0].view(np.uint32)))
    ctx.close()


if __name__ == "__main__":
    main()
