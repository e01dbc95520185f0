#!/usr/bin/env python3
"""Does the cost of the map (and of the pixel loads) depend on the ROW STRIDES being powers of two?  All strips of a launch walk their rows
at about the same pace; with a power-of-two stride the rows that thousands of wavefronts touch at one moment are a multiple of the stride
apart -- if the memory channels are selected by plain address bits they all land on the same few channels (partition camping).
usage: python tools/stride_probe.py [pairs=2] [size=8192] [mode=4] [rows=0] [map|img]     (runs on the GPU box)"""
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402

arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
pairs, size, mode, rows = arg(1, 2), arg(2, 8192), arg(3, 4), arg(4, 0)
ctx = ssim_amd.Context(0, mode=mode)
ctx.set_tuning(rows, 0)


def run(img_pad, map_pad, want_map):
    istride, mstride = size + img_pad, size + map_pad
    keep, params = [], (ssim_amd.Params * pairs)()
    for i in range(pairs):
        da, db = ctx.alloc(istride * size), ctx.alloc(istride * size)
        ctx.synth_pair(da.ptr, istride, db.ptr, istride, size, size, synth.BASE_SEED + i)
        dm = ctx.alloc(4 * mstride * size) if want_map else None
        keep += [da, db] + ([dm] if dm else [])
        params[i] = ssim_amd.make_params(size, size, da.ptr, 1, istride, db.ptr, 1, istride, dm.ptr if dm else None, 1, mstride)
    sums = ctx.alloc(8 * pairs)
    keep.append(sums)
    for _ in range(3):
        ctx.enqueue_batch(params, pairs, sums.ptr)
    ctx.synchronize()
    t = []
    for _ in range(5):
        ctx.set_profiling(True)
        for _ in range(5):
            ctx.enqueue_batch(params, pairs, sums.ptr)
        ctx.synchronize()
        n, ms = ctx.get_profile()
        ctx.set_profiling(False)
        t.append(ms / n)
    v = ssim_amd.finalize(sums.download(np.float64, (pairs,)), size, size)[0]
    for d in keep:
        d.free()
    return statistics.median(t), min(t), v


px = float(size) * size * pairs
print("pairs %d size %d mode %d strip rows %d (plan: %d)" % (pairs, size, mode, rows, ssim_amd.get_plan(size, size, pairs, ctx).stripRows))
SETS = {"map": [(0, 0, 0), (0, 0, 1), (0, 16, 1), (0, 64, 1), (0, 1024, 1), (0, 1040, 1), (256, 0, 1), (256, 64, 1), (4160, 1040, 1)],
        "img": [(0, 0, 0), (64, 0, 0), (256, 0, 0), (0, 0, 0), (1024, 0, 0), (4160, 0, 0), (0, 0, 0), (64, 0, 0), (256, 0, 0)]}
which = sys.argv[5] if len(sys.argv) > 5 else "map"
run(0, 0, 0); run(0, 0, 1)          # warm the clocks
for (ip, mp, wm) in SETS[which]:
    med, best, v = run(ip, mp, wm)
    print("image stride %5d  map %-12s: median %.4f ms (%.1f Gpix/s)  best %.4f ms (%.1f Gpix/s)  ssim %.9f"
          % (size + ip, ("stride %d" % (size + mp)) if wm else "none", med, px / med / 1e6, best, px / best / 1e6, v))
ctx.close()
