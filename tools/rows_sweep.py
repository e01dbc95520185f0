#!/usr/bin/env python3
"""Strip-height sweep for one resident batch: kernel time (HIP events) per requested strip height.
usage: python tools/rows_sweep.py pairs width height mode rows,rows,...   (0 = the library's own plan)"""
import statistics
import sys

import numpy as np

sys.path.insert(0, ".")
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402

pairs, w, h, mode = (int(v) for v in sys.argv[1:5])
rows_list = [int(v) for v in sys.argv[5].split(",")]
ctx = ssim_amd.Context(0, mode=mode)
params = (ssim_amd.Params * pairs)()
for i in range(pairs):
    da, db = ctx.alloc(w * h), ctx.alloc(w * h)
    ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
    params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
sums = ctx.alloc(8 * pairs)
for _ in range(30):
    ctx.enqueue_batch(params, pairs, sums.ptr)
ctx.synchronize()
res = {r: [] for r in rows_list}
for rnd in range(4):
    for r in rows_list:
        ctx.set_tuning(r, 2 if r else 0)
        ctx.enqueue_batch(params, pairs, sums.ptr)
        ctx.synchronize()
        ctx.get_profile(); ctx.set_profiling(True)
        for _ in range(4):
            ctx.enqueue_batch(params, pairs, sums.ptr)
        ctx.synchronize()
        n, ms = ctx.get_profile(); ctx.set_profiling(False)
        res[r].append(ms / n)
px = float(pairs) * w * h
out = []
for r in rows_list:
    ctx.set_tuning(r, 2 if r else 0)
    p = ssim_amd.get_plan(w, h, pairs, ctx)
    med = statistics.median(res[r])
    out.append("%4d->%3dx%-2d %6.1f" % (r, p.stripRows, p.stripsY, px / med / 1e6))
print("%4d x %dx%d mode %d: " % (pairs, w, h, mode) + " | ".join(out))
