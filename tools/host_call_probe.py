#!/usr/bin/env python3
"""Times the unchanged drop-in call (host pointers, pageable memory) with and without a map, per band count of the
pipelined path (RMGR_SSIM_HIP_BANDS; 1 = no overlap between copy-in, kernel and copy-out).
usage: python tools/host_call_probe.py [size=4096] [bands=1,2,4,6,8,12,16]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    bands = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,6,8,12,16").split(",")]
    ctx = ssim_amd.Context(0)
    da, db = ctx.alloc(size * size), ctx.alloc(size * size)
    ctx.synth_pair(da.ptr, size, db.ptr, size, size, size, synth.BASE_SEED)
    ctx.synchronize()
    a, b = da.download(np.uint8, (size, size)), db.download(np.uint8, (size, size))
    m = np.zeros((size, size), np.float32)
    px = float(size) * size

    def timed(fn, n=8):
        fn(); fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts), sorted(ts)[len(ts) // 2]

    os.environ.pop("RMGR_SSIM_HIP_BANDS", None)
    best, med = timed(lambda: ssim_amd.compute_ssim(a, b))
    print("%dx%d no map, default bands: best %.3f ms (%.1f Gpix/s), median %.3f ms" % (size, size, best * 1e3, px / best / 1e9, med * 1e3))
    for nb in bands:
        os.environ["RMGR_SSIM_HIP_BANDS"] = str(nb)
        best, med = timed(lambda: ssim_amd.compute_ssim(a, b))
        print("%dx%d no map, %2d bands: best %.3f ms (%.2f Gpix/s), median %.3f ms (%.2f Gpix/s)" % (size, size, nb, best * 1e3, px / best / 1e9, med * 1e3, px / med / 1e9))
    for nb in bands:
        os.environ["RMGR_SSIM_HIP_BANDS"] = str(nb)
        best, med = timed(lambda: ssim_amd.compute_ssim(a, b, out_map=m))
        print("%dx%d map, %2d bands: best %.3f ms (%.2f Gpix/s), median %.3f ms (%.2f Gpix/s)" % (size, size, nb, best * 1e3, px / best / 1e9, med * 1e3, px / med / 1e9))


if __name__ == "__main__":
    main()
