#!/usr/bin/env python3
"""HBM roofline position of luminance_kernel (BT.601 luminance of interleaved RGB / RGBA, SURVEY.md 8 f1;
src/ssim-cli.cpp:158-186): a byte kernel, 3-4 B read + 1 B written per pixel, nothing to compute.

usage (GPU box): python tools/luminance_probe.py [size=8192]
Prints GB/s of algorithmic traffic (bytes read + written / time) per layout; time = wall over 50 back-to-back calls.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    ctx = ssim_amd.Context(0)
    print(ctx.describe())
    rng = np.random.default_rng(1)
    for label, step, pad in (("RGB packed (step 3, 4-byte aligned rows)", 3, 0), ("RGBA (step 4)", 4, 0), ("RGB, rows not 4-byte aligned (stride = 3W + 1)", 3, 1)):
        stride = n * step + pad
        host = rng.integers(0, 256, (n, stride), dtype=np.uint8)
        src = ctx.upload(host)
        dst = ctx.alloc(n * n)
        call = lambda: ssim_amd.api._check("rmgr_ssim_hip_luminance_device", ctx.lib.rmgr_ssim_hip_luminance_device(
            ctx.handle, ctypes.c_void_p(dst.ptr), n, ctypes.c_void_p(src.ptr), step, stride, n, n))
        call(); ctx.synchronize()
        # correctness on a corner (integer arithmetic: exact)
        y = dst.download(np.uint8, (n, n))
        px = host[:64, :64 * step].reshape(64, 64, step).astype(np.uint32)
        want = ((px[..., 0] * 19595 + px[..., 1] * 38470 + px[..., 2] * 7471 + 32768) >> 16).astype(np.uint8)
        assert np.array_equal(y[:64, :64], want), label
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            call()
        ctx.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        alg = n * n * (step + 1)
        print("%-50s %dx%d: %.1f us per call, %.0f GB/s of %d algorithmic bytes per pixel = %.1f %% of 8 TB/s (%.1f Gpix/s)"
              % (label, n, n, dt * 1e6, alg / dt / 1e9, step + 1, alg / dt / 8e12 * 100, n * n / dt / 1e9))
        src.free(); dst.free()
    ctx.close()


if __name__ == "__main__":
    main()
