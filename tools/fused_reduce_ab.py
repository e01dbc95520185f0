#!/usr/bin/env python3
"""Interleaved A/B of the two executors of the per-image reduction (ssim_kernels.hip image_sum()): the last strip of an
image finishing it inside the strip kernel (default where strips_finish_images() says so) against the separate
ssim_reduce_kernel launch ($RMGR_SSIM_HIP_FUSED_REDUCE=0), in one process, two contexts on one device.

usage: python3 tools/fused_reduce_ab.py [rounds=7]
Per workload: wall time per call of the blocking device-pointer call (one pair), of back-to-back enqueues, and the
stream time per launch of a batch; the per-image fp64 sums of both executors must be the same bits.
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


def make_ctx(fused):
    os.environ["RMGR_SSIM_HIP_FUSED_REDUCE"] = "1" if fused else "0"
    return ssim_amd.Context(0)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    ctxs = {"fused": make_ctx(True), "separate": make_ctx(False)}
    print(ctxs["fused"].describe())
    for (w, h, pairs, mode) in ((4096, 4096, 1, 0), (1920, 1080, 1, 0), (256, 256, 1, 0), (4096, 4096, 1, 4), (4096, 4096, 8, 0), (1920, 1080, 32, 0),
                                (4096, 4096, 32, 0), (1920, 1080, 128, 0), (8192, 8192, 1, 0)):
        base = ctxs["fused"]
        imgs = base.alloc(2 * w * h * pairs)
        params = (ssim_amd.Params * pairs)()
        for i in range(pairs):
            a = imgs.ptr + 2 * w * h * i
            base.synth_pair(a, w, a + w * h, w, w, h, synth.BASE_SEED + i)
            params[i] = ssim_amd.make_params(w, h, a, 1, w, a + w * h, 1, w)
        base.synchronize()
        sums = {k: c.alloc(8 * pairs) for k, c in ctxs.items()}
        t_block = {k: [] for k in ctxs}
        t_pipe = {k: [] for k in ctxs}
        bits = {}
        for c in ctxs.values():
            c.set_mode(mode)
        for k, c in ctxs.items():      # warm both, settle the clock
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.05:
                c.enqueue_batch(params, pairs, sums[k].ptr)
                c.synchronize()
        n_block = 200 if pairs == 1 else 20
        for _ in range(rounds):
            for k, c in ctxs.items():
                if pairs == 1:
                    for _ in range(5):
                        c.compute_device(params[0])
                    t = time.perf_counter()
                    for _ in range(n_block):
                        c.compute_device(params[0])
                    t_block[k].append((time.perf_counter() - t) / n_block)
                else:
                    t = time.perf_counter()
                    for _ in range(n_block):
                        c.enqueue_batch(params, pairs, sums[k].ptr)
                        c.synchronize()
                    t_block[k].append((time.perf_counter() - t) / n_block)
                t = time.perf_counter()
                for _ in range(n_block):
                    c.enqueue_batch(params, pairs, sums[k].ptr)
                c.synchronize()
                t_pipe[k].append((time.perf_counter() - t) / n_block)
                bits[k] = sums[k].download(np.float64, (pairs,)).view(np.uint64).copy()
        same = np.array_equal(bits["fused"], bits["separate"])
        px = float(w) * h * pairs
        line = "%d x %dx%d mode %d:" % (pairs, w, h, mode)
        for k in ctxs:
            mb, mp = statistics.median(t_block[k]), statistics.median(t_pipe[k])
            line += "  %s: blocking %.1f us (%.1f Gpix/s), back to back %.1f us (%.1f Gpix/s)" % (k, mb * 1e6, px / mb / 1e9, mp * 1e6, px / mp / 1e9)
        mb = {k: statistics.median(t_block[k]) for k in ctxs}
        mp = {k: statistics.median(t_pipe[k]) for k in ctxs}
        line += "  | fused vs separate: blocking %+.1f %%, back to back %+.1f %% | sums %s" % (
            (mb["separate"] / mb["fused"] - 1) * 100, (mp["separate"] / mp["fused"] - 1) * 100, "bit-identical" if same else "DIFFER")
        print(line)
        sys.stdout.flush()
        for c in ctxs.values():
            c.set_mode(0)
        for s in sums.values():
            s.free()
        imgs.free()
        assert same
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
