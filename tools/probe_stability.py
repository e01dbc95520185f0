#!/usr/bin/env python3
"""Is rmgr_ssim_hip_probe_valu stable in a process that does other things?  Probes at 2 / 3 / 4 / 8 waves per SIMD interleaved with strip-kernel launches of different
lengths (long batches, single pairs, tiny images) and idle gaps; prints every sample with the clock it ran at.   usage: python3 tools/probe_stability.py [rounds=6]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = ssim_amd.Context(0)
    shapes = [(4096, 4096, 8), (4096, 4096, 1), (256, 256, 1), (1920, 1080, 32)]
    work = []
    for (w, h, n) in shapes:
        params = (ssim_amd.Params * n)()
        keep = []
        for i in range(n):
            da, db = ctx.alloc(w * h), ctx.alloc(w * h)
            ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
            keep += [da, db]
            params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
        work.append((w, h, n, params, ctx.alloc(8 * n), keep))
    samples = {2: [], 3: [], 4: [], 8: []}
    for r in range(rounds):
        for k, (w, h, n, params, sums, _) in enumerate(work):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.1:                 # 100 ms of this workload (single pairs: mostly launch latency, the chip half idle)
                ctx.enqueue_batch(params, n, sums.ptr)
                ctx.synchronize()
            if (r + k) % 3 == 0:
                time.sleep(0.2)                                   # an idle gap now and then
            line = []
            for waves in (2, 3, 4, 8):
                t, mhz, lo = ctx.probe_valu(waves, 0, 5, with_clock=True)
                samples[waves].append(t)
                line.append("%d waves %.2f T @ %.0f MHz (slowest XCD %.0f)" % (waves, t, mhz, lo))
            print("round %d after %4d x %dx%-4d: %s" % (r, n, w, h, "   ".join(line)))
            sys.stdout.flush()
    for waves, v in samples.items():
        print("# %d waves: min %.2f max %.2f spread %.1f %%" % (waves, min(v), max(v), 100.0 * (max(v) - min(v)) / min(v)))
    ctx.close()


if __name__ == "__main__":
    main()
