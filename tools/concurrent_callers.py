#!/usr/bin/env python3
"""N host threads looping the UNCHANGED drop-in call rmgr_ssim_compute_ssim() on pageable host images (ctypes releases the GIL for
the duration of a call): aggregate Mpix/s against one thread's.  The reference's function is re-entrant and parallel across callers
(src/ssim.cpp:933-1106: no global state); here every ctx == NULL call leases one of the default contexts ($RMGR_SSIM_HIP_POOL).

usage: python tools/concurrent_callers.py [size=4096] [map=1] [threads=1,2,4,6] [seconds=2] [--json]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def run(size, want_map, nthreads, seconds):
    pairs = [synth.pair_numpy(size, size, synth.BASE_SEED + t) for t in range(nthreads)]
    maps = [np.zeros((size, size), np.float32) if want_map else None for _ in range(nthreads)]
    vals, counts = [None] * nthreads, [0] * nthreads
    for t in range(nthreads):                               # touch everything once (page faults, context creation) outside the timing
        vals[t] = ssim_amd.compute_ssim(pairs[t][0], pairs[t][1], out_map=maps[t])[0]
    start = threading.Barrier(nthreads + 1)
    stop_at = [0.0]

    def worker(t):
        start.wait()
        while time.perf_counter() < stop_at[0]:
            v = ssim_amd.compute_ssim(pairs[t][0], pairs[t][1], out_map=maps[t])[0]
            assert np.float32(v).view(np.uint32) == np.float32(vals[t]).view(np.uint32)
            counts[t] += 1

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    stop_at[0] = time.perf_counter() + seconds
    t0 = time.perf_counter()
    start.wait()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    return sum(counts) * size * size / dt / 1e6, sum(counts), dt


def main():
    argv = [a for a in sys.argv if a != "--json"]
    arg = lambda i, d: argv[i] if len(argv) > i else d
    size, want_map = int(arg(1, 4096)), int(arg(2, 1))
    threads = [int(v) for v in arg(3, "1,2,4,6").split(",")]
    seconds = float(arg(4, 2))
    base = None
    out = {}
    for n in threads:
        mpix, calls, dt = run(size, want_map, n, seconds)
        base = base or mpix
        out[str(n)] = {"mpix_s": round(mpix, 1), "calls": calls, "vs_one_thread": round(mpix / base, 3)}
        if "--json" not in sys.argv:
            print("%d caller thread(s): %8.1f Mpix/s aggregate (%d calls in %.2f s), %.2fx one thread; default contexts: %d of %d"
                  % (n, mpix, calls, dt, mpix / base, ssim_amd.default_pool()[0], ssim_amd.default_pool()[1]), flush=True)
    if "--json" in sys.argv:
        print(json.dumps({"size": size, "map": bool(want_map), "seconds_per_point": seconds, "threads": out,
                          "default_contexts": ssim_amd.default_pool()[0], "pool_limit": ssim_amd.default_pool()[1]}))


if __name__ == "__main__":
    main()
