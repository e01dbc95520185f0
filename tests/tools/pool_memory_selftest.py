#!/usr/bin/env python3
"""Memory policy of the default contexts, in a process of its own (the pool reads $RMGR_SSIM_HIP_POOL_RETAIN_MB once): four threads make one
unchanged rmgr_ssim_compute_ssim() call each on an 8192 x 8192 pair WITH the map (134 MB + 268 MB of device staging per call), concurrently;
then the device's free memory (rmgr_ssim_hip_get_memory_info = hipMemGetInfo) is compared with the figure before the calls.

usage: pool_memory_selftest.py <expect: released|retained>      prints one JSON line; exit 0 iff every check holds
  released  the cap in force (default 256 MB per context, or 0) makes every context give its staging back when its call ends
  retained  RMGR_SSIM_HIP_POOL_RETAIN_MB=-1: the staging stays until rmgr_ssim_hip_trim_default_pool(), then goes
Results are checked bit for bit either way (SURVEY.md 8(d) known answer for the 8192^2 seed-0x5EED pair; the four maps identical)."""
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402

MB = 1 << 20


def main():
    expect = sys.argv[1]
    w = h = 8192
    a, b = synth.pair_numpy(w, h, synth.BASE_SEED)
    small_a, small_b = synth.pair_numpy(256, 256, synth.BASE_SEED)
    ssim_amd.compute_ssim(small_a, small_b)                     # the runtime's one-time allocations and the first default context
    ssim_amd.trim_default_pool()
    free0, total = ssim_amd.memory_info()
    maps = [np.zeros((h, w), np.float32) for _ in range(4)]
    vals = [None] * 4

    def call(i):
        vals[i] = ssim_amd.compute_ssim(a, b, out_map=maps[i])[0]
    ts = [threading.Thread(target=call, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    contexts, limit = ssim_amd.default_pool()
    free1, _ = ssim_amd.memory_info()
    held_dev, held_pin, cap = ssim_amd.default_pool_memory()
    ssim_amd.trim_default_pool()
    free2, _ = ssim_amd.memory_info()
    trimmed_dev, trimmed_pin, _ = ssim_amd.default_pool_memory()
    # the engine still works after a trim, and gives the same bits
    v_after, m_after = ssim_amd.compute_ssim(a, b, want_map=True)
    digest = [hashlib.sha256(m.tobytes()).hexdigest() for m in maps]
    ok = True
    why = []

    def check(cond, what):
        nonlocal ok
        if not cond:
            ok = False
            why.append(what)
    check(all(int(np.float32(v).view(np.uint32)) == 0x3f64b5b4 for v in vals), "global values %r" % vals)
    check(len(set(digest)) == 1 and hashlib.sha256(m_after.tobytes()).hexdigest() == digest[0], "maps differ")
    check(int(np.float32(v_after).view(np.uint32)) == 0x3f64b5b4, "value after the trim")
    check(abs(float(maps[0].mean(dtype=np.float64)) - float(vals[0])) < 1e-6, "map mean")
    check(contexts >= 2, "the calls did not overlap: %d contexts" % contexts)
    per_call = 2 * w * h + 4 * w * h
    if expect == "released":
        check(free0 - free1 <= 64 * MB, "free memory %d MB below the pre-call figure without a trim" % ((free0 - free1) // MB))
        check(held_dev + held_pin <= contexts * cap and held_dev <= 64 * MB, "pool reports %d MB device staging held" % (held_dev // MB))
    else:
        check(held_dev >= contexts * per_call * 0.9, "pool reports only %d MB device staging for %d contexts" % (held_dev // MB, contexts))
        check(free0 - free1 >= contexts * per_call * 0.9, "free memory fell by only %d MB" % ((free0 - free1) // MB))
    check(free0 - free2 <= 64 * MB, "after the trim free memory is still %d MB below the pre-call figure" % ((free0 - free2) // MB))
    check(trimmed_dev == 0 and trimmed_pin == 0, "pool reports %d / %d bytes after the trim" % (trimmed_dev, trimmed_pin))
    print(json.dumps({"ok": ok, "why": why, "expect": expect, "contexts": contexts, "limit": limit, "cap_mb": None if cap == 2 ** 64 - 1 else cap // MB,
                      "free_before_mb": free0 // MB, "free_after_calls_mb": free1 // MB, "free_after_trim_mb": free2 // MB,
                      "held_device_mb": held_dev // MB, "held_pinned_mb": held_pin // MB}))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
