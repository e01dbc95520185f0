#!/usr/bin/env python3
"""Exercise an ASan/UBSan build of the oracle's C restatement (make sanitize): edge sizes, both rounding orders, threaded
and serial, the naive double oracle, the synthetic generator.  Run under LD_PRELOAD=libasan.so; any report aborts."""
import sys

import numpy as np

sys.path.insert(0, sys.argv[2] if len(sys.argv) > 2 else ".")
import oracle  # noqa: E402

oracle.ORACLE_SO = sys.argv[1]
rng = np.random.default_rng(1)
for (h, w) in ((1, 1), (1, 37), (37, 1), (10, 10), (11, 11), (63, 255), (65, 257), (200, 300), (360, 640)):
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    b = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = None
    for fused in (True, False):
        for th in (1, 4):
            v, s, m = oracle.ssim_f32(a, b, want_map=True, fused=fused, threads=th)
            if fused:
                assert ref is None or (np.array_equal(ref, m)), (h, w)
                ref = m
    oracle.ssim_naive_f64(a, b, want_map=True, threads=3)
a, b = oracle.synth_pair(301, 77, 5)
a2, b2 = oracle.synth_pair_numpy(301, 77, 5)
assert np.array_equal(a, a2) and np.array_equal(b, b2)
print("oracle under ASan/UBSan: ok")
