"""Full-size bit comparison of the GPU maps with the oracle's (run on the GPU box:
python tests/tools/fullsize_check.py [pairs=6] [size=4096]).  The pytest suite checks full-size runs through
known answers, properties and a 512x512 corner + all four borders; this tool compares EVERY pixel of every map,
in both bit-exact modes, plus the tolerance modes against the same oracle maps.
Last run (final round-2 kernels): 6 x 4096^2 and 2 x 8192^2, 0 differing pixels in exact / unfused;
fast mode within 1.1e-7 (global) / 9.5e-5 (per pixel) of the FMA-order maps."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import ssim_amd  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = ssim_amd.Context(0)
bad = 0
for i in range(pairs):
    a, b = oracle.synth_pair(size, size, 0x5EED + i)
    if i % 3 == 1:      # a differently textured pair: heavy noise
        rng = np.random.default_rng(i)
        b = np.clip(a.astype(np.int32) + rng.integers(-90, 91, a.shape), 0, 255).astype(np.uint8)
    for mode, fused in ((ssim_amd.MODE_EXACT, True), (ssim_amd.MODE_UNFUSED, False)):
        ov, _, om = oracle.ssim_f32(a, b, want_map=True, fused=fused, threads=oracle.oracle_lib().oracle_max_threads())
        ctx.set_mode(mode)
        v, m = ctx.ssim_planes(a, b, want_map=True)
        diff = int(np.count_nonzero(m.view(np.uint32) != om.view(np.uint32)))
        ulps = abs(int(np.float32(v).view(np.int32)) - int(np.float32(ov).view(np.int32)))
        print("pair %d %dx%d mode %d: %d differing pixels, global %.9f vs %.9f (%d ulp)" % (i, size, size, mode, diff, v, ov, ulps))
        bad += diff + (ulps > 1)
        if mode == ssim_amd.MODE_EXACT:
            ctx.set_mode(ssim_amd.MODE_FAST)
            vf, mf = ctx.ssim_planes(a, b, want_map=True)
            print("        fast mode vs FMA-order oracle: global |d| %.3g, per-pixel max |d| %.3g" % (abs(float(vf) - float(ov)), float(np.abs(mf.astype(np.float64) - om).max())))
            bad += (abs(float(vf) - float(ov)) > 1.5e-6) + (float(np.abs(mf.astype(np.float64) - om).max()) > 6.3e-4)
print("full-size check done, failures:", bad)
