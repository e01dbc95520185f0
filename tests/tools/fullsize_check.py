"""Full-size bit comparison of the GPU maps with the oracle's (run on the GPU box:
python tests/tools/fullsize_check.py [pairs=6] [size=4096]).  The pytest suite checks full-size runs through
known answers, properties and a 512x512 corner + all four borders; this tool compares EVERY pixel of every map,
in both bit-exact modes, plus the tolerance modes against the same oracle maps.
Round 3: MODE_FAST is the hybrid (reference-order E planes, separable mu planes) and is held to north_star's FMA-relative
tolerance on every pixel here too; MODE_SEPARABLE's distance from the FMA-order maps is printed (its contract is against the
exact value, tests/test_gpu_modes.py checks it at 4096^2).  Results of the last run: profiles/r03_fullsize_check.txt."""
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
            ctx.set_mode(ssim_amd.MODE_SEPARABLE)
            vs, ms = ctx.ssim_planes(a, b, want_map=True)
            print("        separable mode vs FMA-order oracle (reported only): global |d| %.3g, per-pixel max |d| %.3g" % (abs(float(vs) - float(ov)), float(np.abs(ms.astype(np.float64) - om).max())))
print("full-size check done, failures:", bad)
