"""GPU: the two non-default arithmetic modes.

MODE_FAST   separable fp32 blur; north_star tolerance versus the FMA reference:
            global |d| <= 1.5e-6, per-pixel |d| <= 6.3e-4 (README.md:89-92 of the reference).
MODE_DOUBLE RMGR_SSIM_USE_DOUBLE semantics (BASELINE.json config 5): per-pixel error versus
            tests/ssim_naive.h <= 1e-7, here against the committed naive maps and the oracle's
            restatement of it.
"""
import os

import numpy as np
import pytest

import ssim_amd
from conftest import GOLDEN, image_entries, load_pair

pytestmark = pytest.mark.gpu

GLOBAL_TOL = 1.5e-6
PIXEL_TOL = 6.3e-4


@pytest.mark.parametrize("variant", [2, 1])
def test_fast_mode_within_documented_tolerance(gpu_ctx, manifest, oracle, variant):
    gpu_ctx.set_mode(ssim_amd.MODE_FAST)
    gpu_ctx.set_tuning(0, variant)
    worst_g = worst_p = 0.0
    try:
        for name in image_entries(manifest):
            ent = manifest[name]
            a, b = load_pair(ent)
            _, _, ref = oracle.ssim_f32(a, b, want_map=True)      # == the FMA reference map (test_oracle_golden)
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            dg = abs(float(v) - float(ent["fma"]["ssim"]))
            dp = float(np.abs(m.astype(np.float64) - ref.astype(np.float64)).max())
            worst_g, worst_p = max(worst_g, dg), max(worst_p, dp)
            assert dg <= GLOBAL_TOL, (name, dg)
            assert dp <= PIXEL_TOL, (name, dp)
        for seed in (0x5EED, 0x5EEE):
            a, b = oracle.synth_pair(1920, 1080, seed)
            ov, _, om = oracle.ssim_f32(a, b, want_map=True, threads=8)
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert abs(float(v) - float(ov)) <= GLOBAL_TOL
            assert float(np.abs(m.astype(np.float64) - om.astype(np.float64)).max()) <= PIXEL_TOL
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        gpu_ctx.set_tuning(0, 0)
    print("fast mode worst global %.3g, worst per-pixel %.3g" % (worst_g, worst_p))


def test_fast_mode_full_size_4k(gpu_ctx, oracle):
    """MODE_FAST at BASELINE's 4096^2 size: every pixel and the global value inside the tolerance."""
    gpu_ctx.set_mode(ssim_amd.MODE_FAST)
    try:
        a, b = oracle.synth_pair(4096, 4096, 0x5EED)
        ov, _, om = oracle.ssim_f32(a, b, want_map=True, threads=oracle.oracle_lib().oracle_max_threads())
        v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
        assert abs(float(v) - float(ov)) <= GLOBAL_TOL
        assert float(np.abs(m.astype(np.float64) - om.astype(np.float64)).max()) <= PIXEL_TOL
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)


def test_double_mode_vs_naive_oracle(gpu_ctx, manifest, oracle):
    gpu_ctx.set_mode(ssim_amd.MODE_DOUBLE)
    try:
        for name in image_entries(manifest):
            ent = manifest[name]
            a, b = load_pair(ent)
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert abs(float(v) - float(ent["naive_f64"]["ssim"])) <= 6e-8 + 1e-9, name   # float rounding of the result
            if "map" in ent["naive_f64"]:
                nmap = np.load(os.path.join(GOLDEN, ent["naive_f64"]["map"]))
            else:
                _, _, nmap = oracle.ssim_naive_f64(a, b, want_map=True, threads=8)
            assert float(np.abs(m.astype(np.float64) - nmap).max()) <= 1e-7, name
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)


def test_double_mode_4k_config5(gpu_ctx, oracle, manifest):
    """BASELINE.json config 5 at full size: 4096^2, fp64 internals, map on; per-pixel vs the naive
    oracle on sampled windows (the naive 121-tap gather is too slow for 16.7 Mpx on the CPU) and
    the global value vs the reference's naive<double> known answer."""
    gpu_ctx.set_mode(ssim_amd.MODE_DOUBLE)
    keep = []
    try:
        a, b = oracle.synth_pair(4096, 4096, 0x5EED)
        da, db, dm = gpu_ctx.upload(a), gpu_ctx.upload(b), gpu_ctx.alloc(4 * 4096 * 4096)
        keep += [da, db, dm]
        p = ssim_amd.make_params(4096, 4096, da.ptr, 1, 4096, db.ptr, 1, 4096, dm.ptr, 1, 4096)
        v = gpu_ctx.compute_device(p)
        assert abs(float(v) - 0.893428737869049) <= 6e-8 + 1e-9     # SURVEY.md 8(d) KAT, naive<double>
        m = dm.download(np.float32, (4096, 4096))
        for (y0, x0) in ((0, 0), (0, 4096 - 256), (4096 - 256, 0), (1900, 2000), (4096 - 256, 4096 - 256)):
            ys, xs = slice(y0, y0 + 256), slice(x0, x0 + 256)
            _, _, nm = oracle.ssim_naive_f64(np.ascontiguousarray(a[ys, xs]), np.ascontiguousarray(b[ys, xs]), want_map=True, threads=8)
            iy = slice(0 if y0 == 0 else 5, 256 if y0 + 256 == 4096 else 251)
            ix = slice(0 if x0 == 0 else 5, 256 if x0 + 256 == 4096 else 251)
            assert float(np.abs(m[ys, xs][iy, ix].astype(np.float64) - nm[iy, ix]).max()) <= 1e-7
    finally:
        for d in keep:
            d.free()
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)


def test_process_wide_mode_switch(manifest):
    """RMGR_SSIM_HIP_MODE selects the arithmetic of the unchanged drop-in call (what RMGR_SSIM_USE_DOUBLE
    does at build time in the reference, src/ssim_internal.h:26-37): mode 2 must return the double-mode value."""
    import subprocess
    import sys
    ent = manifest["einstein_jpg"]
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import ssim_amd; "
            "a=np.fromfile(%r,np.uint8).reshape(256,256); b=np.fromfile(%r,np.uint8).reshape(256,256); "
            "print(repr(float(ssim_amd.compute_ssim(a,b)[0])))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(GOLDEN, ent["a"]), os.path.join(GOLDEN, ent["b"])))
    out = {}
    for mode in ("0", "2", "3"):
        env = dict(os.environ, RMGR_SSIM_HIP_MODE=mode)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-800:]
        out[mode] = float(r.stdout.split()[-1])
    assert np.float32(out["0"]) == np.float32(float(ent["fma"]["ssim"]))
    assert np.float32(out["3"]) == np.float32(float(ent["avx"]["ssim"]))
    assert abs(out["2"] - float(ent["naive_f64"]["ssim"])) <= 6e-8 + 1e-9
    assert out["2"] != out["0"]
