"""GPU: randomized layouts.  ImgParams allows any step/stride (interleaved, padded, bottom-up, mirrored,
column-major) and the map any ssimStep/ssimStride (reference include/rmgr/ssim.h:481-516); every random
combination must give the oracle's per-pixel map bit for bit, in both bit-exact modes and both kernels."""
import ctypes

import numpy as np
import pytest

import ssim_amd
from conftest import ulp_diff

pytestmark = pytest.mark.gpu


def make_layout(rng, img):
    """Embed an H x W uint8 image into a larger buffer with a random step/stride; returns (buffer, offset of
    pixel (0,0), step, stride)."""
    h, w = img.shape
    step = int(rng.integers(1, 5))
    row = w * step + int(rng.integers(0, 9))
    flip_y = bool(rng.integers(0, 2))
    flip_x = bool(rng.integers(0, 2))
    transpose = bool(rng.integers(0, 4) == 0)
    if transpose:                      # column-major: step walks down a column of the buffer
        row = h * step + int(rng.integers(0, 9))
        buf = rng.integers(0, 256, (w * row + 16,), dtype=np.uint8)
        view = np.lib.stride_tricks.as_strided(buf[4:], shape=(h, w), strides=(step, row))
        view[...] = img
        return buf, 4, row, step
    buf = rng.integers(0, 256, (h * row + 16,), dtype=np.uint8)
    view = np.lib.stride_tricks.as_strided(buf[4:], shape=(h, w), strides=(row, step))
    src = img[::-1] if flip_y else img
    src = src[:, ::-1] if flip_x else src
    view[...] = src
    off, st, sd = 4, step, row
    if flip_y:
        off += (h - 1) * row
        sd = -row
    if flip_x:
        off += (w - 1) * step
        st = -step
    return buf, off, st, sd


def test_random_layouts_bit_exact(gpu_ctx, oracle):
    rng = np.random.default_rng(20260101)
    lib = oracle.oracle_lib()
    for case in range(60):
        h, w = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, (h, w)), 0, 255).astype(np.uint8) if rng.integers(0, 3) else rng.integers(0, 256, (h, w), dtype=np.uint8)
        fused = bool(rng.integers(0, 4))                      # mostly the FMA order, sometimes the unfused one
        variant, rows = int(rng.integers(0, 4)), int(rng.choice([0, 0, 1, 3, 16, 50]))   # 0 default, 1 one column, 2 two columns, 3 two columns with early row sums
        ov, osum, om = oracle.ssim_f32(a, b, want_map=True, fused=fused)

        ba, oa, sa, da_ = make_layout(rng, a)
        bb, ob, sb, db_ = make_layout(rng, b)
        # the oracle on the same strided views (checks that both sides address the same pixels)
        out, s = ctypes.c_float(), ctypes.c_double()
        rc = lib.oracle_ssim_f32(ctypes.byref(out), ctypes.byref(s), w, h, ctypes.c_void_p(ba.ctypes.data + oa), sa, da_,
                                 ctypes.c_void_p(bb.ctypes.data + ob), sb, db_, None, 0, 0, int(fused), 1)
        assert rc == 0 and s.value == osum, case

        mstep = int(rng.integers(1, 4))
        mrow = w * mstep + int(rng.integers(0, 5))
        mflip = bool(rng.integers(0, 2))
        mbuf = np.full((h * mrow + 8,), -3.0, np.float32)
        keep = []
        try:
            da, db, dm = gpu_ctx.upload(ba), gpu_ctx.upload(bb), gpu_ctx.upload(mbuf)
            keep += [da, db, dm]
            moff = 2 + ((h - 1) * mrow if mflip else 0)
            p = ssim_amd.make_params(w, h, da.ptr + oa, sa, da_, db.ptr + ob, sb, db_, dm.ptr + 4 * moff, mstep, -mrow if mflip else mrow)
            gpu_ctx.set_mode(ssim_amd.MODE_EXACT if fused else ssim_amd.MODE_UNFUSED)
            gpu_ctx.set_tuning(rows, variant)
            v = gpu_ctx.compute_device(p)
            got = dm.download(np.float32, mbuf.shape)
        finally:
            for d in keep:
                d.free()
        view = np.lib.stride_tricks.as_strided(got[2:], shape=(h, w), strides=(4 * mrow, 4 * mstep))
        gm = view[::-1] if mflip else view
        bad = np.count_nonzero(np.ascontiguousarray(gm).view(np.uint32) != om.view(np.uint32))
        assert bad == 0, (case, w, h, sa, da_, sb, db_, mstep, mflip, variant, rows, bad)
        assert ulp_diff(v, ov) <= 1, (case, float(v), float(ov))
        # nothing outside the map's own elements was written
        mask = np.ones(mbuf.shape, bool)
        idx = (2 + np.arange(h)[:, None] * mrow + np.arange(w)[None, :] * mstep).ravel()
        mask[idx] = False
        assert np.all(got[mask] == -3.0), case
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    gpu_ctx.set_tuning(0, 0)


@pytest.mark.parametrize("mode", [ssim_amd.MODE_EXACT, ssim_amd.MODE_UNFUSED])
def test_division_corner_statistics_bit_exact(gpu_ctx, oracle, mode):
    """The two-column kernel divides with the compiler's IEEE sequence minus v_div_scale / v_div_fixup (operands
    are provably in range, ssim_kernels.hip div_inrange_*).  Exercise the corners of that range: anti-correlated
    textures whose covariance sweeps 2*sAB + c2 through zero (tiny and sign-changing numerators), flat images
    (variances exactly zero), saturated black / white, maximal contrast.  The oracle divides on the CPU."""
    gpu_ctx.set_mode(mode)
    rng = np.random.default_rng(4242)
    h, w = 200, 640
    xx = np.arange(w)[None, :].repeat(h, 0)
    try:
        seen_small = seen_neg = 0
        for kk in (1.0, 0.5, 2.0, 1.0):
            amp = xx / w * (12.0 / np.sqrt(kk))
            t = rng.choice([-1.0, 1.0], (h, w))
            base = int(rng.integers(60, 196))
            a = np.clip(np.rint(base + amp * t), 0, 255).astype(np.uint8)
            b = np.clip(np.rint(base - kk * amp * t), 0, 255).astype(np.uint8)
            ov, _, om = oracle.ssim_f32(a, b, want_map=True, fused=(mode == ssim_amd.MODE_EXACT))
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert np.array_equal(m.view(np.uint32), om.view(np.uint32)), kk
            assert ulp_diff(v, ov) <= 1
            seen_small += int((np.abs(om) < 1e-3).sum())
            seen_neg += int((om < 0).sum())
        assert seen_small > 100 and seen_neg > 1000, (seen_small, seen_neg)     # the sweep really crossed zero
        for a, b in [(np.zeros((40, 300), np.uint8), np.full((40, 300), 255, np.uint8)),
                     (np.full((40, 300), 255, np.uint8), np.full((40, 300), 255, np.uint8)),
                     (rng.choice([0, 255], (90, 400)).astype(np.uint8), rng.choice([0, 255], (90, 400)).astype(np.uint8)),
                     ((np.indices((64, 256)).sum(0) % 2 * 255).astype(np.uint8), (255 - np.indices((64, 256)).sum(0) % 2 * 255).astype(np.uint8))]:
            ov, _, om = oracle.ssim_f32(a, b, want_map=True, fused=(mode == ssim_amd.MODE_EXACT))
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert np.array_equal(m.view(np.uint32), om.view(np.uint32))
            assert ulp_diff(v, ov) <= 1
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
