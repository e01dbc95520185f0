"""CPU, build container only: the oracle against the REAL reference kernels in oracle/_ref.

oracle/_ref is compiled from /root/reference (oracle/Makefile).  Where it is absent (a machine
without the reference tree and without the prebuilt .so) these tests skip; the golden-vector
tests in test_oracle_golden.py still pin the oracle there.
"""
import numpy as np
import pytest

import oracle as o
from conftest import f32_hex

pytestmark = pytest.mark.skipif(not o.have_ref(), reason="oracle/_ref/libssim_ref.so not built")


@pytest.mark.parametrize("impl,fused", [(5, True), (4, False)])
def test_random_images_bit_exact(impl, fused):
    rng = np.random.default_rng(1234)
    for (h, w) in ((1, 1), (5, 7), (63, 255), (64, 256), (65, 257), (130, 600), (300, 301)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        kind = rng.integers(0, 3)
        if kind == 0:
            b = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 1:
            b = np.clip(a.astype(np.int32) + rng.integers(-9, 10, (h, w)), 0, 255).astype(np.uint8)
        else:
            b = (a // 2 + 40).astype(np.uint8)
        v, s, m = o.ssim_f32(a, b, want_map=True, fused=fused)
        rv, rs, rm = o.ref_ssim(a, b, want_map=True, impl=impl)
        assert f32_hex(v) == f32_hex(rv), (h, w)
        assert s == rs, (h, w)
        assert np.array_equal(m.view(np.uint32), rm.view(np.uint32)), (h, w)


def test_flat_and_extreme_images_bit_exact():
    for va, vb in ((0, 0), (255, 255), (0, 255), (17, 18)):
        a = np.full((70, 300), va, np.uint8)
        b = np.full((70, 300), vb, np.uint8)
        v, s, m = o.ssim_f32(a, b, want_map=True)
        rv, rs, rm = o.ref_ssim(a, b, want_map=True)
        assert f32_hex(v) == f32_hex(rv) and s == rs
        assert np.array_equal(m.view(np.uint32), rm.view(np.uint32))


def test_naive_f64_identical():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (97, 131), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    v, _, m = o.ssim_naive_f64(a, b, want_map=True)
    rv, rm = o.ref_naive_f64(a, b, want_map=True)
    assert v == rv and np.array_equal(m, rm)


def test_reference_openmp_path_same_float():
    a, b = o.synth_pair(1920, 1080, 0x5EED)
    v1, _, _ = o.ref_ssim(a, b, threads=1)
    v8, _, _ = o.ref_ssim(a, b, threads=8)
    ov, _, _ = o.ssim_f32(a, b, threads=8)
    assert f32_hex(v1) == f32_hex(v8) == f32_hex(ov) == "0x3f64bb1f"
