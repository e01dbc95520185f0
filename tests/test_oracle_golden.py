"""CPU: the oracle (oracle/ssim_oracle.c) against the committed golden vectors.

Golden vectors = what the reference itself produced (tests/tools/make_fixtures.py, real FMA/AVX kernel
objects + tests/ssim_naive.h) on the reference's own test images, plus the quad-precision
constants of the reference's tests (tests/rmgr-ssim-tests.cpp:352-360).
"""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, f32_hex, image_entries, load_pair


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def test_kernel_taps_generic_equals_literal(oracle):
    """The SIMD paths' literal k21 table is the generic path's runtime kernel."""
    lit = (ctypes.c_float * 21)()
    gen = (ctypes.c_float * 21)()
    oracle.oracle_lib().oracle_kernel21_f32(lit)
    oracle.oracle_lib().oracle_generic_kernel21_f32(gen)
    assert list(lit) == list(gen)
    k = (ctypes.c_double * 121)()
    oracle.oracle_lib().oracle_kernel121_f64(k)
    assert abs(sum(k) - 1.0) < 1e-15
    k = np.array(k).reshape(11, 11)
    assert np.array_equal(k, k.T) and np.array_equal(k, k[::-1]) and np.array_equal(k, k[:, ::-1])


def test_fixture_names(manifest):
    names = image_entries(manifest)
    assert len(names) == 18
    assert sum(n.startswith("einstein_") for n in names) == 6


@pytest.mark.parametrize("path", ["fma", "avx"])
def test_oracle_matches_reference_outputs(oracle, manifest, path):
    """Global float bit-equal, fp64 sum equal, per-pixel map bit-equal (sha256) on every fixture."""
    for name in image_entries(manifest):
        ent = manifest[name]
        a, b = load_pair(ent)
        v, s, m = oracle.ssim_f32(a, b, want_map=True, fused=(path == "fma"))
        assert f32_hex(v) == ent[path]["ssim_hex"], name
        assert repr(s) == ent[path]["sum"], name
        assert sha(m) == ent[path]["map_sha256"], name
        if "map" in ent[path]:
            ref = np.load(os.path.join(GOLDEN, ent[path]["map"]))
            assert np.array_equal(ref.view(np.uint32), m.view(np.uint32)), name


def test_oracle_naive_f64_matches_reference(oracle, manifest):
    for name in image_entries(manifest):
        ent = manifest[name]
        a, b = load_pair(ent)
        v, _, m = oracle.ssim_naive_f64(a, b, want_map=True)
        assert repr(v) == ent["naive_f64"]["ssim"], name
        assert sha(m) == ent["naive_f64"]["map_sha256"], name


def test_reference_test_goldens(oracle, manifest):
    """tests/rmgr-ssim-tests.cpp: naive oracle within REF_TOLERANCE 1e-13 of the hard-coded constants
    (:72, :286) and the float paths within GLOBAL_TOLERANCE 2e-6 / PIXEL_TOLERANCE 1e-3 of it (:98-104)."""
    for name in image_entries(manifest):
        ent = manifest[name]
        if "reference_test_golden" not in ent:
            continue
        gold = float(ent["reference_test_golden"])
        a, b = load_pair(ent)
        nv, _, nmap = oracle.ssim_naive_f64(a, b, want_map=True)
        assert abs(nv - gold) < 1e-13, name
        for fused in (True, False):
            v, _, m = oracle.ssim_f32(a, b, want_map=True, fused=fused)
            assert abs(float(v) - gold) < 2e-6, name
            assert np.abs(m.astype(np.float64) - nmap).max() < 1e-3, name


def test_synthetic_generator_kats(oracle, manifest):
    """generator checksums, C and numpy generators agree."""
    for key, ent in manifest["_synthetic"].items():
        w, h, seed = ent["width"], ent["height"], ent["seed"]
        if w * h > 1920 * 1080:
            continue
        a, b = oracle.synth_pair(w, h, seed)
        assert int(a.sum(dtype=np.int64)) == ent["sumA"] and int(b.sum(dtype=np.int64)) == ent["sumB"], key
        assert a[0, :4].tolist() == ent["first4A"] and b[0, :4].tolist() == ent["first4B"], key
        a2, b2 = oracle.synth_pair_numpy(w, h, seed)
        assert np.array_equal(a, a2) and np.array_equal(b, b2), key
    a, b = oracle.synth_pair(256, 256, 0x5EED)
    assert a[0, :4].tolist() == [45, 59, 51, 6] and b[0, :4].tolist() == [51, 58, 48, 5]


def test_synthetic_ssim_kats(oracle, manifest):
    """Known answers of the FMA reference on the synthetic pairs (256^2 and 1080p here; 4K/8K on the GPU box)."""
    for key in ("256x256_5eed", "1920x1080_5eed"):
        ent = manifest["_synthetic"][key]
        a, b = oracle.synth_pair(ent["width"], ent["height"], ent["seed"])
        v, _, _ = oracle.ssim_f32(a, b, threads=8)
        assert f32_hex(v) == ent["fma"]["ssim_hex"], key
    ent = manifest["_synthetic"]["256x256_5eed"]
    a, b = oracle.synth_pair(256, 256, 0x5EED)
    nv, _, _ = oracle.ssim_naive_f64(a, b, threads=8)
    assert repr(nv) == ent["naive_f64"]


def test_oracle_threads_do_not_change_result(oracle):
    a, b = oracle.synth_pair(700, 333, 7)
    v1, s1, m1 = oracle.ssim_f32(a, b, want_map=True, threads=1)
    v8, s8, m8 = oracle.ssim_f32(a, b, want_map=True, threads=8)
    assert s1 == s8 and f32_hex(v1) == f32_hex(v8) and np.array_equal(m1, m8)


def test_oracle_identity_and_edges(oracle):
    rng = np.random.default_rng(1)
    for (h, w) in ((1, 1), (1, 37), (37, 1), (3, 200), (10, 10), (11, 11), (64, 256), (65, 257)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        v, _, m = oracle.ssim_f32(a, a, want_map=True)
        assert f32_hex(v) == f32_hex(1.0), (h, w)
        assert np.all(m == np.float32(1.0)), (h, w)


def test_oracle_strided_layouts(oracle, manifest):
    """Interleaved RGB (step 3) and bottom-up (negative stride) address the same pixels."""
    il = manifest["_interleaved"]
    w, h, ch = il["width"], il["height"], il["channels"]
    A = np.fromfile(os.path.join(GOLDEN, il["a"]), np.uint8).reshape(h, w, ch)
    B = np.fromfile(os.path.join(GOLDEN, il["b"]), np.uint8).reshape(h, w, ch)
    lib = oracle.oracle_lib()
    for c, name in enumerate(il["per_channel"]):
        out = ctypes.c_float()
        s = ctypes.c_double()
        rc = lib.oracle_ssim_f32(ctypes.byref(out), ctypes.byref(s), w, h,
                                 ctypes.c_void_p(A.ctypes.data + c), ch, w * ch,
                                 ctypes.c_void_p(B.ctypes.data + c), ch, w * ch, None, 0, 0, 1, 1)
        assert rc == 0 and f32_hex(out.value) == manifest[name]["fma"]["ssim_hex"]
        # bottom-up view of the vertically flipped images must give the same pixels
        Af, Bf = A[::-1].copy(), B[::-1].copy()
        rc = lib.oracle_ssim_f32(ctypes.byref(out), ctypes.byref(s), w, h,
                                 ctypes.c_void_p(Af.ctypes.data + (h - 1) * w * ch + c), ch, -w * ch,
                                 ctypes.c_void_p(Bf.ctypes.data + (h - 1) * w * ch + c), ch, -w * ch, None, 0, 0, 1, 1)
        assert rc == 0 and f32_hex(out.value) == manifest[name]["fma"]["ssim_hex"]


@pytest.mark.parametrize("set_name", ["bbb255", "bbb257", "bbb360", "bbb1080"])
def test_oracle_matches_reference_on_the_reference_test_sets(oracle, refsets, set_name):
    """The reference's own full-size image sets (tests/rmgr-ssim-tests.cpp:388-465: 33 (quality, channel) pairs each):
    the C restatement reproduces the real FMA / AVX kernels' float maps bit for bit (sha256), their global floats and
    fp64 sums, and the reference's tests/ssim_naive.h double map and value -- at 640x360 and 1920x1080, not only on crops."""
    from conftest import refset_pair
    pairs = refsets[set_name]["pairs"]
    assert len(pairs) == 33
    for key in sorted(pairs):
        ent = pairs[key]
        a, b = refset_pair(ent)
        for path, fused in (("fma", True), ("avx", False)):
            v, s, m = oracle.ssim_f32(a, b, want_map=True, fused=fused, threads=1)     # serial tile order: the fp64 sum is pinned too
            assert f32_hex(v) == ent[path]["ssim_hex"], (set_name, key, path)
            assert repr(s) == ent[path]["sum"], (set_name, key, path)
            assert sha(m) == ent[path]["map_sha256"], (set_name, key, path)
        nv, _, nm = oracle.ssim_naive_f64(a, b, want_map=True, threads=8)
        assert repr(nv) == ent["naive_f64"]["ssim"], (set_name, key)
        assert sha(nm) == ent["naive_f64"]["map_sha256"], (set_name, key)
        # the reference's own tolerances against its oracle (tests/rmgr-ssim-tests.cpp:98-104) hold for the reference ...
        assert ent["fma_vs_naive"]["global"] < 2e-6 and ent["fma_vs_naive"]["pixel"] < 1e-3


def test_exact_arithmetic_is_outside_the_fma_relative_pixel_bound_on_bbb1080(refsets):
    """A measured fact the mode contracts rest on: on the reference's bbb1080 set the FMA path itself is
    up to 6.46e-4 away from the exact per-pixel value -- more than north_star's 6.3e-4 'documented single-precision tolerance'
    (README: 6.22e-4 on stb-decoded pixels).  So NO arithmetic that is not correlated with the reference's rounding -- exact
    arithmetic included -- can be within 6.3e-4 of the FMA map on every pixel of that set; only forms that reproduce the
    reference's dominant roundings (MODE_EXACT: all; MODE_FAST: the three E[.] planes) can."""
    worst = max(e["fma_vs_naive"]["pixel"] for e in refsets["bbb1080"]["pairs"].values())
    over = sorted(k for k, e in refsets["bbb1080"]["pairs"].items() if e["fma_vs_naive"]["pixel"] > 6.3e-4)
    assert 6.4e-4 < worst < 6.5e-4 and over == ["q20_ch0", "q30_ch0", "q70_ch1"], (worst, over)
    assert max(e["fma_vs_naive"]["pixel"] for e in refsets["bbb360"]["pairs"].values()) < 6.3e-4
