"""GPU: parity of the HIP path (through the C ABI) with the oracle and the committed golden
vectors.  MODE_EXACT must reproduce the reference's FMA path bit for bit per pixel; the global
value is float(fp64 sum / N) where only the fp64 summation ORDER differs from the reference
(SURVEY.md A.3), so it is required bit-equal on the goldens and within 1 float ulp elsewhere.
north_star tolerance (global 1.5e-6, per-pixel 6.3e-4) is therefore met with margin zero-to-tiny.
"""
import ctypes
import errno
import hashlib
import os
import subprocess

import numpy as np
import pytest

import ssim_amd
from conftest import GOLDEN, f32_hex, image_entries, load_pair, ulp_diff

pytestmark = pytest.mark.gpu


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def assert_same_map(got, want, what):
    bad = np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad == 0, "%s: %d of %d pixels differ (max |d| %.3g)" % (what, bad, got.size, np.abs(got - want).max())


def test_extension_is_loaded_and_device_is_gfx950(gpu_ctx):
    d = gpu_ctx.describe()
    assert "gfx950" in d, d


@pytest.mark.parametrize("variant", [0, 1, 2])      # library default (picks by launch size), one column per lane, two columns per lane
def test_golden_fixtures_exact_mode(gpu_ctx, manifest, variant):
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    gpu_ctx.set_tuning(0, variant)
    for name in image_entries(manifest):
        ent = manifest[name]
        a, b = load_pair(ent)
        v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
        assert sha(m) == ent["fma"]["map_sha256"], name
        assert f32_hex(v) == ent["fma"]["ssim_hex"], (name, float(v), ent["fma"]["ssim"])
        v2, _ = gpu_ctx.ssim_planes(a, b, want_map=False)
        assert f32_hex(v2) == f32_hex(v), name
    gpu_ctx.set_tuning(0, 0)


def test_golden_fixtures_unfused_mode(gpu_ctx, manifest):
    """The reference's AVX/SSE/generic arithmetic (separately rounded multiply-add)."""
    gpu_ctx.set_mode(ssim_amd.MODE_UNFUSED)
    try:
        for name in image_entries(manifest):
            ent = manifest[name]
            a, b = load_pair(ent)
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert sha(m) == ent["avx"]["map_sha256"], name
            assert f32_hex(v) == ent["avx"]["ssim_hex"], name
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)


def test_reference_test_tolerances_vs_naive(gpu_ctx, manifest):
    """What tests/rmgr-ssim-tests.cpp asserts of an implementation: global within 2e-6 and every
    pixel within 1e-3 of the naive double oracle (:98-104, :310-326), on its einstein set."""
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    for name in image_entries(manifest):
        ent = manifest[name]
        if "map" not in ent["naive_f64"]:
            continue
        a, b = load_pair(ent)
        v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
        nmap = np.load(os.path.join(GOLDEN, ent["naive_f64"]["map"]))
        assert abs(float(v) - float(ent["naive_f64"]["ssim"])) < 2e-6, name
        assert np.abs(m.astype(np.float64) - nmap).max() < 1e-3, name


SIZES = [(1, 1), (1, 37), (37, 1), (3, 200), (200, 3), (10, 10), (11, 11), (63, 255), (64, 256), (65, 257),
         (129, 128), (127, 129), (130, 600), (300, 301)]


@pytest.mark.parametrize("variant,strip_rows", [(0, 0), (2, 0), (0, 7), (0, 64), (2, 5), (1, 0), (1, 33)])
def test_random_and_ragged_sizes_vs_oracle(gpu_ctx, oracle, variant, strip_rows):
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    gpu_ctx.set_tuning(strip_rows, variant)
    rng = np.random.default_rng(99)
    try:
        for (h, w) in SIZES:
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
            b = np.clip(a.astype(np.int32) + rng.integers(-25, 26, (h, w)), 0, 255).astype(np.uint8)
            if (h + w) % 3 == 0:
                b = rng.integers(0, 256, (h, w), dtype=np.uint8)
            ov, osum, om = oracle.ssim_f32(a, b, want_map=True, threads=4)
            v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
            assert_same_map(m, om, "%dx%d" % (w, h))
            assert ulp_diff(v, ov) <= 1, (w, h, float(v), float(ov))
    finally:
        gpu_ctx.set_tuning(0, 0)


def test_flat_black_white_images(gpu_ctx, oracle):
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    for va, vb in ((0, 0), (255, 255), (0, 255), (17, 18)):
        a = np.full((70, 300), va, np.uint8)
        b = np.full((70, 300), vb, np.uint8)
        ov, _, om = oracle.ssim_f32(a, b, want_map=True)
        v, m = gpu_ctx.ssim_planes(a, b, want_map=True)
        assert_same_map(m, om, (va, vb))
        assert f32_hex(v) == f32_hex(ov)


def device_params(ctx, keep, a_arr, a_off, a_step, a_stride, b_arr, b_off, b_step, b_stride, w, h, map_floats=0, map_off=0, map_step=1, map_stride=None):
    da, db = ctx.upload(a_arr), ctx.upload(b_arr)
    keep += [da, db]
    dm = None
    if map_floats:
        dm = ctx.alloc(4 * map_floats)
        dm.upload(np.full(map_floats, -7.0, np.float32))
        keep.append(dm)
    p = ssim_amd.make_params(w, h, da.ptr + a_off, a_step, a_stride, db.ptr + b_off, b_step, b_stride,
                             (dm.ptr + 4 * map_off) if dm else None, map_step, map_stride)
    return p, dm


def test_interleaved_bottom_up_and_column_major_addressing(gpu_ctx, manifest):
    """ImgParams semantics (reference ssim.h:481-499): pixel = topLeft + x*step + y*stride, any sign."""
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    il = manifest["_interleaved"]
    w, h, ch = il["width"], il["height"], il["channels"]
    A = np.fromfile(os.path.join(GOLDEN, il["a"]), np.uint8).reshape(h, w, ch)
    B = np.fromfile(os.path.join(GOLDEN, il["b"]), np.uint8).reshape(h, w, ch)
    keep = []
    try:
        for c, name in enumerate(il["per_channel"]):
            want = manifest[name]["fma"]
            # interleaved RGB, step 3 (init_interleaved, src/ssim.cpp:156-178)
            p, dm = device_params(gpu_ctx, keep, A, c, ch, w * ch, B, c, ch, w * ch, w, h, map_floats=w * h)
            v = gpu_ctx.compute_device(p)
            assert f32_hex(v) == want["ssim_hex"], name
            assert sha(dm.download(np.float32, (h, w))) == want["map_sha256"], name
            # bottom-up: flipped storage, negative stride, map written bottom-up too
            Af, Bf = np.ascontiguousarray(A[::-1]), np.ascontiguousarray(B[::-1])
            p, dm = device_params(gpu_ctx, keep, Af, (h - 1) * w * ch + c, ch, -w * ch, Bf, (h - 1) * w * ch + c, ch, -w * ch, w, h,
                                  map_floats=w * h, map_off=(h - 1) * w, map_step=1, map_stride=-w)
            v = gpu_ctx.compute_device(p)
            assert f32_hex(v) == want["ssim_hex"], name
            assert sha(dm.download(np.float32, (h, w))[::-1]) == want["map_sha256"], name
            # column-major storage of the same channel: step = h, stride = 1; map with step 2
            At, Bt = np.ascontiguousarray(A[:, :, c].T), np.ascontiguousarray(B[:, :, c].T)
            p, dm = device_params(gpu_ctx, keep, At, 0, h, 1, Bt, 0, h, 1, w, h, map_floats=2 * w * h, map_step=2, map_stride=2 * w)
            v = gpu_ctx.compute_device(p)
            assert f32_hex(v) == want["ssim_hex"], name
            raw = dm.download(np.float32, (h, 2 * w))
            assert sha(np.ascontiguousarray(raw[:, 0::2])) == want["map_sha256"], name
            assert np.all(raw[:, 1::2] == -7.0), "map elements between steps must stay untouched"
    finally:
        for d in keep:
            d.free()


@pytest.mark.parametrize("mode", [ssim_amd.MODE_EXACT, ssim_amd.MODE_FAST, ssim_amd.MODE_SEPARABLE])
def test_mirrored_storage_and_far_apart_pixels(gpu_ctx, oracle, mode):
    """Negative pixel steps (images and map stored right-to-left, several strips wide) and steps too large for
    the two-column kernel's 32-bit lane offsets (fits_strip2() false -> the 64-bit one-column kernel)."""
    gpu_ctx.set_mode(mode)
    rng = np.random.default_rng(1234)
    keep = []
    try:
        h, w = 37, 301
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, (h, w)), 0, 255).astype(np.uint8)
        want_v, want_m = gpu_ctx.ssim_planes(a, b, want_map=True)
        if mode == ssim_amd.MODE_EXACT:
            ov, _, om = oracle.ssim_f32(a, b, want_map=True)
            assert_same_map(want_m, om, "plain")
        am, bm = np.ascontiguousarray(a[:, ::-1]), np.ascontiguousarray(b[:, ::-1])
        # A mirrored (step -1), B plain, map mirrored (step -1) and bottom-up (stride -w)
        p, dm = device_params(gpu_ctx, keep, am, w - 1, -1, w, b, 0, 1, w, w, h,
                              map_floats=w * h, map_off=(h - 1) * w + (w - 1), map_step=-1, map_stride=-w)
        v = gpu_ctx.compute_device(p)
        assert f32_hex(v) == f32_hex(want_v)
        assert_same_map(np.ascontiguousarray(dm.download(np.float32, (h, w))[::-1, ::-1]), want_m, "mirrored")
        # both mirrored with a step of -2 inside double-width rows
        a2 = np.zeros((h, 2 * w), np.uint8); a2[:, 0::2] = am
        b2 = np.zeros((h, 2 * w), np.uint8); b2[:, 1::2] = bm
        p, dm = device_params(gpu_ctx, keep, a2, 2 * (w - 1), -2, 2 * w, b2, 2 * (w - 1) + 1, -2, 2 * w, w, h, map_floats=w * h, map_step=1, map_stride=w)
        v = gpu_ctx.compute_device(p)
        assert f32_hex(v) == f32_hex(want_v)
        assert_same_map(dm.download(np.float32, (h, w)), want_m, "step -2")

        # pixels 2 MiB apart, rows 7 bytes apart (overlapping rows are legal: the library only reads)
        h, w, step, stride = 9, 6, (1 << 21) + 3, 7
        n = (w - 1) * step + (h - 1) * stride + 1
        bufa = rng.integers(0, 256, n, dtype=np.uint8)
        bufb = rng.integers(0, 256, n, dtype=np.uint8)
        idx = (np.arange(h)[:, None] * stride + np.arange(w)[None, :] * step)
        a, b = bufa[idx], bufb[idx]
        mstep = 1 << 21
        p, dm = device_params(gpu_ctx, keep, bufa, 0, step, stride, bufb, 0, step, stride, w, h,
                              map_floats=(w - 1) * mstep + (h - 1) * 5 + 1, map_step=mstep, map_stride=5)
        v = gpu_ctx.compute_device(p)
        ref_v, ref_m = gpu_ctx.ssim_planes(a, b, want_map=True)
        assert f32_hex(v) == f32_hex(ref_v)
        got = dm.download(np.float32, ((w - 1) * mstep + (h - 1) * 5 + 1,))[(np.arange(h)[:, None] * 5 + np.arange(w)[None, :] * mstep)]
        assert_same_map(np.ascontiguousarray(got), ref_m, "far apart")
        if mode == ssim_amd.MODE_EXACT:
            ov, _, om = oracle.ssim_f32(a, b, want_map=True)
            assert_same_map(ref_m, om, "far apart vs oracle")
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        for d in keep:
            d.free()


def test_dropin_host_entry_points(gpu_ctx, manifest, lib):
    """rmgr_ssim_compute_ssim / _openmp on HOST pointers (the unchanged reference call), incl. map
    with ssimStep != 1, map-only, global-only, user allocator, and allocator failure -> ENOMEM."""
    ent = manifest["bbb257x65_q50_ch1"]
    a, b = load_pair(ent)
    h, w = a.shape
    want = np.load(os.path.join(GOLDEN, ent["fma"]["map"]))
    v, m = ssim_amd.compute_ssim(a, b, want_map=True)
    assert f32_hex(v) == ent["fma"]["ssim_hex"]
    assert_same_map(m, want, "host map")
    v, _ = ssim_amd.compute_ssim(a, b, want_map=False, openmp=True)
    assert f32_hex(v) == ent["fma"]["ssim_hex"]
    v, m = ssim_amd.compute_ssim(a, b, want_map=True, allocator=True)
    assert f32_hex(v) == ent["fma"]["ssim_hex"]
    assert_same_map(m, want, "host map, user allocator")

    # map with step 3 and a padded stride, global not requested (ssim == NULL is legal with a map)
    big = np.full((h, 3 * w + 5), -7.0, np.float32)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w, big.ctypes.data, 3, 3 * w + 5)
    assert lib.rmgr_ssim_compute_ssim(None, ctypes.byref(p), None) == 0
    assert_same_map(np.ascontiguousarray(big[:, 0:3 * w:3]), want, "strided host map")
    assert np.all(big[:, 1:3 * w:3] == -7.0) and np.all(big[:, 3 * w:] == -7.0)

    # bottom-up host map
    flip = np.zeros((h, w), np.float32)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w, flip.ctypes.data + 4 * (h - 1) * w, 1, -w)
    out = ctypes.c_float()
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None) == 0
    assert_same_map(np.ascontiguousarray(flip[::-1]), want, "bottom-up host map")
    assert f32_hex(out.value) == ent["fma"]["ssim_hex"]

    # allocator that fails -> ENOMEM (src/ssim.cpp:1050-1052)
    calls = []

    @ssim_amd.api.AllocFct
    def failing_alloc(size, alignment):
        calls.append((size, alignment))
        return None

    @ssim_amd.api.DeallocFct
    def dealloc(ptr):
        calls.append("free")
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w)
    p.alloc, p.dealloc = failing_alloc, dealloc
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None) == errno.ENOMEM
    assert len(calls) == 1 and calls[0][1] == 64

    # a thread pool's dispatch function is CALLED (round 6: one job per row band; tests/test_gpu_round6.py); a serial one here
    jobs = []

    @ssim_amd.api.ThreadPoolFct
    def dispatch(ctx, fct, args, threads, job_count):
        for j in range(job_count):
            jobs.append(j)
            fct(args[0], j)
        return 0
    tp = ssim_amd.ThreadPool(dispatch, None, 8)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), ctypes.byref(tp)) == 0
    assert f32_hex(out.value) == ent["fma"]["ssim_hex"] and len(jobs) >= 1

    # zero-sized image: not rejected by the reference, 0/0 -> NaN (SURVEY.md 8(b) "Errors")
    p = ssim_amd.make_params(0, 0, a.ctypes.data, 1, w, b.ctypes.data, 1, w)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), None) == 0
    assert np.isnan(out.value)


@pytest.mark.parametrize("static", [False, True])
def test_reference_style_cxx98_client_on_gpu(tmp_path, manifest, static):
    """A program written against the reference's API, linked against the shared library or the static archive."""
    from test_abi_cpu import build_dropin_client
    exe = build_dropin_client(tmp_path, static)
    il = manifest["_interleaved"]
    out = subprocess.run([exe, os.path.join(GOLDEN, il["a"]), os.path.join(GOLDEN, il["b"]), str(il["width"]), str(il["height"]), "3"],
                         check=True, capture_output=True, text=True).stdout.splitlines()
    assert out[0] == "version 2.1.0 2.1.0"
    for c, name in enumerate(il["per_channel"]):
        want = manifest[name]["fma"]["ssim_hex"]
        fields = out[1 + c].split()
        assert fields[:4] == ["channel", str(c), "ssim", want], out[1 + c]
        assert fields[4:8] == ["openmp_rc", "0", "openmp", want], out[1 + c]
        # round 6: the deprecated Params block's own thread pool is called (>= 1 job), gives the same float, and a pool that reports failure is ECHILD (src/ssim.cpp:1094-1097)
        assert fields[10:12] == ["pool", want] and fields[12] == "jobs" and int(fields[13]) >= 1 and fields[14:16] == ["failing_pool_errno", str(errno.ECHILD)], out[1 + c]
    # the same program on a full-size frame of the reference's bbb1080 set (interleaved RGB, 1920 x 1080, JPEG quality 50)
    from conftest import _decode_rgb
    import json
    ref = json.load(open(os.path.join(GOLDEN, "refsets.json")))["sets"]["bbb1080"]["pairs"]
    e = ref["q50_ch0"]
    pa, pb = str(tmp_path / "a.rgb.u8"), str(tmp_path / "b.rgb.u8")
    np.ascontiguousarray(_decode_rgb(e["a_file"])).tofile(pa)
    np.ascontiguousarray(_decode_rgb(e["b_file"])).tofile(pb)
    out = subprocess.run([exe, pa, pb, str(e["width"]), str(e["height"]), "3"], check=True, capture_output=True, text=True).stdout.splitlines()
    for c in range(3):
        want = ref["q50_ch%d" % c]["fma"]["ssim_hex"]
        fields = out[1 + c].split()
        assert fields[:4] == ["channel", str(c), "ssim", want] and fields[4:8] == ["openmp_rc", "0", "openmp", want], out[1 + c]


def test_batch_equals_single_calls(gpu_ctx, oracle):
    """enqueue_batch: one launch over N pairs; per-image fp64 sums identical to N single calls."""
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    w, h, n = 333, 210, 7
    keep, params = [], (ssim_amd.Params * n)()
    singles = []
    try:
        for i in range(n):
            a, b = oracle.synth_pair(w, h, 0x5EED + i)
            da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
            keep += [da, db]
            params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
            singles.append(gpu_ctx.compute_device(params[i]))
            ov, _, _ = oracle.ssim_f32(a, b, threads=4)
            assert ulp_diff(singles[-1], ov) <= 1
        sums = gpu_ctx.alloc(8 * n)
        keep.append(sums)
        gpu_ctx.enqueue_batch(params, n, sums.ptr)
        gpu_ctx.enqueue_batch(params, n, sums.ptr)      # re-enqueue reuses the uploaded descriptor table
        gpu_ctx.synchronize()
        got = ssim_amd.finalize(sums.download(np.float64, (n,)), w, h)
        assert [f32_hex(x) for x in got] == [f32_hex(x) for x in singles]
        # a sub-batch gives the same per-image sums (what sharding over GPUs relies on)
        sub = (ssim_amd.Params * 3)(params[4], params[1], params[6])
        gpu_ctx.enqueue_batch(sub, 3, sums.ptr)
        gpu_ctx.synchronize()
        got3 = ssim_amd.finalize(sums.download(np.float64, (3,)), w, h)
        assert [f32_hex(x) for x in got3] == [f32_hex(singles[i]) for i in (4, 1, 6)]
        # mismatched sizes are refused
        bad = (ssim_amd.Params * 2)(params[0], params[1])
        bad[1].width = w - 1
        with pytest.raises(ssim_amd.SsimError) as ei:
            gpu_ctx.enqueue_batch(bad, 2, sums.ptr)
        assert ei.value.errno == errno.EINVAL
    finally:
        for d in keep:
            d.free()


def test_batches_beyond_one_launch_and_sibling_channel_order(gpu_ctx, oracle):
    """70 000 tiny pairs: more than grid.z allows, so the ABI issues two launches (65 535 + 4 465), with strip
    totals that are not multiples of 8 (general XCD renumbering).  Then interleaved RGB pairs whose channel
    descriptors sit next to each other (interleaved_group(): sibling-minor strip order).  Every sum is checked."""
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    rng = np.random.default_rng(31)
    w, h, pool, n = 13, 9, 37, 70000
    keep = []
    try:
        want = []
        bufs = []
        for k in range(pool):
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
            b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, (h, w)), 0, 255).astype(np.uint8)
            da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
            keep += [da, db]
            bufs.append((da.ptr, db.ptr))
            want.append(oracle.ssim_f32(a, b)[0])
        params = (ssim_amd.Params * n)()
        for i in range(n):
            pa, pb = bufs[(i * 7 + i // pool) % pool]
            params[i] = ssim_amd.make_params(w, h, pa, 1, w, pb, 1, w)
        sums = gpu_ctx.alloc(8 * n)
        keep.append(sums)
        gpu_ctx.enqueue_batch(params, n, sums.ptr)
        gpu_ctx.synchronize()
        got = ssim_amd.finalize(sums.download(np.float64, (n,)), w, h)
        idx = np.array([(i * 7 + i // pool) % pool for i in range(n)])
        exp = np.array(want, np.float32)[idx]
        assert np.all(np.abs(got.view(np.int32).astype(np.int64) - exp.view(np.int32).astype(np.int64)) <= 1)

        # interleaved RGBA pairs (4 sibling channels) and RGB pairs (3), several pairs per launch
        for ch, pairs, (hh, ww) in ((4, 3, (70, 301)), (3, 5, (33, 129)), (2, 2, (64, 256))):
            params = (ssim_amd.Params * (ch * pairs))()
            exp = []
            for i in range(pairs):
                a = rng.integers(0, 256, (hh, ww, ch), dtype=np.uint8)
                b = np.clip(a.astype(np.int32) + rng.integers(-40, 41, a.shape), 0, 255).astype(np.uint8)
                da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
                keep += [da, db]
                for c in range(ch):
                    params[ch * i + c] = ssim_amd.make_params(ww, hh, da.ptr + c, ch, ww * ch, db.ptr + c, ch, ww * ch)
                    exp.append(oracle.ssim_f32(np.ascontiguousarray(a[:, :, c]), np.ascontiguousarray(b[:, :, c]))[0])
            s2 = gpu_ctx.alloc(8 * ch * pairs)
            keep.append(s2)
            gpu_ctx.enqueue_batch(params, ch * pairs, s2.ptr)
            gpu_ctx.synchronize()
            got = ssim_amd.finalize(s2.download(np.float64, (ch * pairs,)), ww, hh)
            assert all(ulp_diff(g, e) <= 1 for g, e in zip(got, exp)), (ch, pairs)
    finally:
        for d in keep:
            d.free()


def test_full_size_known_answers_and_properties(gpu_ctx, oracle, manifest):
    """BASELINE.json configs at full size: 4096^2 global (C2) and 8192^2 with map (C3), checked by
    the reference's known answers, the oracle, and size-independent properties."""
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    keep = []
    try:
        # --- C2: 4096 x 4096, global only ---
        ent = manifest["_synthetic"]["4096x4096_5eed"]
        a, b = oracle.synth_pair(4096, 4096, 0x5EED)
        assert int(a.sum(dtype=np.int64)) == ent["sumA"] and int(b.sum(dtype=np.int64)) == ent["sumB"]
        da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
        keep += [da, db]
        p = ssim_amd.make_params(4096, 4096, da.ptr, 1, 4096, db.ptr, 1, 4096)
        v = gpu_ctx.compute_device(p)
        assert f32_hex(v) == ent["fma"]["ssim_hex"], float(v)
        # identity: SSIM(a, a) == 1 exactly
        p_id = ssim_amd.make_params(4096, 4096, da.ptr, 1, 4096, da.ptr, 1, 4096)
        assert f32_hex(gpu_ctx.compute_device(p_id)) == f32_hex(1.0)
        # symmetry: every operation of the path is symmetric in (a, b)
        p_sw = ssim_amd.make_params(4096, 4096, db.ptr, 1, 4096, da.ptr, 1, 4096)
        assert f32_hex(gpu_ctx.compute_device(p_sw)) == f32_hex(v)
        # strip height and kernel variant do not change the per-pixel values
        dm = gpu_ctx.alloc(4 * 4096 * 4096)
        keep.append(dm)
        pm = ssim_amd.make_params(4096, 4096, da.ptr, 1, 4096, db.ptr, 1, 4096, dm.ptr, 1, 4096)
        gpu_ctx.compute_device(pm)
        base = dm.download(np.float32, (4096, 4096))
        for rows, variant in ((32, 0), (100, 0), (77, 0), (0, 1), (0, 2), (0, 3)):
            gpu_ctx.set_tuning(rows, variant)
            assert f32_hex(gpu_ctx.compute_device(pm)) == f32_hex(v)
            assert_same_map(dm.download(np.float32, (4096, 4096)), base, (rows, variant))
        gpu_ctx.set_tuning(0, 0)
        # oracle on the full image
        ov, _, om = oracle.ssim_f32(a, b, want_map=True, threads=oracle.oracle_lib().oracle_max_threads())
        assert f32_hex(ov) == ent["fma"]["ssim_hex"]
        assert_same_map(base, om, "4096^2 map vs oracle")
        # mirror: horizontally flipped inputs give the horizontally flipped map (fold is commutative)
        da2, db2 = gpu_ctx.upload(a[:, ::-1]), gpu_ctx.upload(b[:, ::-1])
        keep += [da2, db2]
        pf = ssim_amd.make_params(4096, 4096, da2.ptr, 1, 4096, db2.ptr, 1, 4096, dm.ptr, 1, 4096)
        gpu_ctx.compute_device(pf)
        assert_same_map(dm.download(np.float32, (4096, 4096))[:, ::-1], base, "mirror")
        for d in keep:
            d.free()
        keep = []
        del base, om

        # --- C3: 8192 x 8192 with map ---
        ent = manifest["_synthetic"]["8192x8192_5eed"]
        a, b = oracle.synth_pair(8192, 8192, 0x5EED)
        assert int(a.sum(dtype=np.int64)) == ent["sumA"]
        da, db, dm = gpu_ctx.upload(a), gpu_ctx.upload(b), gpu_ctx.alloc(4 * 8192 * 8192)
        keep += [da, db, dm]
        p = ssim_amd.make_params(8192, 8192, da.ptr, 1, 8192, db.ptr, 1, 8192, dm.ptr, 1, 8192)
        v = gpu_ctx.compute_device(p)
        assert f32_hex(v) == ent["fma"]["ssim_hex"], float(v)
        m = dm.download(np.float32, (8192, 8192))
        # global value is the mean of the map it wrote
        assert abs(float(m.astype(np.float64).mean()) - float(v)) < 1e-7
        # EVERY pixel of the 67 Mpixel map against the oracle (all host threads; the reference's tests assert every pixel,
        # tests/rmgr-ssim-tests.cpp:315-326), in slabs so that no second 256 MB temporary is needed
        ov, _, om = oracle.ssim_f32(a, b, want_map=True, threads=oracle.oracle_lib().oracle_max_threads())
        assert f32_hex(ov) == ent["fma"]["ssim_hex"]
        for y in range(0, 8192, 1024):
            assert_same_map(m[y:y + 1024], om[y:y + 1024], "8192^2 map vs oracle, rows %d.." % y)
    finally:
        for d in keep:
            d.free()


@pytest.mark.parametrize("openmp", [False, True])
@pytest.mark.parametrize("heap", [False, True])
@pytest.mark.parametrize("want_map", [False, True])
def test_reference_test_matrix_on_host_entry_points(manifest, refsets, openmp, heap, want_map):
    """The reference's own test matrix (tests/rmgr-ssim-tests.cpp:468-507): {stack, heap} x {map, nomap} x
    {serial, openmp} over EVERY image set of the reference's tests (:520-524: einstein, bbb360, bbb1080, bbb255, bbb257 --
    11 JPEG qualities x 3 channels each), here through the unchanged host-pointer entry points.  Where the
    reference allows 2e-6 / 1e-3 against its naive oracle, the GPU path must match the FMA reference exactly."""
    from conftest import refset_pair
    for name in image_entries(manifest):
        ent = manifest[name]
        a, b = load_pair(ent)
        v, m = ssim_amd.compute_ssim(a, b, want_map=want_map, openmp=openmp, allocator=heap)
        assert f32_hex(v) == ent["fma"]["ssim_hex"], name
        if want_map:
            assert sha(m) == ent["fma"]["map_sha256"], name
    n = 0
    for set_name in ("bbb255", "bbb257", "bbb360", "bbb1080"):
        for key in sorted(refsets[set_name]["pairs"]):
            ent = refsets[set_name]["pairs"][key]
            a, b = refset_pair(ent)
            v, m = ssim_amd.compute_ssim(a, b, want_map=want_map, openmp=openmp, allocator=heap)
            assert f32_hex(v) == ent["fma"]["ssim_hex"], (set_name, key)
            if want_map:
                assert sha(m) == ent["fma"]["map_sha256"], (set_name, key)
            n += 1
    assert n == 132


def test_dropin_call_is_thread_safe(manifest):
    """The reference's compute_ssim is re-entrant and parallel across callers (SURVEY.md 8(b) "Threading"; src/ssim.cpp:933-1106 has no
    global state).  The drop-in call leases one of the process-wide default contexts per call (round 5; rounds 1-4: one context under a
    lock).  Hammer it from several host threads with different inputs, with and without the map: every result bit-exact, and the pool
    must have grown beyond one context but not beyond its limit -- six threads, four leases."""
    import threading
    names = [n for n in image_entries(manifest)][:8]
    pairs = [(load_pair(manifest[n]), manifest[n]["fma"]["ssim_hex"], manifest[n]["fma"]["map_sha256"]) for n in names]
    errors = []
    gate = threading.Barrier(6)

    def work(tid):
        try:
            gate.wait()
            for it in range(12):
                (a, b), want, want_map = pairs[(tid + it) % len(pairs)]
                v, m = ssim_amd.compute_ssim(a, b, want_map=(it % 2 == 0))
                if f32_hex(v) != want or (m is not None and sha(m) != want_map):
                    errors.append((tid, it))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    contexts, limit = ssim_amd.default_pool()
    assert limit == int(os.environ.get("RMGR_SSIM_HIP_POOL", "4")) and 1 <= contexts <= limit
    assert contexts >= 2 or limit == 1, "six concurrent callers were served by a single default context"


def test_default_pool_follows_the_mode_of_the_drop_in_calls(manifest):
    """rmgr_ssim_hip_set_mode(NULL, ...) -- what rmgr::ssim::select_impl drives -- is a property of the pool: every default context, the
    ones that exist and the ones concurrent callers create later, computes in that mode."""
    import threading
    lib = ssim_amd.load_library()
    ent = manifest["einstein_blur"]
    a, b = load_pair(ent)
    assert lib.rmgr_ssim_hip_set_mode(None, ssim_amd.MODE_UNFUSED) == 0
    try:
        got = []
        gate = threading.Barrier(5)

        def work():
            gate.wait()
            got.append(f32_hex(ssim_amd.compute_ssim(a, b)[0]))
        threads = [threading.Thread(target=work) for _ in range(5)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert got == [ent["avx"]["ssim_hex"]] * 5, got
    finally:
        assert lib.rmgr_ssim_hip_set_mode(None, ssim_amd.MODE_EXACT) == 0
    assert f32_hex(ssim_amd.compute_ssim(a, b)[0]) == ent["fma"]["ssim_hex"]


def test_two_contexts_and_mode_switching(gpu_ctx, manifest):
    """Independent contexts (own stream, own scratch) coexist; switching modes on one does not leak into the other."""
    ent = manifest["einstein_blur"]
    a, b = load_pair(ent)
    other = ssim_amd.Context(0, mode=ssim_amd.MODE_UNFUSED)
    try:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        for _ in range(3):
            v0, _ = gpu_ctx.ssim_planes(a, b)
            v1, _ = other.ssim_planes(a, b)
            assert f32_hex(v0) == ent["fma"]["ssim_hex"] and f32_hex(v1) == ent["avx"]["ssim_hex"]
        other.set_mode(ssim_amd.MODE_EXACT)
        assert f32_hex(other.ssim_planes(a, b)[0]) == ent["fma"]["ssim_hex"]
    finally:
        other.close()


def test_pipelined_host_batch_equals_single_host_calls(oracle):
    """rmgr_ssim_hip_compute_ssim_batch_host: chunked, double-buffered staging on a second stream.  Results must be
    the single-pair host call's, bit for bit, for small pairs (pinned gather path), large pairs (direct copies,
    several chunks), odd layouts, and a batch that spans many chunks."""
    rng = np.random.default_rng(77)

    def pairs_of(h, w, n, flip=False):
        out = []
        for i in range(n):
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
            b = np.clip(a.astype(np.int32) + rng.integers(-35, 36, (h, w)), 0, 255).astype(np.uint8)
            if flip and i % 2:
                a, b = a[::-1, ::-1], b[:, ::-1]          # negative strides / steps as views
            out.append((a, b))
        return out

    for (h, w, n, flip) in [(64, 96, 300, False), (33, 17, 50, True), (480, 640, 40, False), (1080, 1920, 30, False), (2100, 4099, 5, True)]:
        ps = pairs_of(h, w, n, flip)
        got = ssim_amd.compute_ssim_batch(ps)
        for i in range(0, n, max(1, n // 7)):
            a, b = ps[i]
            single, _ = ssim_amd.compute_ssim(np.ascontiguousarray(a), np.ascontiguousarray(b))
            assert f32_hex(got[i]) == f32_hex(single), (h, w, i)
        ov = oracle.ssim_f32(np.ascontiguousarray(ps[-1][0]), np.ascontiguousarray(ps[-1][1]), threads=4)[0]
        assert ulp_diff(got[-1], ov) <= 1
    # argument checks: maps are not supported, sizes must agree
    a = np.zeros((8, 8), np.uint8)
    m = np.zeros((8, 8), np.float32)
    lib = ssim_amd.load_library()
    p = (ssim_amd.Params * 1)(ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8, m.ctypes.data, 1, 8))
    out = (ctypes.c_float * 1)()
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 1, p, out) == errno.EINVAL
    p2 = (ssim_amd.Params * 2)(ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8), ssim_amd.make_params(8, 4, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8))
    out2 = (ctypes.c_float * 2)()
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 2, p2, out2) == errno.EINVAL
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host(None, 0, None, None) == 0


def test_one_process_several_devices_batch(oracle):
    """rmgr_ssim_hip_compute_ssim_batch_host_devices ('one process x 8 devices'): the batch sharded by
    image over a device list, one worker thread + context per entry.  Every list -- all visible devices, one device, the
    same device three times (three contexts on one GPU: what a 1-GPU box can exercise), more entries than pairs -- must
    return the single-device floats bit for bit, in every arithmetic mode; uneven splits included."""
    rng = np.random.default_rng(4242)
    ndev = ssim_amd.device_count()
    for (h, w, n) in [(96, 130, 37), (720, 1280, 11), (8, 8, 2)]:
        ps = []
        for i in range(n):
            a = rng.integers(0, 256, (h, w), dtype=np.uint8)
            ps.append((a, np.clip(a.astype(np.int32) + rng.integers(-35, 36, (h, w)), 0, 255).astype(np.uint8)))
        want = ssim_amd.compute_ssim_batch(ps)                       # default context, MODE_EXACT
        ov = oracle.ssim_f32(ps[0][0], ps[0][1])[0]
        assert f32_hex(want[0]) == f32_hex(ov)
        for devices in (None, [0], [0, 0, 0], [ndev - 1] * 5, list(range(ndev)) * 2):
            got = ssim_amd.compute_ssim_batch_devices(ps, devices)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (h, w, n, devices)
    ps = ps[:2] + [(np.ascontiguousarray(ps[0][0][::-1]), ps[0][1])]
    for mode in (ssim_amd.MODE_UNFUSED, ssim_amd.MODE_FAST, ssim_amd.MODE_SEPARABLE, ssim_amd.MODE_DOUBLE):
        with ssim_amd.Context(0, mode=mode) as ctx:
            want = ssim_amd.compute_ssim_batch(ps, ctx)
        got = ssim_amd.compute_ssim_batch_devices(ps, [0, 0], mode)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), mode
    # argument checks
    lib = ssim_amd.load_library()
    a = np.zeros((8, 8), np.uint8)
    p = (ssim_amd.Params * 1)(ssim_amd.make_params(8, 8, a.ctypes.data, 1, 8, a.ctypes.data, 1, 8))
    out = (ctypes.c_float * 1)()
    bad = (ctypes.c_int32 * 1)(ndev)
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(bad, 1, 0, 1, p, out) == errno.EINVAL          # not a visible device
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(bad, 0, 0, 1, p, out) == errno.EINVAL          # a list of length 0
    assert lib.rmgr_ssim_hip_compute_ssim_batch_host_devices(None, 0, 0, 0, None, None) == 0


def test_context_on_another_device_than_the_current_one():
    """an entry point must keep the CONTEXT's device current for its whole duration -- the multi-channel and
    luminance host calls allocate, create events and launch after their staging helper returns.  Never skipped: on a box with two or
    more GPUs (the driver's multi-GPU boxes) the default contexts are put on EVERY device in turn ($RMGR_SSIM_HIP_DEVICE, a process each)
    while device 0 stays the current one of the calling thread, and every device must return device 0's bits; on a one-GPU box the same
    child-process path runs for device 0 alone (the guard's scope is then covered by construction only)."""
    ndev = ssim_amd.device_count()
    assert ndev >= 1
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    want, want_map = ssim_amd.compute_ssim_channels(a, b, want_map=True)          # default context: device 0
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import ssim_amd; rng = np.random.default_rng(3); "
            "a = rng.integers(0, 256, (120, 200, 3), dtype=np.uint8); b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8); "
            "v, m = ssim_amd.compute_ssim_channels(a, b, want_map=True); y, _ = ssim_amd.compute_ssim_luminance(a, b); "
            "print(' '.join('%%08x' %% int(x) for x in v.view(np.uint32)), '%%08x' %% int(np.float32(y).view(np.uint32)))"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for dev in range(ndev):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, RMGR_SSIM_HIP_DEVICE=str(dev)))
        assert r.returncode == 0, (dev, r.stderr[-800:])
        outs.append(r.stdout.split())
    assert all(o == outs[0] for o in outs) and outs[0][:3] == ["%08x" % int(x) for x in want.view(np.uint32)], outs
    if ndev >= 2:
        # a caller-owned context on the LAST device while device 0 is current: device-pointer path, host path, a batch over every physical device
        rng = np.random.default_rng(5)
        pa = rng.integers(0, 256, (300, 500), dtype=np.uint8)
        pb = np.clip(pa.astype(np.int32) + rng.integers(-20, 21, pa.shape), 0, 255).astype(np.uint8)
        with ssim_amd.Context(0) as c0, ssim_amd.Context(ndev - 1) as c1:
            v0, m0 = c0.ssim_planes(pa, pb, want_map=True)
            v1, m1 = c1.ssim_planes(pa, pb, want_map=True)
            assert int(v0.view(np.uint32)) == int(v1.view(np.uint32)) and np.array_equal(m0.view(np.uint32), m1.view(np.uint32))
        ps = [(pa, pb)] * (2 * ndev + 1)
        got = ssim_amd.compute_ssim_batch_devices(ps, list(range(ndev)))
        assert np.all(got.view(np.uint32) == int(v0.view(np.uint32)))
