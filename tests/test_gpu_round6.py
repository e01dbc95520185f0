"""Round 6 (GPU): the caller's thread pool is CALLED (one job per row band, ECHILD when it fails), the default contexts' memory policy, the
in-library VALU probe, the measured plan choice (rmgr_ssim_hip_tune)."""
import ctypes
import errno
import json
import os
import random
import subprocess
import sys
import threading

import numpy as np
import pytest

import ssim_amd
from conftest import GOLDEN, ROOT, f32_hex, load_pair

pytestmark = pytest.mark.gpu


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def make_pool(order="forward", threads=1, fail=False, skip=0, repeat=False):
    """A rmgr_ssim_ThreadPool whose dispatch function is Python: runs the jobs in `order` on `threads` Python threads (ctypes releases the GIL
    inside the library: the jobs really overlap), returns non-zero when `fail`, leaves out the last `skip` jobs, runs job 0 twice when `repeat`."""
    seen = {"calls": 0, "jobs": [], "threads": None, "job_count": None}

    @ssim_amd.api.ThreadPoolFct
    def dispatch(context, fct, args, thread_count, job_count):
        seen["calls"] += 1
        seen["threads"], seen["job_count"] = thread_count, job_count
        jobs = list(range(job_count))
        if order == "reverse":
            jobs.reverse()
        elif order == "shuffled":
            random.Random(7).shuffle(jobs)
        jobs = jobs[:len(jobs) - skip] if skip else jobs
        if repeat and jobs:
            jobs.append(jobs[0])
        lock = threading.Lock()

        def worker(t):
            while True:
                with lock:
                    if not jobs:
                        return
                    j = jobs.pop(0)
                    seen["jobs"].append(j)
                fct(args[t], j)                     # args[t]: used by one thread at a time
        n = max(1, min(threads, thread_count))
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return 1 if fail else 0
    return dispatch, seen


def call(lib, a, b, tp, want_map=False, want_global=True, map_step=1, map_stride=None, out_map=None):
    h, w = a.shape
    m = None
    if want_map:
        m = out_map if out_map is not None else np.full((h, (map_stride or w * map_step)), -7.0, np.float32)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, a.strides[0], b.ctypes.data, 1, b.strides[0], m.ctypes.data if want_map else None, map_step, map_stride or w * map_step)
    out = ctypes.c_float(-1.0)
    rc = lib.rmgr_ssim_compute_ssim(ctypes.byref(out) if want_global else None, ctypes.byref(p), ctypes.byref(tp) if tp is not None else None)
    return rc, np.float32(out.value), m


def test_thread_pool_runs_the_row_band_jobs(manifest):
    """include/rmgr/ssim.h:448-466, src/ssim.cpp:1048-1100: dispatch is called once, with min(threadCount, 64) threads; it must call the job function
    exactly jobCount times; the result is the reference's -- and the no-pool call's, bit for bit -- for any job order and any number of threads."""
    from ssim_amd import synth
    lib = ssim_amd.load_library()
    ent = manifest["bbb257x65_q50_ch1"]
    a, b = load_pair(ent)
    dispatch, seen = make_pool()
    tp = ssim_amd.ThreadPool(dispatch, None, 8)
    rc, v, _ = call(lib, a, b, tp)
    assert rc == 0 and f32_hex(v) == ent["fma"]["ssim_hex"]
    assert seen["calls"] == 1 and seen["job_count"] >= 1 and sorted(seen["jobs"]) == list(range(seen["job_count"])) and seen["threads"] == 8
    # the reference caps the thread count at 64 (src/ssim.cpp:1025)
    dispatch, seen = make_pool()
    rc, v, _ = call(lib, a, b, ssim_amd.ThreadPool(dispatch, None, 1000))
    assert rc == 0 and seen["threads"] == 64 and f32_hex(v) == ent["fma"]["ssim_hex"]
    # a large pair with the map: several jobs; serial, reversed on 4 threads, shuffled on 3 -- value and every map pixel equal the no-pool call's
    w = h = 4096
    A, B = synth.pair_numpy(w, h, synth.BASE_SEED)
    rc, v0, m0 = call(lib, A, B, None, want_map=True)
    assert rc == 0 and int(v0.view(np.uint32)) == 0x3f64b7be
    for order, threads in (("forward", 1), ("reverse", 4), ("shuffled", 3)):
        dispatch, seen = make_pool(order, threads)
        rc, v, m = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, threads), want_map=True)
        assert rc == 0 and seen["job_count"] >= 4 and sorted(seen["jobs"]) == list(range(seen["job_count"])), (order, seen)
        assert int(v.view(np.uint32)) == 0x3f64b7be and np.array_equal(bits(m), bits(m0)), order
    # global only (no map): fewer bands, same bits
    dispatch, seen = make_pool("reverse", 2)
    rc, v, _ = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, 2))
    assert rc == 0 and int(v.view(np.uint32)) == 0x3f64b7be and 1 <= seen["job_count"] <= 4
    # a map with ssimStep 2 and a padded stride goes back through the context's bounce buffers, job by job
    dispatch, seen = make_pool("shuffled", 4)
    rc, v, big = call(lib, A[:1500, :1200], B[:1500, :1200], ssim_amd.ThreadPool(dispatch, None, 4), want_map=True, map_step=2, map_stride=2 * 1200 + 3)
    rc0, v00, m00 = call(lib, np.ascontiguousarray(A[:1500, :1200]), np.ascontiguousarray(B[:1500, :1200]), None, want_map=True)
    assert rc == 0 and rc0 == 0 and int(v.view(np.uint32)) == int(v00.view(np.uint32))
    assert np.array_equal(bits(big[:, 0:2400:2]), bits(m00)) and np.all(big[:, 1:2400:2] == -7.0) and np.all(big[:, 2400:] == -7.0)
    # the map alone (ssim == NULL) through the pool
    dispatch, seen = make_pool("forward", 2)
    rc, _, m = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, 2), want_map=True, want_global=False)
    assert rc == 0 and np.array_equal(bits(m), bits(m0))
    # an empty image has no jobs: dispatch is still called, with jobCount 0 (the reference dispatches its zero tiles), and 0 / 0 is NaN
    dispatch, seen = make_pool()
    p = ssim_amd.make_params(0, 0, a.ctypes.data, 1, 1, b.ctypes.data, 1, 1)
    out = ctypes.c_float()
    tp = ssim_amd.ThreadPool(dispatch, None, 2)
    assert lib.rmgr_ssim_compute_ssim(ctypes.byref(out), ctypes.byref(p), ctypes.byref(tp)) == 0 and np.isnan(out.value)
    assert seen["calls"] == 1 and seen["job_count"] == 0


def test_thread_pool_failure_is_echild(manifest):
    """src/ssim.cpp:1091-1097: a dispatch function that returns non-zero makes the call fail with ECHILD -- when a global value was asked for; with
    ssim == NULL its result is not looked at.  A pool that reports success without having run every job once is a failed worker too."""
    from ssim_amd import synth
    lib = ssim_amd.load_library()
    A, B = synth.pair_numpy(2048, 2048, synth.BASE_SEED)
    dispatch, seen = make_pool(fail=True)
    tp = ssim_amd.ThreadPool(dispatch, None, 4)
    rc, v, _ = call(lib, A, B, tp)
    assert rc == errno.ECHILD and seen["calls"] == 1 and len(seen["jobs"]) == seen["job_count"] >= 1
    rc, _, m = call(lib, A, B, tp, want_map=True, want_global=False)           # ssim == NULL: the pool's result is ignored
    rc0, _, m0 = call(lib, A, B, None, want_map=True, want_global=False)
    assert rc == 0 and rc0 == 0 and np.array_equal(bits(m), bits(m0))
    for kw in ({"skip": 1}, {"repeat": True}):
        dispatch, seen = make_pool(**kw)
        rc, _, _ = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, 4), want_map=True)
        assert rc == errno.ECHILD, kw
    # validation is the reference's: a dispatch function with threadCount 0 is EINVAL before anything runs (src/ssim.cpp:976-978)
    dispatch, seen = make_pool()
    rc, _, _ = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, 0))
    assert rc == errno.EINVAL and seen["calls"] == 0
    # ... and the library is in working order afterwards
    rc, v, _ = call(lib, A, B, None)
    dispatch, seen = make_pool("reverse", 3)
    rc2, v2, _ = call(lib, A, B, ssim_amd.ThreadPool(dispatch, None, 3))
    assert rc == 0 and rc2 == 0 and int(v.view(np.uint32)) == int(v2.view(np.uint32))


@pytest.mark.parametrize("expect,env", [("released", {}), ("released", {"RMGR_SSIM_HIP_POOL_RETAIN_MB": "0"}), ("retained", {"RMGR_SSIM_HIP_POOL_RETAIN_MB": "-1"})],
                         ids=["default-cap-256MB", "cap-0", "no-cap-then-trim"])
def test_default_pool_memory_policy(expect, env):
    """The reference retains nothing past the call (src/ssim.cpp:1048-1088).  Four concurrent 8192^2 + map calls on the default contexts: with the
    default cap (or 0) every context gives its 402 MB of staging back when its call ends; without a cap it stays until
    rmgr_ssim_hip_trim_default_pool(); either way hipMemGetInfo returns to within 64 MB of the pre-call figure and the results keep their bits."""
    e = dict(os.environ)
    e.pop("RMGR_SSIM_HIP_POOL_RETAIN_MB", None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "pool_memory_selftest.py"), expect], env=e, capture_output=True, text=True, timeout=600)
    line = json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip() else {}
    assert r.returncode == 0 and line.get("ok"), (line, r.stderr[-2000:])


def test_trim_of_a_callers_context(gpu_ctx):
    """rmgr_ssim_hip_trim(ctx): the staging goes, the context keeps working and gives the same bits."""
    from ssim_amd import synth
    a, b = synth.pair_numpy(2048, 2048, synth.BASE_SEED)
    h, w = a.shape
    m1 = np.zeros((h, w), np.float32)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w, m1.ctypes.data, 1, w)
    v1 = gpu_ctx.compute_host(p)
    free1, _ = ssim_amd.memory_info(gpu_ctx)
    gpu_ctx.trim()
    free2, _ = ssim_amd.memory_info(gpu_ctx)
    assert free2 - free1 >= 2 * w * h + 4 * w * h - (1 << 20), (free1, free2)      # the staged images and the dense map came back
    m2 = np.zeros((h, w), np.float32)
    p = ssim_amd.make_params(w, h, a.ctypes.data, 1, w, b.ctypes.data, 1, w, m2.ctypes.data, 1, w)
    v2 = gpu_ctx.compute_host(p)
    assert int(v1.view(np.uint32)) == int(v2.view(np.uint32)) and np.array_equal(bits(m1), bits(m2))


def test_valu_probe(gpu_ctx):
    """rmgr_ssim_hip_probe_valu: a packed-fp32 stream at a forced occupancy.  On an MI355X: 60...70 T lane-ops/s at two waves per SIMD, more at
    eight, never above the 78.6 T data-sheet peak; the dependent-chain stream is no faster than the independent one.  A call is the best of three bursts because
    a burst can run degraded (the rate of one wave fewer per SIMD at an unchanged clock: profiles/r06_probe_bimodal.txt; ~3 % of two-wave CALLS still do), so the
    box's peak at an occupancy is taken as bench.py takes it: the best of a few calls; those agree within 3 %."""
    peak = lambda waves, kind=0, calls=3: max(gpu_ctx.probe_valu(waves, kind) for _ in range(calls))
    r2, r3, r8 = peak(2), peak(3), peak(8)
    assert 55.0 < r2 < r3 * 1.02 and r3 < r8 * 1.02 and r8 < 79.0, (r2, r3, r8)
    assert peak(2, 1) < r2 * 1.03
    assert abs(peak(2) - r2) / r2 < 0.03
    lib = ssim_amd.load_library()
    t = ctypes.c_double()
    for waves, kind, n in ((5, 0, 5), (0, 0, 5), (2, 2, 5), (2, 0, 0), (2, 0, 65)):
        assert lib.rmgr_ssim_hip_probe_valu(gpu_ctx.handle, waves, kind, n, ctypes.byref(t), None, None) == errno.EINVAL
    assert lib.rmgr_ssim_hip_probe_valu(None, 2, 0, 5, ctypes.byref(t), None, None) == errno.EINVAL
    # the shader clock the probe's timed launches ran at (workgroup 0: s_memtime cycles per s_memrealtime tick): an MI355X under load holds 1.9 ... 2.4 GHz, and at that
    # clock a SIMD retires a packed instruction every 4.0 ... 5.0 cycles at two waves
    rate, mhz, slowest = max(gpu_ctx.probe_valu(2, 0, 5, with_clock=True) for _ in range(3))
    assert 1800.0 < slowest <= mhz < 2450.0, (mhz, slowest)
    clk_per_instr = 32768.0 * mhz * 1e6 / (rate * 1e12) * 4.0
    assert 3.99 < clk_per_instr < 5.2, (rate, mhz, clk_per_instr)
    # the SSIM path is untouched by it
    from ssim_amd import synth
    a, b = synth.pair_numpy(256, 256, synth.BASE_SEED)
    v, _ = gpu_ctx.ssim_planes(a, b)
    assert abs(float(v) - 0.892903090) < 6e-8      # the 256 x 256 known answer of the FMA path


def test_profiled_launches_report_their_shader_clock(gpu_ctx):
    """rmgr_ssim_hip_get_profile_clock: while profiling is on, the first workgroups of every strip-kernel launch (one per XCD) add their shader cycles and reference ticks
    to per-XCD device counters -- in every kernel form (strips, balanced chunks, one column per lane, fp64) -- and the sums keep their bits with the instrumentation on."""
    from ssim_amd import synth
    w, h, n = 1920, 1080, 24
    keep, params = [], (ssim_amd.Params * n)()
    try:
        for i in range(n):
            da, db = gpu_ctx.alloc(w * h), gpu_ctx.alloc(w * h)
            gpu_ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
            keep += [da, db]
            params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
        sums = gpu_ctx.alloc(8 * n)
        keep.append(sums)
        for mode, variant in ((ssim_amd.MODE_EXACT, 0), (ssim_amd.MODE_EXACT, 2), (ssim_amd.MODE_EXACT, 1), (ssim_amd.MODE_FAST, 0), (ssim_amd.MODE_SEPARABLE, 0), (ssim_amd.MODE_DOUBLE, 0)):
            gpu_ctx.set_mode(mode)
            gpu_ctx.set_tuning(0, variant)
            gpu_ctx.enqueue_batch(params, n, sums.ptr)
            gpu_ctx.synchronize()
            plain = sums.download(np.float64, (n,)).view(np.uint64).copy()
            gpu_ctx.set_profiling(True)
            gpu_ctx.get_profile_clock()
            for _ in range(4):
                gpu_ctx.enqueue_batch(params, n, sums.ptr)
            gpu_ctx.synchronize()
            launches, ms = gpu_ctx.get_profile()
            mhz, slowest, counted = gpu_ctx.get_profile_clock()
            gpu_ctx.set_profiling(False)
            assert launches == 4 and counted == 4 and 1500.0 < slowest <= mhz < 2450.0, (mode, variant, launches, counted, mhz, slowest)
            assert np.array_equal(sums.download(np.float64, (n,)).view(np.uint64), plain), (mode, variant)
            assert gpu_ctx.get_profile_clock() == (0.0, 0.0, 0)                # read once, then cleared
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        gpu_ctx.set_tuning(0, 0)
        gpu_ctx.set_profiling(False)
        for d in keep:
            d.free()


def test_tune_measures_the_candidates_and_keeps_the_bits(gpu_ctx):
    """rmgr_ssim_hip_tune: candidates[0] is the untuned default; the winner (if any beats it by > 0.5 %) is what later launches of the shape run --
    with the same per-image sums bit for bit; clear_tuned / explicit tuning take precedence."""
    from ssim_amd import synth
    w, h, n = 1920, 1080, 24
    keep, params = [], (ssim_amd.Params * n)()
    try:
        for i in range(n):
            da, db = gpu_ctx.alloc(w * h), gpu_ctx.alloc(w * h)
            gpu_ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, synth.BASE_SEED + i)
            keep += [da, db]
            params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
        sums = gpu_ctx.alloc(8 * n)
        keep.append(sums)

        def run():
            sums.upload(np.zeros(n))
            gpu_ctx.enqueue_batch(params, n, sums.ptr)
            gpu_ctx.synchronize()
            return sums.download(np.float64, (n,)).view(np.uint64).copy()
        for mode in (ssim_amd.MODE_EXACT, ssim_amd.MODE_SEPARABLE, ssim_amd.MODE_DOUBLE):
            gpu_ctx.set_mode(mode)
            gpu_ctx.clear_tuned()
            before = run()
            r = gpu_ctx.tune(w, h, n)
            assert 2 <= len(r["candidates"]) <= 8 and r["candidates"][0][:2] == (0, 0), r
            assert r["default_ms"] == r["candidates"][0][2], r
            assert r["best"] == (0, 0) or (r["best_ms"] == min(c[2] for c in r["candidates"]) and r["best_ms"] <= 0.995 * r["default_ms"]), r
            assert 0 < r["best_ms"] <= r["default_ms"] < 10.0 and r["default_ms"] / r["best_ms"] < 1.25, r          # the default is never far off
            assert np.array_equal(run(), before), mode                                                              # whatever was chosen: the same bits
            gpu_ctx.set_tuning(64, 2)
            assert np.array_equal(run(), before), mode
            gpu_ctx.set_tuning(0, 0)
            gpu_ctx.clear_tuned()
            assert np.array_equal(run(), before), mode
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        assert ssim_amd.finalize(run().view(np.float64)[:3], w, h).view(np.uint32).tolist() == [0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30]
        # with a map: the strips only; a shape of one small pair: the one-column kernel is among the candidates
        r = gpu_ctx.tune(2048, 2048, 2, True)
        assert all(v not in (6, 7) for v, _, _ in r["candidates"]), r
        r = gpu_ctx.tune(256, 256, 1)
        assert len(r["candidates"]) >= 2, r
        # a host that tuned once keeps the choices: read them back, install them on another context without measuring, same bits
        gpu_ctx.clear_tuned()
        assert gpu_ctx.tuned() == []
        gpu_ctx.set_tuned(w, h, n, False, 2, 64)                      # plain 64-row strips for the 24 x 1080p launch
        gpu_ctx.set_tuned(w, h, n, False, 3, 216)                     # replaced, not duplicated
        gpu_ctx.set_tuned(640, 480, 7, True, 1, 0)
        assert gpu_ctx.tuned() == [(w, h, n, False, ssim_amd.MODE_EXACT, 3, 216), (640, 480, 7, True, ssim_amd.MODE_EXACT, 1, 0)]
        tuned_bits = run()
        gpu_ctx.clear_tuned()
        assert np.array_equal(run(), tuned_bits)
        lib = ssim_amd.load_library()
        e = ssim_amd.TunedEntry()
        assert lib.rmgr_ssim_hip_get_tuned(gpu_ctx.handle, 0, ctypes.byref(e)) == errno.ENOENT
        for bad in ((0, 8, 1, 0, 2, 8), (8, 8, 0, 0, 2, 8), (8, 8, 1, 0, -1, 8), (8, 8, 1, 0, 0, 0)):
            assert lib.rmgr_ssim_hip_set_tuned(gpu_ctx.handle, *bad) == errno.EINVAL, bad
        assert lib.rmgr_ssim_hip_tune(gpu_ctx.handle, 0, 16, 1, 0, None) == errno.EINVAL and lib.rmgr_ssim_hip_tune(None, 16, 16, 1, 0, None) == errno.EINVAL
        assert lib.rmgr_ssim_hip_tune(gpu_ctx.handle, 300, 200, 3, 0, None) == 0          # result may be NULL
    finally:
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        gpu_ctx.set_tuning(0, 0)
        gpu_ctx.clear_tuned()
        for d in keep:
            d.free()
