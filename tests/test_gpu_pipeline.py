"""GPU: round-2 additions to the C ABI.

* BASELINE.json configs[3] at full count (1024 x 1080p): known answers, 40 pairs against the oracle, and the 8-GPU
  sharding emulated on one device -- no communicator anywhere (the RCCL exchange is tests/test_gpu_zz_rccl.py);
* the cell-based fp64 reduction: per-image sums bit-identical for any strip height, kernel variant, batch split;
* the descriptor-table ring: different batches enqueued back to back without draining the stream;
* the banded (pipelined) host-pointer call: value and map bit-identical to the device path, for every map layout;
* (the accuracy contracts of MODE_FAST / MODE_SEPARABLE / MODE_DOUBLE live in tests/test_gpu_refsets.py and test_gpu_modes.py)
* the synthetic generator kernel against its host twins.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import ssim_amd
from ssim_amd import synth
from conftest import GOLDEN, ROOT, f32_hex, image_entries, load_pair

pytestmark = pytest.mark.gpu


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


CONFIG4_KATS = (0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30)      # the reference's FMA path on 1080p seeds 0x5EED, 0x5EEE, 0x5EEF


def config4_checked_pairs(total):
    """The pairs of the configs[3] batch that are compared with the oracle: both ends, every boundary of the 8-way shard
    table (the last pair of one rank and the first of the next), and pseudo-randomly chosen ones in between: 40 in all."""
    from ssim_amd import sharding
    picks = {0, 1, 2, total - 1}
    for first, last in sharding.split_batch(total, 8):
        picks.update(i for i in (first - 1, first, last - 1, last) if 0 <= i < total)
    rng = np.random.default_rng(0xC4)
    while len(picks) < min(40, total):
        picks.add(int(rng.integers(0, total)))
    return sorted(picks)


def test_config4_1024_pairs_1080p_sharded_equals_single_batch(gpu_ctx, oracle):
    """BASELINE.json configs[3] at its stated size on ONE GPU and without any communicator: all 1024 synthetic 1920x1080
    pairs (seeds 0x5EED + i) resident in HBM.
      1. one batch of 1024 through rmgr_ssim_hip_enqueue_batch -> the per-image fp64 sums;
      2. the reference's FMA-path known answers for pairs 0, 1, 2;
      3. 40 pairs -- ends, all eight shard boundaries (127/128 ... 895/896), random picks -- against the ORACLE on the same
         pixels: the device-generated images equal their host twins byte for byte, the global float is bit-equal, the fp64
         sum agrees to 1e-13 relative (only the summation order differs);
      4. the 8-GPU run emulated: the 8 shards of sharding.split_batch, each enqueued as its own batch into ITS slice of one
         zeroed double[1024] (what the other ranks would add in the all-reduce is exact zeros) == (1) bit for bit; likewise
         an uneven 3-way split and a 5-way split at another strip height.
    The RCCL all-reduce on top of (4) is tests/test_gpu_zz_rccl.py's business: nothing here can wait for a peer."""
    from ssim_amd import sharding
    W, H, total = 1920, 1080, 1024
    ctx = gpu_ctx
    imgs = ctx.alloc(2 * W * H * total)
    single = ctx.alloc(8 * total)
    try:
        params = (ssim_amd.Params * total)()
        for i in range(total):
            a = imgs.ptr + 2 * W * H * i
            ctx.synth_pair(a, W, a + W * H, W, W, H, synth.BASE_SEED + i)
            params[i] = ssim_amd.make_params(W, H, a, 1, W, a + W * H, 1, W)
        ctx.enqueue_batch(params, total, single.ptr)
        ctx.synchronize()
        s_single = single.download(np.float64, (total,))
        res = ssim_amd.finalize(s_single, W, H)
        for i, k in enumerate(CONFIG4_KATS):
            assert int(res[i].view(np.uint32)) == k, "pair %d: 0x%08x, want 0x%08x" % (i, int(res[i].view(np.uint32)), k)
        assert np.all(np.isfinite(res)) and res.min() > 0.88 and res.max() < 0.90, (res.min(), res.max())

        threads = oracle.oracle_lib().oracle_max_threads()
        checked = config4_checked_pairs(total)
        assert {0, 127, 128, 895, 896, 1023} <= set(checked) and len(checked) >= 32
        for i in checked:
            ha, hb = oracle.synth_pair(W, H, synth.BASE_SEED + i)
            base = imgs.ptr + 2 * W * H * i
            assert np.array_equal(ctx.download(base, np.uint8, (H, W)), ha), "generator mismatch (A, pair %d)" % i
            assert np.array_equal(ctx.download(base + W * H, np.uint8, (H, W)), hb), "generator mismatch (B, pair %d)" % i
            ov, osum, _ = oracle.ssim_f32(ha, hb, threads=threads)
            assert f32_hex(res[i]) == f32_hex(ov), "pair %d: GPU %s, oracle %s" % (i, f32_hex(res[i]), f32_hex(ov))
            assert abs(s_single[i] - osum) <= 1e-13 * abs(osum), (i, s_single[i], osum)

        for world, strip_rows in ((8, 0), (3, 0), (5, 64)):
            ctx.set_tuning(strip_rows, 0)
            vec = ctx.alloc(8 * total).upload(np.zeros(total, np.float64))
            try:
                for first, last in sharding.split_batch(total, world):
                    if last > first:
                        shard = (ssim_amd.Params * (last - first))(*[params[i] for i in range(first, last)])
                        ctx.enqueue_batch(shard, last - first, vec.ptr + 8 * first)
                ctx.synchronize()
                s_sharded = vec.download(np.float64, (total,))
            finally:
                vec.free()
                ctx.set_tuning(0, 0)
            bad = np.flatnonzero(bits64(s_sharded) != bits64(s_single))
            assert bad.size == 0, "%d-way split: %d of %d sums differ from the single batch (first at %d)" % (world, bad.size, total, bad[0])
    finally:
        single.free()
        imgs.free()


def test_synth_generator_matches_host_twins(gpu_ctx, oracle):
    for (w, h, seed) in ((301, 77, 0x5EED), (1920, 9, 0x5EEE), (64, 64, 0x5EED + 1023), (1, 1, 7)):
        da, db = gpu_ctx.alloc(w * h), gpu_ctx.alloc(w * h)
        try:
            gpu_ctx.synth_pair(da.ptr, w, db.ptr, w, w, h, seed)
            gpu_ctx.synchronize()
            a, b = da.download(np.uint8, (h, w)), db.download(np.uint8, (h, w))
        finally:
            da.free()
            db.free()
        oa, ob = oracle.synth_pair(w, h, seed)
        na, nb = synth.pair_numpy(w, h, seed)
        assert np.array_equal(a, oa) and np.array_equal(b, ob), (w, h, seed)
        assert np.array_equal(a, na) and np.array_equal(b, nb), (w, h, seed)
    da, db = gpu_ctx.alloc(16), gpu_ctx.alloc(16)
    gpu_ctx.synth_pair(da.ptr, 4, db.ptr, 4, 4, 1, 0x5EED)
    gpu_ctx.synchronize()
    assert list(da.download(np.uint8, (4,))) == [45, 59, 51, 6] and list(db.download(np.uint8, (4,))) == [51, 58, 48, 5]
    da.free()
    db.free()


def hostile_pairs(rng, w, h, n):
    """Pairs whose per-pixel SSIM values span [-1, 1] with many tiny magnitudes: fp64 partial sums of such values
    are NOT exact, so any change of the summation grouping shows up in the last bits of the per-image sum."""
    out = []
    for i in range(n):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        if i % 3 == 0:
            b = 255 - a                                  # anti-correlated: values near -1 .. 0
        elif i % 3 == 1:
            b = rng.integers(0, 256, (h, w), dtype=np.uint8)   # uncorrelated: values around 0
        else:
            b = np.clip(a.astype(np.int32) + rng.integers(-90, 91, (h, w)), 0, 255).astype(np.uint8)
        out.append((a, b))
    return out


def batch_sums(ctx, pairs, keep, split=None):
    """fp64 per-image sums of host pairs through enqueue_batch; `split`: list of (first, last) sub-batches."""
    n = len(pairs)
    h, w = pairs[0][0].shape
    params = (ssim_amd.Params * n)()
    for i, (a, b) in enumerate(pairs):
        da, db = ctx.upload(a), ctx.upload(b)
        keep += [da, db]
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
    sums = ctx.alloc(8 * n).upload(np.zeros(n))
    keep.append(sums)
    for first, last in (split or [(0, n)]):
        sub = (ssim_amd.Params * (last - first))(*[params[i] for i in range(first, last)])
        ctx.enqueue_batch(sub, last - first, sums.ptr + 8 * first)
    ctx.synchronize()
    return sums.download(np.float64, (n,))


@pytest.mark.parametrize("size", [(333, 411), (130, 1100), (200, 2100)])      # 8-row and 32-row reduction cells, each with a short last cell
@pytest.mark.parametrize("mode", [ssim_amd.MODE_EXACT, ssim_amd.MODE_FAST, ssim_amd.MODE_DOUBLE, ssim_amd.MODE_SEPARABLE])
def test_sums_do_not_depend_on_strips_variant_or_batch_split(gpu_ctx, mode, size):
    rng = np.random.default_rng(20260101 + mode)
    pairs = hostile_pairs(rng, size[0], size[1], 6)     # ragged strips in both directions
    keep = []
    gpu_ctx.set_mode(mode)
    try:
        gpu_ctx.set_tuning(0, 0)
        ref = batch_sums(gpu_ctx, pairs, keep)
        assert np.all(np.isfinite(ref))
        for strip_rows, variant, split in ((8, 2, None), (24, 0, None), (64, 1, None), (216, 2, None), (512, 1, None),
                                           (0, 1, None), (0, 2, None), (0, 3, None), (48, 3, None), (0, 0, [(0, 1), (1, 4), (4, 6)]), (40, 0, [(0, 5), (5, 6)])):
            gpu_ctx.set_tuning(strip_rows, variant)
            got = batch_sums(gpu_ctx, pairs, keep, split)
            assert np.array_equal(bits64(got), bits64(ref)), (mode, strip_rows, variant, split, got - ref)
    finally:
        gpu_ctx.set_tuning(0, 0)
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        for d in keep:
            d.free()


@pytest.mark.parametrize("shape", [(640, 360, 120), (130, 2049, 20), (1920, 1080, 24), (257, 65, 700), (512, 512, 16), (2048, 2048, 3)],
                         ids=["8-row-cells", "32-row-cells-short-last", "default-rule-1080p", "many-small", "even-chunks-8-row-cells", "even-chunks-32-row-cells"])
def test_balanced_schedule_equals_the_strips(gpu_ctx, shape):
    """Round 5: the balanced schedule of the two-waves-per-SIMD two-column kernels (launches without a map: one round of equal chunks of
    the flattened [image][strip column][cell row] list, wavefronts continuing into the next strip column or image) must give the strips'
    per-image fp64 sums bit for bit -- forced (tuning variant 6) and as plan()'s default choice -- on hostile values, in both
    bit-exact modes and MODE_FAST, for chunk boundaries inside images, at image ends and with short last cells.  MODE_SEPARABLE has
    no balanced form (it loses with it: profiles/r05_balanced_sweep_modes.txt): variant 6 there runs the strips."""
    w, h, n = shape
    rng = np.random.default_rng(w * 7 + h + n)
    base = hostile_pairs(rng, w, h, 6)
    pairs = [base[i % 6] if i % 7 else (base[i % 6][1], base[(i + 1) % 6][0]) for i in range(n)]      # n pairs out of 6 x 2 uploaded planes
    keep = []
    try:
        planes = {}
        params = (ssim_amd.Params * n)()
        for i, (a, b) in enumerate(pairs):
            for img in (a, b):
                if id(img) not in planes:
                    planes[id(img)] = gpu_ctx.upload(img)
                    keep.append(planes[id(img)])
            params[i] = ssim_amd.make_params(w, h, planes[id(a)].ptr, 1, w, planes[id(b)].ptr, 1, w)
        sums = gpu_ctx.alloc(8 * n)
        keep.append(sums)
        for mode in (ssim_amd.MODE_EXACT, ssim_amd.MODE_UNFUSED, ssim_amd.MODE_FAST, ssim_amd.MODE_SEPARABLE):
            gpu_ctx.set_mode(mode)
            res = {}
            # 6: the balanced schedule with plan()'s interleave of the images in the chunk list; 7: the plain list (round 5); 100 + T: T images interleaved
            for variant, rows in ((2, 0), (3, 0), (6, 0), (7, 0), (102, 0), (105, 0), (164, 0), (0, 0), (6, 64), (103, 64), (2, 8)):
                gpu_ctx.set_tuning(rows, variant)
                if variant in (6, 7) or variant >= 100:
                    chunks = ssim_amd.get_plan(w, h, n, gpu_ctx).balancedChunks
                    assert (chunks == 0) if mode == ssim_amd.MODE_SEPARABLE else (chunks > 0), "the shape is meant to engage the balanced schedule"
                sums.upload(np.zeros(n))
                gpu_ctx.enqueue_batch(params, n, sums.ptr)
                gpu_ctx.synchronize()
                res[(variant, rows)] = bits64(sums.download(np.float64, (n,)))
            for key, got in res.items():
                assert np.array_equal(got, res[(2, 0)]), (mode, key, int((got != res[(2, 0)]).sum()))
        gpu_ctx.set_tuning(0, 0)
        if shape == (1920, 1080, 24):
            for mode in (ssim_amd.MODE_EXACT, ssim_amd.MODE_FAST):
                gpu_ctx.set_mode(mode)
                assert ssim_amd.get_plan(w, h, n, gpu_ctx).balancedChunks > 0    # plan()'s default for this shape (on a 256-CU device)
        if shape in ((512, 512, 16), (2048, 2048, 3)):      # the chunk divides the strip column evenly: the default of the bit-exact modes, not of MODE_FAST
            for mode, taken in ((ssim_amd.MODE_EXACT, True), (ssim_amd.MODE_UNFUSED, True), (ssim_amd.MODE_FAST, False)):
                gpu_ctx.set_mode(mode)
                assert (ssim_amd.get_plan(w, h, n, gpu_ctx).balancedChunks > 0) == taken, mode
    finally:
        gpu_ctx.set_tuning(0, 0)
        gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
        for d in keep:
            d.free()


def test_default_plan_equals_the_plain_strips_on_random_shapes(gpu_ctx):
    """Whatever plan() decides for a launch without a map -- strips of any height, EARLY or not, the balanced form by either of its rules -- is scheduling only: the per-image
    fp64 sums of the default tuning must be the bits of the plain strips (tuning variant 2, 64-row strips) on 24 random launch shapes (power-of-two sizes, video sizes, ragged
    ones; 1 ... 96 pairs), in the bit-exact modes and MODE_FAST."""
    rng = np.random.default_rng(20261007)
    shapes = []
    for i in range(24):
        kind = i % 3
        if kind == 0:
            w, h = int(2 ** rng.integers(6, 12)), int(2 ** rng.integers(6, 12))
        elif kind == 1:
            w, h = [(640, 480), (1280, 720), (1920, 1080), (2560, 1440), (1600, 1200)][int(rng.integers(0, 5))]
        else:
            w, h = int(rng.integers(1, 1800)), int(rng.integers(1, 2400))
        n = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40, 64, 96]))
        if w * h * n <= 1 << 26:
            shapes.append((w, h, n))
    balanced = 0
    for (w, h, n) in shapes:
        base = hostile_pairs(rng, w, h, 2)
        keep = []
        try:
            planes = [gpu_ctx.upload(x) for pair in base for x in pair]
            keep += planes
            params = (ssim_amd.Params * n)()
            for i in range(n):
                a, b = planes[2 * (i % 2)], planes[2 * (i % 2) + 1] if i % 3 else planes[(2 * (i % 2) + 3) % 4]
                params[i] = ssim_amd.make_params(w, h, a.ptr, 1, w, b.ptr, 1, w)
            sums = gpu_ctx.alloc(8 * n)
            keep.append(sums)
            for mode in (ssim_amd.MODE_EXACT, ssim_amd.MODE_UNFUSED, ssim_amd.MODE_FAST):
                gpu_ctx.set_mode(mode)
                got = {}
                for variant, rows in ((0, 0), (2, 64)):
                    gpu_ctx.set_tuning(rows, variant)
                    sums.upload(np.zeros(n))
                    gpu_ctx.enqueue_batch(params, n, sums.ptr)
                    gpu_ctx.synchronize()
                    got[variant] = bits64(sums.download(np.float64, (n,)))
                assert np.array_equal(got[0], got[2]), (w, h, n, mode, int((got[0] != got[2]).sum()))
                gpu_ctx.set_tuning(0, 0)
                balanced += ssim_amd.get_plan(w, h, n, gpu_ctx).balancedChunks > 0
        finally:
            gpu_ctx.set_tuning(0, 0)
            gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
            for d in keep:
                d.free()
    assert balanced >= 6, balanced            # the draw reaches the balanced form


def test_row_bands_without_a_map_take_the_balanced_form_and_equal_the_strips(gpu_ctx):
    """Round 5: plan() gives a launch without a map the balanced form wherever its chunks divide the strip column evenly -- row bands
    (rmgr_ssim_hip_enqueue_rows) included: a band of 2048 rows of an 8192-column image is 64 x 64 cell rows = 2 per wave slot.  The band's
    chunks start at the band's first row (y_begin > 0), not at the image's: cells of two bands under the default tuning (chunks) must be the
    cells of the forced strips (tuning variant 2), bit for bit, and add up to the whole image's."""
    w, h = 8192, 4096
    rng = np.random.default_rng(8192)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-60, 61, (h, w)), 0, 255).astype(np.uint8)
    keep = []
    try:
        da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
        keep += [da, db]
        p = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)                 # no map
        plan = ssim_amd.get_plan(w, h, 1, gpu_ctx)
        ncell = plan.cellsX * plan.cellsY
        assert plan.balancedChunks > 0 and plan.cellRows == 32
        got = {}
        for variant in (0, 2):
            gpu_ctx.set_tuning(0, variant)
            parts = []
            for (y0, rows) in ((0, 2048), (2048, 2048)):
                cells = gpu_ctx.alloc(8 * ncell).upload(np.zeros(ncell, np.float64))
                keep.append(cells)
                gpu_ctx.enqueue_rows(p, y0, rows, cells.ptr)
                gpu_ctx.synchronize()
                parts.append(cells.download(np.float64, (ncell,)))
            assert not parts[0][ncell // 2:].any() and not parts[1][:ncell // 2].any()      # each band wrote its own cell rows only
            got[variant] = parts
        for k in (0, 1):
            assert np.array_equal(bits64(got[0][k]), bits64(got[2][k])), "band %d: chunks and strips disagree in %d cells" % (k, int((bits64(got[0][k]) != bits64(got[2][k])).sum()))
        gpu_ctx.set_tuning(0, 0)
        whole = gpu_ctx.alloc(8 * ncell).upload(np.zeros(ncell, np.float64))
        keep.append(whole)
        gpu_ctx.enqueue_rows(p, 0, h, whole.ptr)
        gpu_ctx.synchronize()
        assert np.array_equal(bits64(whole.download(np.float64, (ncell,))), bits64(got[0][0] + got[0][1]))
    finally:
        gpu_ctx.set_tuning(0, 0)
        for d in keep:
            d.free()


@pytest.mark.parametrize("size", [(1500, 1090), (2100, 2300), (4096, 4100)], ids=["8-row-cells", "32-row-cells", "two-reduce-chunks"])
def test_row_bands_on_separate_contexts_equal_one_launch(gpu_ctx, size):
    """SURVEY.md 8(e), "single huge image across GPUs": one pair cut into row bands (rmgr_ssim_hip_enqueue_rows), every band
    computed by its OWN context -- as another GPU would -- from a buffer that holds nothing but the band's rows and 5 halo rows
    each side, into its own zeroed cell array; the arrays added element-wise (what the all-reduce does: every cell is non-zero
    on one rank only) and reduced (rmgr_ssim_hip_reduce_cells) must give the bits of the single launch over the whole image,
    and the bands' map rows the same map.  Hostile values (fp64 sums inexact), three arithmetic modes, 2 / 3 / 5 bands."""
    w, h = size
    rng = np.random.default_rng(w * 31 + h)
    a, b = hostile_pairs(rng, w, h, 3)[2]
    for mode in (ssim_amd.MODE_EXACT, ssim_amd.MODE_SEPARABLE, ssim_amd.MODE_DOUBLE):
        gpu_ctx.set_mode(mode)
        keep = []
        try:
            da, db, dm, ds = gpu_ctx.upload(a), gpu_ctx.upload(b), gpu_ctx.alloc(4 * w * h), gpu_ctx.alloc(8)
            keep += [da, db, dm, ds]
            p = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w, dm.ptr, 1, w)
            one = (ssim_amd.Params * 1)(p)
            gpu_ctx.enqueue_batch(one, 1, ds.ptr)
            gpu_ctx.synchronize()
            want_sum = ds.download(np.float64, (1,))
            want_map = dm.download(np.float32, (h, w))
            plan = ssim_amd.get_plan(w, h, 1, gpu_ctx)
            cr, ncell = plan.cellRows, plan.cellsX * plan.cellsY
            assert cr == (32 if h >= 2048 else 8) and plan.cellsX == (w + 63) // 64 and plan.cellsY == (h + cr - 1) // cr
            for world in (2, 3, 5):
                from ssim_amd import sharding
                bands = sharding.split_rows(h, cr, world)
                total = np.zeros(ncell, np.float64)
                band_map = gpu_ctx.alloc(4 * w * h)
                keep.append(band_map)
                for r in range(world):
                    y0, y1 = bands[r]
                    lo, hi = max(0, y0 - 5), min(h, y1 + 5)                  # the source rows this band may read
                    rank = ssim_amd.Context(0, mode=mode)                   # a context of its own: nothing shared with the other bands
                    try:
                        ra, rb = rank.upload(a[lo:hi]), rank.upload(b[lo:hi])
                        cells = rank.alloc(8 * ncell).upload(np.zeros(ncell, np.float64))
                        rp = ssim_amd.make_params(w, h, ra.ptr - lo * w, 1, w, rb.ptr - lo * w, 1, w, band_map.ptr, 1, w)
                        rank.enqueue_rows(rp, y0, y1 - y0, cells.ptr)
                        rank.synchronize()
                        part = cells.download(np.float64, (ncell,))
                        first, last = (y0 // cr) * plan.cellsX, ((y1 + cr - 1) // cr) * plan.cellsX
                        assert not part[:first].any() and not part[last:].any(), "band %d of %d wrote cells outside its rows" % (r, world)
                        total += part                                       # adding zeros: exact
                        for d in (ra, rb, cells):
                            d.free()
                    finally:
                        rank.close()
                dc = gpu_ctx.upload(total)
                keep.append(dc)
                gpu_ctx.reduce_cells(w, h, 1, dc.ptr, ds.ptr)
                gpu_ctx.synchronize()
                got = ds.download(np.float64, (1,))
                assert np.array_equal(bits64(got), bits64(want_sum)), (mode, world, got, want_sum)
                assert np.array_equal(band_map.download(np.float32, (h, w)).view(np.uint32), want_map.view(np.uint32)), (mode, world)
            # bands start and end on cell boundaries (or at the last row)
            lib = ssim_amd.load_library()
            cells = gpu_ctx.alloc(8 * ncell)
            keep.append(cells)
            import errno
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), cr + 1, cr, cells.ptr) == errno.EINVAL
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), cr, cr + 3, cells.ptr) == errno.EINVAL
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), h + cr, cr, cells.ptr) == errno.EINVAL
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), cr, 0xFFFFFFFF, cells.ptr) == 0      # "to the end"
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), 0, cr, None) == errno.EINVAL
            # an empty band is a no-op whatever its alignment: split_rows hands ranks beyond the image's cell rows (h, h),
            # and h is not a multiple of the cell height here
            assert h % cr != 0 and sharding.split_rows(h, cr, plan.cellsY + 3)[-1] == (h, h)
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), h, 0, cells.ptr) == 0
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), h, cr, cells.ptr) == 0
            assert lib.rmgr_ssim_hip_enqueue_rows(gpu_ctx.handle, ctypes.byref(p), cr + 1, 0, cells.ptr) == 0
            gpu_ctx.synchronize()
        finally:
            gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
            for d in keep:
                d.free()


def test_different_batches_back_to_back(gpu_ctx, oracle):
    """Six different batches (more than the descriptor ring holds) enqueued without any synchronisation in between,
    then two of them alternating: every result must be the one of its own batch."""
    rng = np.random.default_rng(5)
    w, h = 200, 96
    keep, batches, want = [], [], []
    try:
        for k in range(6):
            n = 2 + k % 3
            params = (ssim_amd.Params * n)()
            sums = []
            for i in range(n):
                a = rng.integers(0, 256, (h, w), dtype=np.uint8)
                b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, (h, w)), 0, 255).astype(np.uint8)
                da, db = gpu_ctx.upload(a), gpu_ctx.upload(b)
                keep += [da, db]
                params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
                sums.append(oracle.ssim_f32(a, b)[0])
            out = gpu_ctx.alloc(8 * n)
            keep.append(out)
            batches.append((params, n, out))
            want.append(np.array(sums, np.float32))
        for rounds in range(3):
            for params, n, out in batches:
                gpu_ctx.enqueue_batch(params, n, out.ptr)
        for _ in range(10):
            for k in (4, 1):
                gpu_ctx.enqueue_batch(batches[k][0], batches[k][1], batches[k][2].ptr)
        gpu_ctx.synchronize()
        for (params, n, out), w_ in zip(batches, want):
            got = ssim_amd.finalize(out.download(np.float64, (n,)), w, h)
            assert np.array_equal(got.view(np.uint32), w_.view(np.uint32))
    finally:
        for d in keep:
            d.free()


def host_call(a, b, a_step, a_stride, a_ptr, b_step, b_stride, b_ptr, w, h, m_ptr=None, m_step=1, m_stride=None, want_global=True):
    lib = ssim_amd.load_library()
    p = ssim_amd.make_params(w, h, a_ptr, a_step, a_stride, b_ptr, b_step, b_stride, m_ptr, m_step, m_stride)
    out = ctypes.c_float(-2.0)
    rc = lib.rmgr_ssim_compute_ssim(ctypes.byref(out) if want_global else None, ctypes.byref(p), None)
    assert rc == 0, rc
    return np.float32(out.value)


@pytest.mark.parametrize("w,h", [(4096, 4096), (3000, 1237), (1031, 260)])
def test_banded_host_call_equals_device_path(gpu_ctx, oracle, w, h):
    """The host-pointer call with a map on images large enough for the banded pipeline: global value and every map
    pixel bit-identical to the one-launch device path -- dense map, padded rows, a column-interleaved map (the
    CPU-scatter path), bottom-up images, and map-only calls.  4096^2 is also checked against the known answer."""
    a, b = oracle.synth_pair(w, h, 0x5EED)
    gpu_ctx.set_mode(ssim_amd.MODE_EXACT)
    v_dev, m_dev = gpu_ctx.ssim_planes(a, b, want_map=True)
    if (w, h) == (4096, 4096):
        assert f32_hex(v_dev) == "0x3f64b7be"
    # dense
    m = np.full((h, w), -7.0, np.float32)
    v = host_call(a, b, 1, w, a.ctypes.data, 1, w, b.ctypes.data, w, h, m.ctypes.data, 1, w)
    assert f32_hex(v) == f32_hex(v_dev)
    assert np.array_equal(m.view(np.uint32), m_dev.view(np.uint32))
    # padded map rows + map-only call (ssim == NULL)
    pad = np.full((h, w + 13), -7.0, np.float32)
    host_call(a, b, 1, w, a.ctypes.data, 1, w, b.ctypes.data, w, h, pad.ctypes.data, 1, w + 13, want_global=False)
    assert np.array_equal(pad[:, :w].view(np.uint32), m_dev.view(np.uint32)) and np.all(pad[:, w:] == -7.0)
    # map elements two floats apart (scatter path), images bottom-up (negative stride)
    inter = np.full((h, 2 * w), -7.0, np.float32)
    af, bf = np.ascontiguousarray(a[::-1]), np.ascontiguousarray(b[::-1])
    v2 = host_call(af, bf, 1, -w, af.ctypes.data + (h - 1) * w, 1, -w, bf.ctypes.data + (h - 1) * w, w, h, inter.ctypes.data, 2, 2 * w)
    assert f32_hex(v2) == f32_hex(v_dev)
    assert np.array_equal(inter[:, 0::2].view(np.uint32), m_dev.view(np.uint32)) and np.all(inter[:, 1::2] == -7.0)
    # a bottom-up map (negative ssimStride): rows land in reverse order
    flip = np.full((h, w), -7.0, np.float32)
    v3 = host_call(a, b, 1, w, a.ctypes.data, 1, w, b.ctypes.data, w, h, flip.ctypes.data + 4 * (h - 1) * w, 1, -w)
    assert f32_hex(v3) == f32_hex(v_dev)
    assert np.array_equal(flip[::-1].view(np.uint32), m_dev.view(np.uint32))
    # every band count the pipeline can be asked for gives the same bits (own processes: the default context reads the env once)
    if (w, h) == (3000, 1237):
        code = ("import sys, numpy as np; sys.path.insert(0, %r); import oracle, ssim_amd; a,b=oracle.synth_pair(%d,%d,0x5EED); "
                "v,m=ssim_amd.compute_ssim(a,b,want_map=True); import hashlib; print('%%08x'%%int(np.float32(v).view(np.uint32)), hashlib.sha256(m.tobytes()).hexdigest())" % (ROOT, w, h))
        import hashlib
        want = "%08x %s" % (int(np.float32(v_dev).view(np.uint32)), hashlib.sha256(m_dev.tobytes()).hexdigest())
        for bands in ("1", "2", "3", "7", "16"):
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, RMGR_SSIM_HIP_BANDS=bands))
            assert r.returncode == 0, r.stderr[-800:]
            assert r.stdout.strip().splitlines()[-1] == want, (bands, r.stdout)


def test_select_impl_switches_the_arithmetic_of_the_dropin_call(manifest):
    """The reference's test-only selector (src/ssim_internal.h:41-53, src/ssim.cpp:808-896) keeps its meaning: after
    select_impl(IMPL_AVX) -- or GENERIC / SSE / SSE2 -- the unchanged rmgr_ssim_compute_ssim returns the reference AVX
    path's bits, after select_impl(IMPL_FMA) the FMA path's; AVX-512 and NEON are reported unsupported."""
    lib = ssim_amd.load_library()
    fn = getattr(lib, "_ZN4rmgr4ssim11select_implENS0_14ImplementationE")
    fn.argtypes, fn.restype = [ctypes.c_int], ctypes.c_uint
    AUTO, GENERIC, SSE, SSE2, AVX, FMA, AVX512, NEON, HIP = range(9)
    names = [n for n in image_entries(manifest) if manifest[n]["fma"]["ssim_hex"] != manifest[n]["avx"]["ssim_hex"]][:4]
    assert names
    try:
        for impl, key in ((AVX, "avx"), (FMA, "fma"), (GENERIC, "avx"), (SSE2, "avx"), (AUTO, "fma"), (SSE, "avx"), (HIP, "fma")):
            mask = fn(impl)
            assert mask & (1 << impl), (impl, mask)
            assert not mask & ((1 << AVX512) | (1 << NEON))
            for n in names:
                a, b = load_pair(manifest[n])
                v, m = ssim_amd.compute_ssim(a, b, want_map=True)
                assert f32_hex(v) == manifest[n][key]["ssim_hex"], (impl, n)
                import hashlib
                assert hashlib.sha256(m.tobytes()).hexdigest() == manifest[n][key]["map_sha256"], (impl, n)
        assert not fn(AVX512) & (1 << AVX512)          # unsupported; the generic arithmetic is selected, as in the reference
        a, b = load_pair(manifest[names[0]])
        assert f32_hex(ssim_amd.compute_ssim(a, b)[0]) == manifest[names[0]]["avx"]["ssim_hex"]
    finally:
        fn(AUTO)
