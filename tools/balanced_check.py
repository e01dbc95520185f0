"""The balanced schedule (tuning variants 6 and 7: a chunk's segments last to first / in list order) against the strips (variants 2, 3): per-image fp64 sums of batches without a map, bit for bit,
over sizes (ragged, shorter than a cell, many cells, power-of-two sizes where plan() takes the chunks by default; 18 fixed + 24 random shapes),
batch sizes, the four fp32 modes and strip heights.  Run on the GPU box."""
import sys, numpy as np
sys.path.insert(0, '.')
import ssim_amd
ctx = ssim_amd.Context(0)
rng = np.random.default_rng(11)
bad = 0
shapes = [(256, 256, 1), (300, 301, 3), (1, 1, 5), (17, 5, 40), (129, 64, 7), (1920, 1080, 3), (1920, 1080, 40), (4096, 4096, 2), (1000, 37, 33),
          (130, 2049, 9), (640, 360, 300), (255, 63, 128), (2048, 2048, 5), (512, 512, 16), (256, 256, 64), (1024, 1024, 4), (4096, 2048, 1), (8192, 4096, 1)]
# + random shapes: power-of-two sizes (plan()'s even-chunks rule engages by default) and ragged ones, 1 ... 80 pairs
for _ in range(24):
    if rng.integers(0, 2):
        shapes.append((int(2 ** rng.integers(6, 12)), int(2 ** rng.integers(6, 12)), int(rng.integers(1, 81))))
    else:
        shapes.append((int(rng.integers(1, 1500)), int(rng.integers(1, 2600)), int(rng.integers(1, 41))))
shapes = [(w, h, n) for (w, h, n) in shapes if w * h * n <= 1 << 27]
for (w, h, n) in shapes:
    keep, params = [], (ssim_amd.Params * n)()
    for i in range(n):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = np.clip(a.astype(np.int32) + rng.integers(-40, 41, (h, w)), 0, 255).astype(np.uint8)
        da, db = ctx.upload(a), ctx.upload(b)
        keep += [da, db]
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w, None, 1, w)
    ds = ctx.alloc(8 * n)
    for mode in (0, 3, 1, 4):
        ctx.set_mode(mode)
        res = {}
        for v in (2, 3, 6, 7, 102, 103, 109, 140, 0):
            for rows in (0, 8, 64):
                ctx.set_tuning(rows, v)
                ds.upload(np.zeros(n, np.float64))
                ctx.enqueue_batch(params, n, ds.ptr); ctx.synchronize()
                res[(v, rows)] = ds.download(np.float64, (n,)).copy().view(np.uint64)
        ref = res[(2, 0)]
        for k, sm in res.items():
            if not np.array_equal(sm, ref):
                bad += 1; print("MISMATCH", w, h, n, mode, k, int((sm != ref).sum()))
    for d in keep + [ds]: d.free()
print("balanced schedule check: %d shapes, mismatches" % len(shapes), bad)
ctx.close()
sys.exit(1 if bad else 0)
