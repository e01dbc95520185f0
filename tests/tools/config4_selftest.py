#!/usr/bin/env python3
"""BASELINE.json configs[3] at its stated size on ONE GPU: all 1024 synthetic 1920x1080 pairs (seeds 0x5EED + i).

  1. one batch of 1024 through rmgr_ssim_hip_enqueue_batch -> the per-image fp64 sums;
  2. the reference's FMA-path known answers for pairs 0, 1, 2 (0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30);
  3. the 8-GPU run emulated: 8 shards of 128 pairs (sharding.split_batch), each enqueued as its own batch into ITS slice
     of one zeroed double[1024], then the RCCL all-reduce of the whole vector (rmgr_ssim_hip_comm_allreduce_sums on a
     1-rank communicator: what the other ranks would add is exact zeros) -> must equal (1) bit for bit;
  4. the same for an uneven 3-way split and a 5-way split run with a different strip height.

Own process, no torch (torch wheels bundle their own RCCL/HSA runtime; see tools/rccl_selftest.py).  Prints "ok".
usage: python3 tests/tools/config4_selftest.py [pairs]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import sharding, synth  # noqa: E402

W, H = 1920, 1080
KATS = (0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30)


def main():
    total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ctx = ssim_amd.Context(0)
    t0 = time.time()
    imgs = ctx.alloc(2 * W * H * total)
    params = (ssim_amd.Params * total)()
    for i in range(total):
        a = imgs.ptr + 2 * W * H * i
        b = a + W * H
        ctx.synth_pair(a, W, b, W, W, H, synth.BASE_SEED + i)
        params[i] = ssim_amd.make_params(W, H, a, 1, W, b, 1, W)
    ctx.synchronize()
    # the generator itself against its numpy twin, on the first rows of two pairs
    for i in (0, total - 1):
        na, nb = synth.pair_numpy(W, 4, synth.BASE_SEED + i)
        base = imgs.ptr + 2 * W * H * i
        assert np.array_equal(ctx.download(base, np.uint8, (4, W)), na), "generator mismatch (A, pair %d)" % i
        assert np.array_equal(ctx.download(base + W * H, np.uint8, (4, W)), nb), "generator mismatch (B, pair %d)" % i
    t_gen = time.time() - t0

    single = ctx.alloc(8 * total)
    t0 = time.time()
    ctx.enqueue_batch(params, total, single.ptr)
    ctx.synchronize()
    t_batch = time.time() - t0
    s_single = single.download(np.float64, (total,))
    res = ssim_amd.finalize(s_single, W, H)
    for i, k in enumerate(KATS[:total]):
        got = int(res[i].view(np.uint32))
        assert got == k, "pair %d: 0x%08x, want 0x%08x" % (i, got, k)
    assert np.all(np.isfinite(res)) and res.min() > 0.88 and res.max() < 0.90, (res.min(), res.max())

    ctx.comm_init(ssim_amd.Context.comm_unique_id(), 1, 0)
    for world, strip_rows in ((8, 0), (3, 0), (5, 64)):
        ctx.set_tuning(strip_rows, 0)
        vec = ctx.alloc(8 * total).upload(np.zeros(total, np.float64))
        for first, last in sharding.split_batch(total, world):
            if last > first:
                shard = (ssim_amd.Params * (last - first))(*[params[i] for i in range(first, last)])
                ctx.enqueue_batch(shard, last - first, vec.ptr + 8 * first)
        ctx.comm_allreduce_sums(vec.ptr, total)
        ctx.synchronize()
        s_sharded = vec.download(np.float64, (total,))
        bad = np.flatnonzero(s_sharded.view(np.uint64) != s_single.view(np.uint64))
        assert bad.size == 0, "%d-way split: %d of %d sums differ from the single batch (first at %d: %r vs %r)" % (
            world, bad.size, total, bad[0], s_sharded[bad[0]], s_single[bad[0]])
        vec.free()
    ctx.set_tuning(0, 0)
    print("config4: %d pairs, generate %.2f s, single batch %.3f s (%.0f Mpix/s incl. launch + sync), shards 8/3/5 bit-identical"
          % (total, t_gen, t_batch, total * W * H / t_batch / 1e6))
    single.free()
    imgs.free()
    ctx.close()
    print("ok")


if __name__ == "__main__":
    main()
