#!/usr/bin/env python3
"""Single-rank self-test of the native RCCL exchange (rmgr_ssim_hip_comm_*).  Runs in its own process
without torch: torch wheels bundle their own librccl/HSA runtime, and a process that has both that
copy and the system ROCm loaded cannot initialise the bundled RCCL ("no ROCm-capable device").

usage: python3 tools/rccl_selftest.py     -> prints "ok" and exits 0
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ssim_amd  # noqa: E402
from ssim_amd import synth  # noqa: E402


def main():
    w, h, n = 200, 120, 4
    ctx = ssim_amd.Context(0)
    params = (ssim_amd.Params * n)()
    keep = []
    for i in range(n):
        a, b = synth.pair_numpy(w, h, 77 + i)
        da, db = ctx.upload(a), ctx.upload(b)
        keep += [da, db]
        params[i] = ssim_amd.make_params(w, h, da.ptr, 1, w, db.ptr, 1, w)
    sums = ctx.alloc(8 * n)
    ctx.enqueue_batch(params, n, sums.ptr)
    ctx.synchronize()
    before = sums.download(np.float64, (n,))
    ctx.comm_init(ssim_amd.Context.comm_unique_id(), 1, 0)
    ctx.comm_allreduce_sums(sums.ptr, n)
    ctx.synchronize()
    after = sums.download(np.float64, (n,))
    assert np.array_equal(before, after), (before, after)
    try:
        ctx.comm_init(ssim_amd.Context.comm_unique_id(), 1, 0)
        raise SystemExit("second comm_init must fail with EINVAL")
    except ssim_amd.SsimError as e:
        assert e.errno == 22
    ctx.close()
    print("ok")


if __name__ == "__main__":
    main()
