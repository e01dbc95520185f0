#!/usr/bin/env python3
"""Self-tests of the native RCCL exchange (rmgr_ssim_hip_comm_*) on ONE GPU, each in its own process.

    python3 tools/rccl_selftest.py [single|absent-peer|two-ranks|shards] [--with-torch]      -> "RESULT ok" on stderr, exits 0

single       a 1-rank communicator leaves the per-image sums bit-identical (what N ranks add is zeros); RCCL itself counts
             one rank; a second comm_init is EINVAL; destroy, then a fresh communicator works again.
absent-peer  rank 1 of 2 whose rank 0 never shows up: comm_init must come back with ETIMEDOUT within the deadline
             ($RMGR_SSIM_HIP_COMM_TIMEOUT_S, set short here) instead of hanging, the context must still compute, and a
             1-rank communicator must still initialise afterwards.  The bounded-failure contract of the reference
             (a failed worker -> ECHILD, src/ssim.cpp:1094-1097) for the multi-GPU exchange.
two-ranks    two processes (this one starts rank 1 with the id on its command line), both on device 0, join ONE communicator:
             the real N > 1 rendezvous as far as a 1-GPU box can take it.  RCCL refuses two ranks on one device (EINVAL) --
             after the ranks have met; both must report the same verdict inside the deadline.
shards       one rank's share of BASELINE.json configs[3] (128 x 1080p) cut into 8 / 3 / 5 emulated shards, each enqueued
             into its slice of a zeroed vector, then rmgr_ssim_hip_comm_allreduce_sums over the whole vector on a 1-rank
             communicator == the single batch, bit for bit.
--with-torch imports torch FIRST: the process then carries torch's bundled HIP runtime and RCCL, and the library must bind
             to THAT RCCL (how bench.py --exchange native runs).  Without it: the system ROCm's RCCL, no torch anywhere.

Every stage is announced on stderr with a timestamp BEFORE it starts, and RCCL's own INIT / BOOTSTRAP / NET log is
switched on, so that a run that is killed from outside says where it was.  A watchdog dumps all Python stacks and exits
non-zero after $RCCL_SELFTEST_LIMIT_S (default 50 s): the process never needs to be killed by pattern or re-exec'd.
"""
import faulthandler
import os
import sys
import time

T0 = time.time()


def stage(msg):
    sys.stderr.write("[rccl_selftest %7.3f s] %s\n" % (time.time() - T0, msg))
    sys.stderr.flush()


os.environ.setdefault("NCCL_DEBUG", "INFO")
os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,BOOTSTRAP,NET")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what this driver supports (bench.py sets the same)
os.environ.setdefault("RMGR_SSIM_HIP_COMM_TIMEOUT_S", "20")
os.environ.setdefault("RMGR_SSIM_HIP_COMM_DEBUG", "1")
LIMIT = float(os.environ.get("RCCL_SELFTEST_LIMIT_S", "50"))
faulthandler.enable()
faulthandler.dump_traceback_later(LIMIT, exit=True)

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
WHAT = ARGS[0] if ARGS else "single"
UID_HEX = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--rank1=")] or [""])[0]
RANK = 1 if UID_HEX else 0
if "--with-torch" in sys.argv:
    # the first import of torch on a fresh box pages in gigabytes (1-2 minutes have been seen): that is the image, not
    # RCCL, so it gets its own generous limit and the RCCL part's watchdog is armed afterwards
    faulthandler.cancel_dump_traceback_later()
    faulthandler.dump_traceback_later(float(os.environ.get("RCCL_SELFTEST_TORCH_IMPORT_LIMIT_S", "300")), exit=True)
    stage("import torch (bundled HIP runtime + RCCL)")
    import torch  # noqa: F401
    stage("torch %s imported" % torch.__version__)
    faulthandler.cancel_dump_traceback_later()
    faulthandler.dump_traceback_later(LIMIT, exit=True)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
stage("import ssim_amd")
import ssim_amd  # noqa: E402
from ssim_amd import sharding, synth  # noqa: E402

ETIMEDOUT, EINVAL = 110, 22


def make_batch(ctx, w, h, n, keep):
    params = (ssim_amd.Params * n)()
    imgs = ctx.alloc(2 * w * h * n)
    keep.append(imgs)
    for i in range(n):
        a = imgs.ptr + 2 * w * h * i
        ctx.synth_pair(a, w, a + w * h, w, w, h, synth.BASE_SEED + i)
        params[i] = ssim_amd.make_params(w, h, a, 1, w, a + w * h, 1, w)
    return params


def new_comm(ctx, ranks=1, rank=0):
    stage("comm_unique_id (dlopen librccl, bootstrap root)")
    uid = ssim_amd.Context.comm_unique_id()
    stage("rccl: " + ssim_amd.Context.comm_describe())
    stage("comm_init(%d ranks, rank %d)" % (ranks, rank))
    t = time.time()
    ctx.comm_init(uid, ranks, rank)
    stage("comm_init done in %.2f s; RCCL counts %d rank(s)" % (time.time() - t, ctx.comm_rank_count()))


def single(ctx):
    keep = []
    w, h, n = 200, 120, 4
    params = make_batch(ctx, w, h, n, keep)
    sums = ctx.alloc(8 * n)
    stage("enqueue + synchronize (no communicator yet)")
    ctx.enqueue_batch(params, n, sums.ptr)
    ctx.synchronize()
    before = sums.download(np.float64, (n,))
    assert ctx.comm_rank_count() == 0
    new_comm(ctx)
    assert ctx.comm_rank_count() == 1
    stage("comm_allreduce_sums x 3 + synchronize")
    for _ in range(3):
        ctx.comm_allreduce_sums(sums.ptr, n)
    ctx.synchronize()
    after = sums.download(np.float64, (n,))
    assert np.array_equal(before.view(np.uint64), after.view(np.uint64)), (before, after)
    stage("second comm_init must be EINVAL")
    try:
        ctx.comm_init(ssim_amd.Context.comm_unique_id(), 1, 0)
        raise SystemExit("second comm_init must fail with EINVAL")
    except ssim_amd.SsimError as e:
        assert e.errno == EINVAL, e
    # cost of one all-reduce of this size on the stream (latency-bound: 8 B per pair)
    ctx.synchronize()
    t = time.time()
    reps = 200
    for _ in range(reps):
        ctx.comm_allreduce_sums(sums.ptr, n)
    t_host = (time.time() - t) / reps
    ctx.synchronize()
    t_all = (time.time() - t) / reps
    stage("all-reduce of %d doubles: %.1f us of host time per call, %.1f us per call incl. the stream" % (n, t_host * 1e6, t_all * 1e6))
    stage("comm_destroy")
    ctx.comm_destroy()
    assert ctx.comm_rank_count() == 0
    new_comm(ctx)                      # a context can be given a communicator again
    ctx.comm_allreduce_sums(sums.ptr, n)
    ctx.synchronize()
    assert np.array_equal(before.view(np.uint64), sums.download(np.float64, (n,)).view(np.uint64))
    ctx.comm_destroy()


def absent_peer(ctx):
    keep = []
    w, h, n = 200, 120, 2
    params = make_batch(ctx, w, h, n, keep)
    sums = ctx.alloc(8 * n)
    ctx.enqueue_batch(params, n, sums.ptr)
    ctx.synchronize()
    before = sums.download(np.float64, (n,))
    limit = float(os.environ["RMGR_SSIM_HIP_COMM_TIMEOUT_S"])
    # the short deadline is for the call under test only: the first load of the 573 MB library on a cold box alone can take 5 s
    os.environ["RMGR_SSIM_HIP_COMM_TIMEOUT_S"] = "30"
    stage("comm_unique_id")
    uid = ssim_amd.Context.comm_unique_id()
    os.environ["RMGR_SSIM_HIP_COMM_TIMEOUT_S"] = str(limit)
    stage("comm_init as rank 1 of 2 -- rank 0 never arrives; deadline %.0f s" % limit)
    t = time.time()
    try:
        ctx.comm_init(uid, 2, 1)
        raise SystemExit("comm_init with an absent peer returned success")
    except ssim_amd.SsimError as e:
        dt = time.time() - t
        stage("comm_init -> errno %d after %.2f s" % (e.errno, dt))
        assert e.errno == ETIMEDOUT, e
        assert limit - 0.5 <= dt <= limit + 2.0, dt          # AT the deadline: the abort of the half-built communicator runs in the background
    os.environ["RMGR_SSIM_HIP_COMM_TIMEOUT_S"] = "30"
    assert ctx.comm_rank_count() == 0
    stage("the context still computes")
    ctx.enqueue_batch(params, n, sums.ptr)
    ctx.synchronize()
    assert np.array_equal(before.view(np.uint64), sums.download(np.float64, (n,)).view(np.uint64))
    new_comm(ctx)                      # and can still be given a working communicator
    ctx.comm_allreduce_sums(sums.ptr, n)
    ctx.synchronize()
    assert np.array_equal(before.view(np.uint64), sums.download(np.float64, (n,)).view(np.uint64))
    ctx.comm_destroy()


def shards(ctx):
    keep = []
    w, h, total = 1920, 1080, 128
    stage("generate %d x %dx%d pairs" % (total, w, h))
    params = make_batch(ctx, w, h, total, keep)
    single_v = ctx.alloc(8 * total)
    ctx.enqueue_batch(params, total, single_v.ptr)
    ctx.synchronize()
    s_single = single_v.download(np.float64, (total,))
    res = ssim_amd.finalize(s_single, w, h)
    for i, k in enumerate((0x3f64bb1f, 0x3f64bbf6, 0x3f64bb30)):
        assert int(res[i].view(np.uint32)) == k, (i, hex(int(res[i].view(np.uint32))))
    new_comm(ctx)
    for world, strip_rows in ((8, 0), (3, 0), (5, 64)):
        stage("%d emulated shards + all-reduce" % world)
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
        assert bad.size == 0, "%d-way split: %d of %d sums differ from the single batch" % (world, bad.size, total)
        vec.free()
    ctx.set_tuning(0, 0)
    ctx.comm_destroy()


def two_ranks(ctx):
    """Two processes, both on device 0 (all a test box has): does this RCCL take two ranks of one communicator on ONE GPU?
    Either answer is fine -- what is checked is that both ranks come back with the SAME verdict inside the deadline: a 2-rank
    communicator that RCCL itself counts as 2 ranks and whose all-reduce adds the two ranks' vectors, or an errno on both."""
    import subprocess
    n = 8
    mine = np.arange(1, n + 1, dtype=np.float64) * (1.0 if RANK == 0 else 0.5)
    if RANK == 0:
        stage("rank 0: comm_unique_id")
        uid = ssim_amd.Context.comm_unique_id()
        stage("rank 0: starting rank 1")
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "two-ranks", "--rank1=" + uid.hex()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    else:
        uid = bytes.fromhex(UID_HEX)
    sums = ctx.alloc(8 * n).upload(mine)
    verdict = "?"
    try:
        stage("rank %d: comm_init(2 ranks)" % RANK)
        t = time.time()
        ctx.comm_init(uid, 2, RANK)
        stage("rank %d: comm_init done in %.2f s; RCCL counts %d rank(s)" % (RANK, time.time() - t, ctx.comm_rank_count()))
        assert ctx.comm_rank_count() == 2
        ctx.comm_allreduce_sums(sums.ptr, n)
        ctx.synchronize()
        got = sums.download(np.float64, (n,))
        assert np.array_equal(got, np.arange(1, n + 1, dtype=np.float64) * 1.5), got
        ctx.comm_destroy()
        verdict = "2-rank communicator on one GPU works"
    except ssim_amd.SsimError as e:
        stage("rank %d: errno %d after %.2f s" % (RANK, e.errno, time.time() - t))
        verdict = "errno %d" % e.errno
    stage("rank %d: VERDICT %s" % (RANK, verdict))
    if RANK == 0:
        out, err = child.communicate(timeout=LIMIT)
        sys.stderr.write("".join("    | " + l + "\n" for l in err.splitlines() if "rccl_selftest" in l or "rmgr-ssim comm" in l or "WARN" in l))
        theirs = [l.split("VERDICT ", 1)[1] for l in err.splitlines() if "VERDICT " in l]
        assert child.returncode == 0 and theirs == [verdict], (child.returncode, theirs, verdict)


def main():
    stage("create context on device 0")
    ctx = ssim_amd.Context(0)
    stage(ctx.describe())
    {"single": single, "absent-peer": absent_peer, "shards": shards, "two-ranks": two_ranks}[WHAT](ctx)
    stage("close")
    ctx.close()
    faulthandler.cancel_dump_traceback_later()
    stage("RESULT ok")                  # the verdict goes to stderr like the stage markers: RCCL's INFO log shares stdout
    sys.stdout.flush()
    os.write(1, b"\nok\n")


if __name__ == "__main__":
    main()
