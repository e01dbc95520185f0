"""CPU, 2 processes over gloo: the N>1 path of bench.py (shard by image, one all-reduce of the
per-image fp64 sums, library-side finalize).  The per-image sums come from the oracle here -- it is
the checker standing in for the device kernel, which needs the GPU; what is under test is that the
sharding/exchange plumbing hands every rank the bit-identical vector a single process gets."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

W, H, PAIRS_PER_RANK = 96, 70, 3


def image_sum(i):
    import oracle
    a, b = oracle.synth_pair(W, H, 0x5EED + i)
    return oracle.ssim_f32(a, b)[1]


def worker(rank, world, port, out_dir, strong_total=0):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import ssim_amd
    from ssim_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if strong_total:      # bench.py --scaling strong: a fixed batch split over the ranks (uneven shares allowed)
        first, last = sharding.split_batch(strong_total, world)[rank]
        total = strong_total
    else:                 # --scaling weak: a fixed share per rank
        first, last = sharding.shard_range(rank, world, PAIRS_PER_RANK)
        total = world * PAIRS_PER_RANK
    sums_all = torch.zeros(total, dtype=torch.float64)
    work = torch.zeros_like(sums_all)
    for i in range(first, last):
        sums_all[i] = image_sum(i)
    full = sharding.exchange_sums(sums_all, work, dist)
    res = ssim_amd.finalize(full.numpy(), W, H)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), res)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_get_identical_complete_results(tmp_path):
    import torch.multiprocessing as mp     # torch stays out of processes that only collect this file
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import ssim_amd
    single = ssim_amd.finalize(np.array([image_sum(i) for i in range(world * PAIRS_PER_RANK)]), W, H)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got.view(np.uint32), single.view(np.uint32)), r


def test_strong_scaling_split_with_uneven_shares(tmp_path):
    """BASELINE.json configs[3] shape (a fixed batch over N ranks) with a batch that does not divide evenly."""
    import torch.multiprocessing as mp
    world, total = 2, 5
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(world, port, str(tmp_path), total), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    import ssim_amd
    single = ssim_amd.finalize(np.array([image_sum(i) for i in range(total)]), W, H)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got.view(np.uint32), single.view(np.uint32)), r


def handoff_worker(rank, world, port, out_dir, fail):
    """The control plane of bench.py --exchange native on 2 CPU ranks (gloo): the id hand-off and the agreement step.  The id
    itself is 128 random bytes here -- rmgr_ssim_hip_comm_get_unique_id needs a GPU -- what is under test is that every
    rank ends up with RANK 0's bytes, that a failure on rank 0 reaches every rank as an exception (nobody is left waiting),
    and that one rank's failed communicator makes ALL ranks fall back."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from ssim_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def make_id():
        calls.append(rank)
        if fail == "id":
            raise OSError(110, "no librccl here")
        return bytes([17 * (rank + 1)] * 64) + os.urandom(64)

    try:
        uid = sharding.handoff_unique_id(dist, make_id, rank)
        outcome = "id:" + uid.hex()
    except RuntimeError as e:
        outcome = "error:" + str(e)
    assert calls == ([0] if rank == 0 else []), calls           # only rank 0 creates the id
    ok_here = not (fail == "init" and rank == 1)                # one rank's comm_init "timed out"
    agreed = sharding.all_agree(dist, ok_here and outcome.startswith("id:"))
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%s\n%d\n" % (outcome, int(agreed)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", ["", "id", "init"])
def test_native_exchange_id_handoff_and_agreement(tmp_path, fail):
    import torch.multiprocessing as mp
    world = 2
    port = 33500 + (os.getpid() % 2000) + {"": 0, "id": 1, "init": 2}[fail]
    mp.spawn(handoff_worker, args=(world, port, str(tmp_path), fail), nprocs=world, join=True)
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().split("\n") for r in range(world)]
    assert got[0][0] == got[1][0], got                           # the same bytes -- or the same error -- on every rank
    if fail == "id":
        assert got[0][0].startswith("error:") and "no librccl here" in got[0][0]
    else:
        assert got[0][0].startswith("id:") and got[0][0][3:3 + 128] == "11" * 64       # rank 0's id, not rank 1's
    assert got[0][1] == got[1][1] == ("1" if fail == "" else "0")       # native only if EVERY rank is up


def digest_worker(rank, world, port, out_dir, corrupt):
    """bench.py's self-diagnosis before timing: every rank's digest of the vector it holds after the exchange, compared on all
    ranks.  corrupt: rank 1 holds a vector that differs in one bit (a rank that missed the exchange / ran on the wrong data)."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from ssim_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = sharding.shard_range(rank, world, PAIRS_PER_RANK)
    sums_all = torch.zeros(world * PAIRS_PER_RANK, dtype=torch.float64)
    for i in range(first, last):
        sums_all[i] = image_sum(i)
    full = sharding.exchange_sums(sums_all, torch.zeros_like(sums_all), dist).numpy().copy()
    if corrupt and rank == 1:
        full.view(np.uint64)[2] ^= 1
    digest = "%016x" % int(np.bitwise_xor.reduce(full.view(np.uint64) * (np.arange(full.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1))))
    ok, lines = sharding.compare_digests(dist, digest, "device %d carrier torch ranks_seen %d" % (rank, world))
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("%d\n%s\n" % (int(ok), "\n".join(lines)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_ranks_compare_result_digests_before_timing(tmp_path, corrupt):
    """a multi-rank bench run must diagnose itself -- the same verdict and the same per-rank lines on every rank,
    agreement when the exchange worked, a flagged rank (and hence a non-zero exit in bench.py) when one rank holds other bits."""
    import torch.multiprocessing as mp
    world = 2
    port = 35500 + (os.getpid() % 2000) + int(corrupt)
    mp.spawn(digest_worker, args=(world, port, str(tmp_path), corrupt), nprocs=world, join=True)
    got = [open(os.path.join(str(tmp_path), "rank%d.txt" % r)).read().strip().split("\n") for r in range(world)]
    assert got[0] == got[1]                                     # every rank sees the same verdict and the same lines
    assert got[0][0] == ("0" if corrupt else "1")
    assert len(got[0]) == 1 + world and got[0][1].startswith("rank 0: device 0 carrier torch ranks_seen 2 digest ")
    assert ("differs from rank 0" in got[0][2]) == corrupt


def test_bench_builds_its_own_rank_launcher():
    """`python bench.py --gpus N` must not depend on an outside launcher: it starts
    torch.distributed.run itself, as a child process, on 127.0.0.1."""
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--print-launch", "--gpus", "4", "--steps", "3", "--scaling", "strong", "--workload", "1080p"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    cmd = r.stdout.split()
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--print-launch" not in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index(bench) + 1:] == ["--gpus", "4", "--steps", "3", "--scaling", "strong", "--workload", "1080p"]


def test_bench_shard_tables_are_what_design_section_6_says():
    """The sharding the timed run uses (`bench.py --print-shards`, the same function the ranks call): BASELINE.json configs[3]
    -- 1024 x 1080p over 8 GPUs, strong scaling -- is eight contiguous blocks of 128 pairs; the default workload is weak
    scaling, 32 pairs of 4096^2 per GPU; a batch that does not divide gives the first ranks one extra pair.  Plus the
    2-rank launcher line of the same configuration."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")

    def run(*argv):
        r = subprocess.run([sys.executable, bench] + list(argv), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        return r.stdout

    t = json.loads(run("--gpus", "8", "--scaling", "strong", "--workload", "1080p", "--print-shards"))
    assert t == {"scaling": "strong", "total": 1024, "shards": [[128 * r, 128 * (r + 1)] for r in range(8)]}
    t = json.loads(run("--gpus", "8", "--print-shards"))
    assert t == {"scaling": "weak", "total": 256, "shards": [[32 * r, 32 * (r + 1)] for r in range(8)]}
    t = json.loads(run("--gpus", "3", "--scaling", "strong", "--workload", "1080p", "--pairs", "10", "--print-shards"))
    assert t["shards"] == [[0, 4], [4, 7], [7, 10]]
    for n in (1, 2, 4, 8):                                  # every pair has exactly one owner, in order, for every N the driver runs
        t = json.loads(run("--gpus", str(n), "--scaling", "strong", "--workload", "1080p", "--print-shards"))
        flat = [i for a, b in t["shards"] for i in range(a, b)]
        assert flat == list(range(1024)) and len(t["shards"]) == n
    cmd = run("--gpus", "2", "--scaling", "strong", "--workload", "1080p", "--print-launch").split()
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index(bench) + 1:] == ["--gpus", "2", "--scaling", "strong", "--workload", "1080p"]


def test_bench_two_ranks_fail_only_at_the_missing_device():
    """On a box without GPUs the self-launched 2-rank run must get as far as every rank looking for its device."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("checks the no-device behaviour")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    # every rank that gets to run says so; torchrun may terminate the slower one as soon as the first has failed
    assert 1 <= r.stderr.count("no HIP device visible") <= 2, r.stderr[-1500:]
    assert "starting 2 ranks" in r.stderr and "WORLD_SIZE" not in r.stderr
    assert "metric" not in r.stdout


def test_shard_helpers():
    from ssim_amd import sharding
    assert sharding.shard_range(3, 8, 128) == (384, 512)
    assert sharding.split_batch(1024, 8) == [(128 * r, 128 * (r + 1)) for r in range(8)]
    parts = sharding.split_batch(10, 4)
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        sharding.shard_range(2, 2, 1)
    # row bands of one image: cell-aligned, contiguous, covering, as even as the cell rows allow
    assert sharding.split_rows(8192, 32, 8) == [(1024 * r, 1024 * (r + 1)) for r in range(8)]
    assert sharding.split_rows(1080, 8, 4) == [(0, 272), (272, 544), (544, 816), (816, 1080)]
    assert sharding.split_rows(20, 8, 5) == [(0, 8), (8, 16), (16, 16), (16, 20), (20, 20)]       # more ranks than cell rows: some bands are empty
    for h, cr, n in ((4100, 32, 3), (77, 8, 2), (1, 8, 4)):
        bands = sharding.split_rows(h, cr, n)
        assert bands[0][0] == 0 and bands[-1][1] == h and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
        assert all(y0 % cr == 0 for y0, _ in bands if y0 < h)


def test_synthetic_generators_agree():
    """ssim_amd.synth (numpy and torch twins, used by bench.py) == the oracle's C generator."""
    sys.path.insert(0, ROOT)
    import oracle
    from ssim_amd import synth
    for seed in (0x5EED, 0x5EEE, 0x5EED + 255):
        a, b = oracle.synth_pair(301, 77, seed)
        na, nb = synth.pair_numpy(301, 77, seed)
        ta, tb = synth.pair_torch(301, 77, seed, device="cpu", rows_per_chunk=32)
        assert np.array_equal(a, na) and np.array_equal(b, nb)
        assert np.array_equal(a, ta.numpy()) and np.array_equal(b, tb.numpy())
