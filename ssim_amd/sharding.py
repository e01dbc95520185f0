"""Image-level sharding of a batch across the GPUs of one node.

Pairs are independent, so rank r owns a contiguous block of the batch and no image data crosses
xGMI.  The only exchange is the per-image fp64 sums: every rank writes its own slice of a
zero-initialised vector and one all-reduce gives every rank every sum.  Adding zeros is exact, so
the result is bit-identical for any GPU count.  This replaces the per-thread partials + final loop
of src/ssim.cpp:902-926, :1094-1100.

Two carriers for that all-reduce (bench.py --exchange):
  native  the product's own: rmgr_ssim_hip_comm_allreduce_sums behind the C ABI (RCCL over xGMI,
          include/rmgr/ssim-hip.h).  The launcher's process group is then only the control plane:
          it hands rank 0's 128-byte communicator id to the other ranks (handoff_unique_id) and lets
          the ranks agree on whether every communicator came up (all_agree).
  torch   torch.distributed's all_reduce (backend "nccl" = torch's bundled RCCL; gloo in the CPU tests).
"""


def shard_range(rank, world, pairs_per_rank):
    """[first, last) global pair indices owned by `rank` (weak scaling: fixed pairs per rank)."""
    if not (0 <= rank < world) or pairs_per_rank < 0:
        raise ValueError("bad shard (%d of %d, %d pairs)" % (rank, world, pairs_per_rank))
    return rank * pairs_per_rank, (rank + 1) * pairs_per_rank


def split_batch(total_pairs, world):
    """Strong-scaling split of `total_pairs` (BASELINE.json config 4: 1024 pairs over 8 GPUs):
    contiguous blocks, the first `total % world` ranks take one extra pair."""
    base, extra = divmod(total_pairs, world)
    out, first = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((first, first + n))
        first += n
    return out


def split_rows(height, cell_rows, world):
    """Row bands of ONE image for `world` ranks (rmgr_ssim_hip_enqueue_rows): contiguous [y0, y1) ranges that start on
    reduction-cell boundaries (cell_rows = 8, or 32 for images of >= 2048 rows: rmgr_ssim_hip_get_plan) and split the
    image's cell rows as evenly as they go; ranks beyond the number of cell rows get an empty band."""
    cells_y = (height + cell_rows - 1) // cell_rows
    cuts = [min(height, cell_rows * ((cells_y * r + world - 1) // world)) for r in range(world)] + [height]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def exchange_sums(sums_all, work, dist):
    """sums_all: float64 tensor [total pairs], zero outside this rank's slice.  Returns the tensor
    holding every rank's sums (work, all-reduced) -- or sums_all itself for a single process."""
    if dist is None or not dist.is_initialized():
        return sums_all
    work.copy_(sums_all)
    dist.all_reduce(work)
    return work



def handoff_unique_id(dist, make_id, rank):
    """The id hand-off of the native exchange: rank 0 calls `make_id()` (rmgr_ssim_hip_comm_get_unique_id: 128 bytes),
    every rank returns rank 0's bytes.  Carried by the already-initialised process group (any backend); a rank-0 failure
    reaches every rank as the same exception instead of leaving the others waiting."""
    box = [None]
    if rank == 0:
        try:
            uid = bytes(make_id())
            if len(uid) != 128:
                raise ValueError("communicator id of %d bytes" % len(uid))
            box[0] = ("id", uid)
        except Exception as e:           # noqa: BLE001 -- shipped to the other ranks, re-raised below on all of them
            box[0] = ("error", "%s: %s" % (type(e).__name__, e))
    dist.broadcast_object_list(box, src=0)
    kind, payload = box[0]
    if kind != "id":
        raise RuntimeError("rank 0 could not create the communicator id: %s" % payload)
    return payload


def all_agree(dist, ok, device=None):
    """True on every rank iff `ok` is true on every rank (one MIN all-reduce of a flag on the process group)."""
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


def exchange_sums_native(ctx, sums_all, work):
    """The native carrier: `work` <- `sums_all` (zero outside this rank's slice), then the in-place RCCL all-reduce of
    `work` through the C ABI on the context's stream.  Both tensors are float64 device tensors of the whole batch."""
    work.copy_(sums_all)
    ctx.comm_allreduce_sums(work.data_ptr(), work.numel())
    return work


def compare_digests(dist, digest, describe=""):
    """Every rank contributes the digest of the result vector IT holds after the exchange (and a line describing itself);
    returns (ok, lines) on every rank: ok iff all digests equal rank 0's, lines = one "rank r: <describe> digest <d>" per rank.
    The all-reduce gives every rank every sum, so the digests must agree; a rank that disagrees computed something else
    (wrong device, missed exchange, a communicator that did not span the launch) and the run must not produce a number."""
    if dist is None or not dist.is_initialized():
        return True, ["rank 0: %s digest %s" % (describe, digest)]
    world = dist.get_world_size()
    box = [None] * world
    dist.all_gather_object(box, (dist.get_rank(), describe, digest))
    box.sort()
    lines = ["rank %d: %s digest %s%s" % (r, d, g, "" if g == box[0][2] else "   <-- differs from rank 0") for r, d, g in box]
    return all(g == box[0][2] for _, _, g in box), lines
