"""Image-level sharding of a batch across the GPUs of one node.

Pairs are independent, so rank r owns a contiguous block of the batch and no image data crosses
xGMI.  The only exchange is the per-image fp64 sums: every rank writes its own slice of a
zero-initialised vector and one all-reduce (RCCL over xGMI with backend "nccl"; gloo in the CPU
tests) gives every rank every sum.  Adding zeros is exact, so the result is bit-identical for any
GPU count.  This replaces the per-thread partials + final loop of src/ssim.cpp:902-926, :1094-1100.
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


def exchange_sums(sums_all, work, dist):
    """sums_all: float64 tensor [total pairs], zero outside this rank's slice.  Returns the tensor
    holding every rank's sums (work, all-reduced) -- or sums_all itself for a single process."""
    if dist is None or not dist.is_initialized():
        return sums_all
    work.copy_(sums_all)
    dist.all_reduce(work)
    return work

