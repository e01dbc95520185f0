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


class PipelinedExchange(object):
    """The same exchange for a loop of steps, overlapped with the next step's kernels: two result vectors alternate,
    the all-reduce of step k is issued asynchronously (on the collective's own stream, ordered after the copy of
    step k's sums) and only awaited when its vector is needed again, two steps later -- or by drain().  On xGMI the
    8 B-per-pair all-reduce is latency-bound (tens of microseconds); this keeps it off the compute stream's critical path."""

    def __init__(self, sums_all, dist):
        self.sums_all, self.dist = sums_all, dist
        self.active = dist is not None and dist.is_initialized()
        self.bufs = [sums_all.clone(), sums_all.clone()] if self.active else []
        self.handles = [None, None]
        self.k = 0

    def step(self):
        """Call after this step's kernels were enqueued on the current stream.  Returns the tensor that will hold every
        rank's sums once the collective has completed (valid after drain(), or after a later wait on its handle)."""
        if not self.active:
            return self.sums_all
        i = self.k & 1
        self.k += 1
        if self.handles[i] is not None:
            self.handles[i].wait()            # the current stream waits for the collective of two steps ago
        self.bufs[i].copy_(self.sums_all)
        self.handles[i] = self.dist.all_reduce(self.bufs[i], async_op=True)
        return self.bufs[i]

    def drain(self):
        """Make the current stream wait for every outstanding collective; returns the most recent result vector."""
        last = self.sums_all
        for i in (self.k & 1, (self.k + 1) & 1):      # older first
            if self.handles[i] is not None:
                self.handles[i].wait()
                self.handles[i] = None
        if self.active and self.k:
            last = self.bufs[(self.k - 1) & 1]
        return last
