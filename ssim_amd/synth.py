"""Synthetic image pairs of SURVEY.md 8(d): integer-only, bit-reproducible on CPU and GPU.

    r = splitmix64(seed ^ ((y << 32) | x));  g = ((3x + 5y) >> 2) & 255
    A = (3g + (r & 255)) >> 2;  B = clamp(A + ((r >> 8) % 33) - 16, 0, 255)

Pair i of a batch uses seed 0x5EED + i.  The torch version runs on the device the batch lives on
(bench.py); the numpy version is the host twin.  Known answers: A[0,:4] = 45,59,51,6 for 0x5EED.
"""
import numpy as np

BASE_SEED = 0x5EED

_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_GOLD = 0x9E3779B97F4A7C15


def _s64(v):
    """Python int (uint64 bit pattern) -> the int64 with the same bits."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def pair_numpy(width, height, seed=BASE_SEED):
    y, x = np.meshgrid(np.arange(height, dtype=np.uint64), np.arange(width, dtype=np.uint64), indexing="ij")
    z = np.uint64(seed) ^ ((y << np.uint64(32)) | x)
    with np.errstate(over="ignore"):
        z = z + np.uint64(_GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
    r = z ^ (z >> np.uint64(31))
    g = ((np.uint64(3) * x + np.uint64(5) * y) >> np.uint64(2)) & np.uint64(255)
    a = ((np.uint64(3) * g + (r & np.uint64(255))) >> np.uint64(2)).astype(np.int64)
    n = ((r >> np.uint64(8)) % np.uint64(33)).astype(np.int64) - 16
    return a.astype(np.uint8), np.clip(a + n, 0, 255).astype(np.uint8)


def pair_torch(width, height, seed=BASE_SEED, device="cuda", rows_per_chunk=512):
    """Same generator with int64 tensor arithmetic (logical shifts emulated by masking)."""
    import torch

    def lsr(v, k):
        return (v >> k) & ((1 << (64 - k)) - 1)

    A = torch.empty((height, width), dtype=torch.uint8, device=device)
    B = torch.empty((height, width), dtype=torch.uint8, device=device)
    xs = torch.arange(width, dtype=torch.int64, device=device)[None, :]
    for y0 in range(0, height, rows_per_chunk):
        y1 = min(height, y0 + rows_per_chunk)
        ys = torch.arange(y0, y1, dtype=torch.int64, device=device)[:, None]
        z = ((ys << 32) | xs) ^ _s64(seed)
        z = z + _s64(_GOLD)
        z = (z ^ lsr(z, 30)) * _s64(_M1)
        z = (z ^ lsr(z, 27)) * _s64(_M2)
        r = z ^ lsr(z, 31)
        g = ((3 * xs + 5 * ys) >> 2) & 255
        a = (3 * g + (r & 255)) >> 2
        hi = lsr(r, 8)                       # 56-bit non-negative value
        n = torch.remainder(hi, 33) - 16
        A[y0:y1] = a.to(torch.uint8)
        B[y0:y1] = torch.clamp(a + n, 0, 255).to(torch.uint8)
    return A, B
