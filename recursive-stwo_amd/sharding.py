"""Multi-GPU host logic: one process per GPU, proofs sharded by contiguous index range, ONE exchange per
batch (all-gather of the per-rank accept bitmaps over RCCL/xGMI; `torch.distributed` backend "nccl" on GPUs,
"gloo" in the CPU tests).  There is no data-path collective: proofs are independent units."""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous [lo, hi) of proof indices owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def bitmap_words(n: int) -> int:
    return (n + 31) // 32


def pack_bitmap(accept: np.ndarray) -> np.ndarray:
    """accept bytes (0/1) -> little-endian bitmap of uint32 words (same layout as rsv_accept_bitmap_dev)."""
    n = len(accept)
    bits = np.zeros(bitmap_words(n) * 32, np.uint8)
    bits[:n] = accept != 0
    return np.packbits(bits, bitorder="little").view(np.uint32)


def unpack_bitmap(words: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), bitorder="little")[:n]


def gather_accept_bitmap(local_bitmap, n_total: int, rank: int, world: int, dist, torch):
    """All-gather the per-rank bitmaps and reassemble the global accept vector (uint8[n_total]) on every rank.
    `local_bitmap` is an int32 tensor (device of the process group's backend) holding this rank's shard bits.
    Ranks own shards of different sizes (shard_range), so every rank pads to the largest shard's word count."""
    max_words = bitmap_words(shard_range(n_total, 0, world)[1] - shard_range(n_total, 0, world)[0])
    padded = torch.zeros(max_words, dtype=torch.int32, device=local_bitmap.device)
    padded[: local_bitmap.numel()] = local_bitmap
    gathered = torch.zeros(world * max_words, dtype=torch.int32, device=local_bitmap.device)
    if world > 1:
        dist.all_gather_into_tensor(gathered, padded)
    else:
        gathered.copy_(padded)
    g = gathered.cpu().numpy().view(np.uint32).reshape(world, max_words)
    out = np.zeros(n_total, np.uint8)
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        out[lo:hi] = unpack_bitmap(g[r], hi - lo)
    return out
