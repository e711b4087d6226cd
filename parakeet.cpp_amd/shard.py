"""Utterance-batch data parallelism: clips are independent, so the only 'distribution' is a static
shard of the clip list per rank plus a gather of the (tiny) token-id results.  Used by the multi-GPU drivers; the
collective backend is torch.distributed ('nccl' = RCCL over xGMI on MI355X, 'gloo' in the CPU tests)."""
from typing import List, Sequence


def shard_indices(n_items: int, rank: int, world: int, batch: int = 64) -> List[int]:
    """Round-robin over BATCHES (rank r takes batches r, r+world, ...), so every rank runs full batches and the
    tail batch lands on one rank.  Returns the item indices owned by `rank`, in processing order."""
    out = []
    n_batches = (n_items + batch - 1) // batch
    for b in range(rank, n_batches, world):
        out.extend(range(b * batch, min(n_items, (b + 1) * batch)))
    return out


def gather_results(local: Sequence, local_idx: Sequence[int], n_items: int, world: int, dist=None) -> list:
    """All-gather per-rank (index, result) lists and reassemble them in the original clip order on every rank."""
    if world == 1 or dist is None:
        merged = [None] * n_items
        for i, r in zip(local_idx, local):
            merged[i] = r
        return merged
    bucket = [None] * world
    dist.all_gather_object(bucket, (list(local_idx), list(local)))
    merged = [None] * n_items
    for idx, res in bucket:
        for i, r in zip(idx, res):
            merged[i] = r
    return merged
