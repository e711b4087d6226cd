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


def shard_by_audio(lengths: Sequence[int], rank: int, world: int) -> List[int]:
    """Mixed-length clips (round 4): the partition pk_group_transcribe_pcm uses inside one process, for the one-process-per-GPU deployment --
    clips sorted by length, longest first (stable), each dealt to the rank with the least audio so far (lowest rank on ties; equal lengths:
    rank r takes clips r, r + world, ...).  Every rank then packs ITS clips into ragged batches (pk_transcribe_pcm).  Deterministic, the same
    on every rank, no communication.  Returns the clip indices owned by `rank`, longest first."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    load = [0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda q: load[q])
        load[r] += int(lengths[i])
        if r == rank:
            mine.append(i)
    return mine


def shard_sessions(n_sessions: int, rank: int, world: int, group: int = 16) -> List[int]:
    """Streaming sessions (BASELINE configs[4]: N concurrent streams per GPU on the GPUs of a node): a session lives on ONE GPU for its whole
    life (its K / V / conv caches and decoder state are resident there), so the partition is static -- sessions are dealt in lock-step GROUPS of
    `group` (one pk_stream of `group` streams each: the batch dimension of every kernel of a chunk), group g to rank g % world.  A rank then
    advances its groups one after the other per 160 ms tick.  No data-path collective; the same on every rank.  Returns the session ids of `rank`."""
    out = []
    n_groups = (n_sessions + group - 1) // group
    for g in range(rank, n_groups, world):
        out.extend(range(g * group, min(n_sessions, (g + 1) * group)))
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


def gather_token_matrix(local_ids, local_lens, local_idx: Sequence[int], n_items: int, world: int, dist=None, device=None):
    """Collective result exchange of the sharded run: every rank contributes a fixed-stride int32 matrix
    [n_local_padded][1 + max_tokens] (column 0 = clip index or -1 for padding rows, column 1 = length, then the ids) and ONE
    all_gather_into_tensor ('nccl' = RCCL over xGMI on MI355X; 'gloo' in the CPU tests) brings them to every rank; rows are put
    back in clip order.  Returns (ids [n_items][max_tokens], lens [n_items]).  ~2 MB for 8192 clips: latency-bound, off the data path."""
    import numpy as np
    import torch
    local_ids = np.asarray(local_ids, np.int32)
    mt = local_ids.shape[1] if local_ids.ndim == 2 else 0
    n_local = len(local_idx)
    per_rank = (n_items + world - 1) // world if world > 1 else n_items
    # ranks own different numbers of clips (round-robin over batches): pad to the largest shard
    if world > 1 and dist is not None:
        t = torch.tensor([n_local], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = int(t.item())
    mat = np.full((per_rank, 2 + mt), -1, np.int32)
    mat[:n_local, 0] = np.asarray(local_idx, np.int32)
    mat[:n_local, 1] = np.asarray(local_lens, np.int32)
    if mt:
        mat[:n_local, 2:] = local_ids
    if world > 1 and dist is not None:
        mine = torch.from_numpy(mat).to(device) if device is not None else torch.from_numpy(mat)
        out = torch.empty((world * per_rank, 2 + mt), dtype=torch.int32, device=mine.device)
        dist.all_gather_into_tensor(out, mine)
        mat = out.cpu().numpy()
    ids = np.zeros((n_items, mt), np.int32)
    lens = np.zeros(n_items, np.int32)
    seen = 0
    for row in mat:
        if row[0] >= 0:
            ids[row[0]] = row[2:]
            lens[row[0]] = row[1]
            seen += 1
    if seen != n_items:
        raise RuntimeError(f"sharded gather: {seen} of {n_items} clips arrived")
    return ids, lens


def broadcast_file(path: str, rank: int, world: int, dist=None, device=None):
    """Weight distribution of the sharded run: rank 0 reads the safetensors file ONCE, its bytes travel in one
    broadcast ('nccl' = RCCL over xGMI: 7 links x ~153 GB/s make 459 MB a matter of milliseconds; 'gloo' in the CPU tests) and every
    rank builds its replica from memory (pk_model_load_buffer).  Returns the file image as a uint8 numpy array."""
    import numpy as np
    import torch
    if world == 1 or dist is None:
        return np.fromfile(path, np.uint8)
    n = torch.zeros(1, dtype=torch.int64, device=device)
    img = None
    if rank == 0:
        img = torch.from_numpy(np.fromfile(path, np.uint8))
        n[0] = img.numel()
    dist.broadcast(n, src=0)
    if rank == 0:
        buf = img.to(device) if device is not None else img
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=0)
    return buf.cpu().numpy()
