"""Multi-GPU sharding of the overlapping scan: one process per GPU, torch.distributed (RCCL over xGMI).

The haystack is partitioned into contiguous shards, one per rank; each rank scans its shard with
`acgpu_find_overlapping_shard` (max_pattern_len-1 bytes of warm-up left of the seam, matches owned by
`end` in (shard_begin, shard_end]).  There is no data-path collective: the only exchange is the gather
of the (tiny) match-record lists to rank 0, in rank order == global `end` order.
"""
import numpy as np

from .api import MATCH_DTYPE


def plan_shards(span_start, span_end, world):
    """Contiguous shard bounds [(begin, end)] * world tiling [span_start, span_end); sizes differ by <= 1
    64-byte unit so interior seams stay 64-byte aligned relative to span_start."""
    n = max(0, span_end - span_start)
    units = (n + 63) // 64
    bounds = []
    for r in range(world + 1):
        b = span_start + min(n, ((units * r) // world) * 64)
        bounds.append(b)
    bounds[-1] = span_end if span_end >= span_start else span_start
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def gather_matches(local, dst=0, group=None, device=None, offset=0):
    """Gather per-rank match records (numpy MATCH_DTYPE arrays, already in stream order) to rank `dst`.

    Two collectives: all_gather of the counts, then a padded gather of the records viewed as int64 words.
    `offset` is added to every start/end (local shard coordinates -> global haystack offsets).
    Returns the concatenated array on `dst`, None elsewhere.  Works on gloo (CPU tensors) and nccl/RCCL
    (pass device='cuda')."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(local, np.ndarray):
        words = torch.from_numpy(np.ascontiguousarray(local).view(np.int64).reshape(-1, 3).copy()).to(dev)
    else:  # uint8 device tensor holding n*24 bytes
        words = local.view(torch.int64).reshape(-1, 3)
    if offset:
        words = words.clone()
        words[:, 1:] += int(offset)
    n_local = torch.tensor([words.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, 3), dtype=torch.int64, device=dev)
    padded[:words.shape[0]] = words
    if rank == dst:
        bufs = [torch.zeros((cap, 3), dtype=torch.int64, device=dev) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst, group=group)
        parts = [bufs[r][:counts[r]].cpu().numpy().reshape(-1).view(MATCH_DTYPE) for r in range(world)]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)
    dist.gather(padded, None, dst=dst, group=group)
    return None
