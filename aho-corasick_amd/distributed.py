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


class MatchGatherer:
    """Per-step gather of the ranks' match records with ONE collective and persistent buffers.

    Every rank contributes a fixed-size int64 payload [n, words of up to `cap` records]; `all_gather_into_tensor`
    moves the payloads (a few hundred KB over xGMI), rank `dst` copies the gathered block to the host once and
    slices it by the counts in the headers, adding each rank's coordinate offset there.  If any rank has more than
    `cap` records every rank sees it in the headers and all of them take the exact two-collective path
    (`gather_matches`) for that step, then the capacity is doubled."""

    def __init__(self, cap=8192, dst=0, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.group, self.dst = group, dst
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        self._alloc(cap)

    def _alloc(self, cap):
        t = self.torch
        self.cap = int(cap)
        self.payload = t.zeros(1 + 3 * self.cap, dtype=t.int64, device=self.dev)
        self.gathered = t.zeros(self.world * (1 + 3 * self.cap), dtype=t.int64, device=self.dev)

    def gather_device(self, local, n):
        """The collective only: every rank's payload lands in `self.gathered` on every rank (device memory when the
        group is RCCL); no host round trip.  Decode on `dst` with finalize()."""
        t, dist = self.torch, self.dist
        if isinstance(local, np.ndarray):
            words = t.from_numpy(np.ascontiguousarray(local[:n]).view(np.int64).copy()).to(self.dev)
        else:
            words = local[: n * 24].view(t.int64)
        k = min(n, self.cap)
        self.payload[0] = n
        if k:
            self.payload[1:1 + 3 * k] = words[: 3 * k]
        dist.all_gather_into_tensor(self.gathered, self.payload, group=self.group)
        return self.gathered

    def gather_device_async(self, out, totals):
        """gather_device() for the enqueue-only search (AhoCorasick.overlapping_enqueue): the record count is still on
        the device (`totals[0]`), so the payload is assembled there in stream order -- header from `totals`, the first
        `cap` record slots of `out` whatever the count -- and nothing waits for the host.  Decode with finalize()."""
        t = self.torch
        if out.numel() < 24 * self.cap:
            raise ValueError(f"gather_device_async: `out` holds {out.numel()} bytes, the payload takes the first "
                             f"{24 * self.cap} (cap = {self.cap} records)")
        self.payload[0:1] = totals[0:1].view(t.int64).to(self.dev)
        self.payload[1:] = out[: 24 * self.cap].view(t.int64).to(self.dev)
        self.dist.all_gather_into_tensor(self.gathered, self.payload, group=self.group)
        return self.gathered

    def finalize(self, offsets):
        """Host decode of the last gather_device() on `dst` (None elsewhere); raises if a rank overflowed `cap`."""
        block = self.gathered.view(self.world, 1 + 3 * self.cap)
        if self.rank != self.dst:
            return None
        host = block.cpu().numpy()
        counts = host[:, 0]
        if int(counts.max()) > self.cap:
            raise RuntimeError(f"MatchGatherer capacity {self.cap} exceeded ({int(counts.max())} records on one rank)")
        parts = []
        for r in range(self.world):
            a = host[r, 1:1 + 3 * int(counts[r])].copy().view(MATCH_DTYPE)
            if offsets[r]:
                a["start"] += offsets[r]
                a["end"] += offsets[r]
            parts.append(a)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)

    def gather(self, local, n, offsets):
        """local: uint8 tensor (>= n*24 bytes) or numpy MATCH_DTYPE array of this rank's records; offsets[r] is added
        to rank r's start/end on `dst`.  Returns the concatenated numpy array on `dst`, None elsewhere."""
        t, dist = self.torch, self.dist
        if isinstance(local, np.ndarray):
            words = t.from_numpy(np.ascontiguousarray(local[:n]).view(np.int64).copy()).to(self.dev)
        else:
            words = local[: n * 24].view(t.int64)
        k = min(n, self.cap)
        self.payload[0] = n
        if k:
            self.payload[1:1 + 3 * k] = words[: 3 * k]
        dist.all_gather_into_tensor(self.gathered, self.payload, group=self.group)
        block = self.gathered.view(self.world, 1 + 3 * self.cap)
        if self.rank == self.dst:
            host = block.cpu().numpy()
            counts = host[:, 0]
        else:
            host = None
            counts = block[:, 0].cpu().numpy()
        if int(counts.max()) > self.cap:  # rare: exact path for this step (every rank sees the same headers)
            rec = local if isinstance(local, np.ndarray) else local[: n * 24]
            if isinstance(rec, np.ndarray):
                rec = rec[:n]
            out = gather_matches(rec, dst=self.dst, group=self.group, device=self.dev, offset=0)
            self._alloc(max(2 * int(counts.max()), 2 * self.cap))
            if out is not None:
                out = out.copy()
                pos = 0
                for r in range(self.world):
                    c = int(counts[r])
                    out["start"][pos:pos + c] += offsets[r]
                    out["end"][pos:pos + c] += offsets[r]
                    pos += c
            return out
        if host is None:
            return None
        parts = []
        for r in range(self.world):
            c = int(counts[r])
            a = host[r, 1:1 + 3 * c].copy().view(MATCH_DTYPE)
            if offsets[r]:
                a["start"] += offsets[r]
                a["end"] += offsets[r]
            parts.append(a)
        return np.concatenate(parts) if parts else np.zeros(0, dtype=MATCH_DTYPE)
