"""N>1 path on CPU: shard planning + the gloo gather (world_size 2), no GPU needed."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_shards_tiles_the_span():
    from aho_corasick_amd.distributed import plan_shards
    for s, e, w in [(0, 1 << 20, 8), (10, 1000, 3), (5, 5, 4), (0, 63, 2), (7, 7 + 64 * 9 + 1, 4), (0, 1 << 33, 8)]:
        sh = plan_shards(s, e, w)
        assert len(sh) == w and sh[0][0] == s and sh[-1][1] == e
        for (a, b), (c, d) in zip(sh[:-1], sh[1:]):
            assert b == c and a <= b
        for a, b in sh[:-1]:
            assert (b - s) % 64 == 0 or b == e


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from aho_corasick_amd.api import MATCH_DTYPE
    from aho_corasick_amd.distributed import gather_matches
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        n = 5 if rank == 0 else 0 if rank == 1 else 3
        loc = np.zeros(n, dtype=MATCH_DTYPE)
        loc["pattern"] = np.arange(n) + 100 * rank
        loc["start"] = np.arange(n) + 1000 * rank
        loc["end"] = loc["start"] + 4
        out = gather_matches(loc, dst=0)
        if rank == 0:
            q.put([(int(p), int(s), int(e)) for p, s, e in zip(out["pattern"], out["start"], out["end"])])
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gather_matches_gloo_world2_and_3():
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        want = [(i, i, i + 4) for i in range(5)]
        if world == 3:
            want += [(200 + i, 2000 + i, 2004 + i) for i in range(3)]
        assert got == want


def _worker_fixed(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from aho_corasick_amd.api import MATCH_DTYPE
    from aho_corasick_amd.distributed import MatchGatherer
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        g = MatchGatherer(cap=4)
        offsets = [1000 * r for r in range(world)]
        results = []
        for step, sizes in enumerate([(3, 0, 2), (4, 4, 4), (9, 1, 0), (5, 6, 7)]):   # step 2 overflows cap=4, then cap grows
            n = sizes[rank % 3]
            loc = np.zeros(n + 2, dtype=MATCH_DTYPE)          # two trailing records that must not be sent
            loc["pattern"] = np.arange(n + 2) + 100 * rank + 10 * step
            loc["start"] = np.arange(n + 2)
            loc["end"] = loc["start"] + 4
            out = g.gather(loc, n, offsets)
            if step == 0:   # the split forms used by bench.py give the same records
                g.gather_device(loc, n)
                out2 = g.finalize(offsets)
                assert (out2 is None) == (rank != 0)
                if rank == 0:
                    assert out2.tobytes() == out.tobytes()
                # pipelined form: count in a totals tensor, payload = the first `cap` record slots whatever the count
                import torch
                slots = np.zeros(max(g.cap, n + 2), dtype=MATCH_DTYPE)
                slots[: n + 2] = loc
                rec = torch.from_numpy(slots.view(np.uint8).copy())
                g.gather_device_async(rec, torch.tensor([n, 0], dtype=torch.int64))
                out3 = g.finalize(offsets)
                if rank == 0:
                    assert out3.tobytes() == out.tobytes()
            if rank == 0:
                results.append([(int(p), int(s), int(e)) for p, s, e in zip(out["pattern"], out["start"], out["end"])])
            else:
                assert out is None
        if rank == 0:
            q.put(results)
    finally:
        dist.destroy_process_group()


def test_match_gatherer_single_collective_gloo():
    for world in (2, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_fixed, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = q.get(timeout=120)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        for step, sizes in enumerate([(3, 0, 2), (4, 4, 4), (9, 1, 0), (5, 6, 7)]):
            want = []
            for r in range(world):
                n = sizes[r % 3]
                want += [(i + 100 * r + 10 * step, i + 1000 * r, i + 4 + 1000 * r) for i in range(n)]
            assert got[step] == want, (world, step)
