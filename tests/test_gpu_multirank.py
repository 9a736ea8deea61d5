"""-m gpu: the N>1 path of bench.py (one process per rank, shard + seam warm-up + single-collective gather) on ONE
GPU: two/three ranks share cuda:0 and talk over gloo (test knobs of bench.py).  Rank 0 regenerates the whole global
haystack on the host, has the CPU ORACLE search it and requires the gathered sharded result to be identical (seam-straddling occurrences are
planted at every shard seam)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_bench_equals_unsharded(world):
    env = dict(os.environ, ACGPU_BENCH_BACKEND="gloo", ACGPU_BENCH_ONE_DEVICE="1", ACGPU_BENCH_VERIFY="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29610 + world), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--gib", "0.25"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["scaling"] == "weak"
    assert d["config"]["sharded_equals_oracle"] is True
    assert d["config"]["ranks"] == world and d["config"]["collective_backend"] == "gloo"
    assert d["config"]["matches"] > 64
