"""N>1 path on CPU: two gloo ranks through bench.py's harness (sharding of independent frame
pairs, barrier, max-over-ranks timing, one JSON line from rank 0).  The forward path has no
data-path collective; the model itself is replaced by a sleep (--harness-selftest)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def test_shard_units_tiles_the_batch():
    import bench
    for total in (8, 16, 64, 13, 1):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_units(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_harness():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "5", "--warmup", "0", "--harness-selftest"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["shard"] == [0, 8]
    # rank 1 sleeps 20 ms per step: the reported time is the MAX over ranks
    assert d["ms_per_step"] >= 19.0
    assert abs(d["value"] - 16 * 5 / (d["ms_per_step"] * 5e-3)) < 1e-6 * d["value"]


def test_bench_launches_its_own_ranks_without_torchrun():
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE: the parent must start the two ranks itself
    (the driver's N=1 command with a different --gpus) and relay rank 0's single JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "0",
           "--harness-selftest"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["shard"] == [0, 8] and d["ms_per_step"] >= 19.0


@pytest.mark.parametrize("flags,graph", [(["--graph"], True), ([], True), (["--eager"], False)])
def test_two_ranks_replay_a_graph_by_default_and_still_emit_one_line(flags, graph):
    """Multi-rank forward steps are replayed from a hipGraph unless --eager (host-independent ranks); the launcher, the
    sharding and the single JSON line do not change with the flag."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--harness-selftest"] + flags
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["shard"] == [0, 8] and d["hipgraph"] is graph


def test_self_launch_propagates_a_rank_failure():
    """No GPU here: the forward bench must fail in every rank, and the launcher must return non-zero instead of hanging."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
