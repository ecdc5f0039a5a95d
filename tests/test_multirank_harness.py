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
