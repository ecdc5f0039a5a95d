import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _host_cores() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (a GPU box shows every core of
    the host but grants 16: torch's default thread count would oversubscribe them many times over)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    torch.set_num_threads(_host_cores())      # the CPU oracle runs beside every GPU test


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_spec(name="state_dict_spec"):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return [(k, tuple(s), d) for k, s, d in json.load(f)]


@pytest.fixture(scope="session")
def det_sd():
    from oracle.weights import det_tensor
    return {k: det_tensor(k, s) for k, s, _ in golden_spec()}


@pytest.fixture(scope="session")
def det_sd_concat():
    from oracle.weights import det_tensor
    return {k: det_tensor(k, s) for k, s, _ in golden_spec("state_dict_spec_concat")}


@pytest.fixture(scope="session")
def det_sd_sa():
    from oracle.weights import det_tensor
    return {k: det_tensor(k, s) for k, s, _ in golden_spec("state_dict_spec_sa")}


@pytest.fixture(scope="session")
def det_sd_ca():
    from oracle.weights import det_tensor
    return {k: det_tensor(k, s) for k, s, _ in golden_spec("state_dict_spec_ca")}
