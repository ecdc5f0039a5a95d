"""Compiler-reported resources of the hand-written kernels (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).

FF-PWC's cost volume ran 4 x slower than its LDS traffic allows for three rounds because the compiler hoisted every LDS read of a channel
chunk above the arithmetic: 256 registers, 700 bytes of scratch per lane, one wave per SIMD - and no parity test can see that.  The
files checked here hold the bandwidth-shaped kernels, which have no reason to spill at all; the MFMA kernels are scanned by hand
(tools/scan_spills.py: minutes of compile time)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FILES = ("pwc.hip", "norm.hip", "train_ops.hip", "corr_lookup_dma.hip")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("name", FILES)
def test_bandwidth_kernels_do_not_spill(name):
    import scan_spills
    kernels = scan_spills.scan(os.path.join(scan_spills.CSRC, name))
    assert kernels, f"no kernel reported for {name}"
    spilled = {k["name"]: int(k.get("ScratchSize", "0")) for k in kernels if int(k.get("ScratchSize", "0")) > 0}
    assert not spilled, f"{name}: kernels with scratch (bytes per lane): {spilled}"
    if name == "pwc.hip":
        cv = [k for k in kernels if "costvolume_fwd" in k["name"]]
        assert cv and all(int(k["VGPRs"]) <= 168 and int(k["Occupancy"]) >= 3 for k in cv), cv
