"""Deterministic, name-hashed weights for parity tests (test infrastructure).

Every tensor of a state_dict is filled from its own ``torch.Generator`` seeded
with crc32(key), so the reference (in the authoring container), the oracle and
the HIP model can be given bit-identical parameters without shipping 30 MB.
"""
import math
import zlib

import torch


def det_tensor(key: str, shape, dtype=torch.float32, flow_head_damp: float = 0.05) -> torch.Tensor:
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(shape)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    # A raw random flow head makes the 12-step recurrence chaotic (fp32 vs fp64
    # of the SAME code diverge by tens of pixels); a small one gives the
    # contractive behaviour of a trained RAFT, so parity thresholds mean something.
    # 0.05 is contractive for 12 iterations up to 384x512; BASELINE config 5 (544x960, 32 iterations) needs 0.01
    # (tests/golden/make_golden_c5.py: fp32-vs-fp64 spread of the reference 2.8e-4 px instead of 0.45 px).
    damp = flow_head_damp if ".flow_head.conv2." in key else 1.0
    if len(shape) == 4:  # conv weight, OIHW
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * (damp * math.sqrt(2.0 / fan_in))
    if leaf == "weight":  # norm scale
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    return damp * 0.1 * torch.randn(shape, generator=g)  # biases


def det_state_dict(spec) -> dict:
    """spec: iterable of (key, shape) in state_dict order."""
    return {k: det_tensor(k, s) for k, s in spec}


def fill_module(module: torch.nn.Module) -> None:
    """Overwrite every parameter/buffer of ``module`` with its det_tensor."""
    sd = module.state_dict()
    module.load_state_dict({k: det_tensor(k, v.shape) for k, v in sd.items()}, strict=True)
