"""CPU-side checks: the C ABI library builds/loads and exports every symbol the
header declares; host-side module mirrors the reference's state_dict; the
product path refuses to run without a HIP device (no fallback)."""
import ctypes
import json
import os
import re
import sys
from argparse import Namespace

import pytest
import torch

from conftest import ROOT, golden_spec


def _cfg(ft="1x1conv"):
    return Namespace(TRAIN=Namespace(MASK_CHANNEL=3, MASK_MODAL="point"),
                     MODEL=Namespace(FUSION_TYPE=ft, LOAD_MODULE_TO_BRANCH=False))


@pytest.fixture(scope="module")
def lib_path():
    from focusflow_official_amd import build
    return build.build_hip(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    hdr = open(os.path.join(ROOT, "include", "focusflow_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ff_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 18
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in focusflow_hip.h but not exported"
    from focusflow_official_amd import _hip
    assert set(_hip.EXPORTS) == declared, "ctypes table and header disagree"
    lib.ff_abi_version.restype = ctypes.c_int
    ver = int(re.search(r"#define FF_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.ff_abi_version() == ver == _hip.ABI_VERSION == 7


def test_conv_params_struct_layout(lib_path):
    """ctypes mirror of FFConvParams must match the C layout (size check via a C compile)."""
    import subprocess, tempfile
    from focusflow_official_amd._hip import FFConvParams
    src = '#include <stdio.h>\n#include "focusflow_hip.h"\nint main(){printf("%zu %zu %zu", sizeof(FFConvParams), ' \
          '__builtin_offsetof(FFConvParams, w), __builtin_offsetof(FFConvParams, act_res));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        size, off_w, off_ar = map(int, subprocess.run([os.path.join(d, "t")], capture_output=True, text=True).stdout.split())
    assert ctypes.sizeof(FFConvParams) == size
    assert FFConvParams.w.offset == off_w and FFConvParams.act_res.offset == off_ar


def test_stats_parts_hint_is_host_logic(lib_path):
    """ff_conv2d_stats_parts (no GPU work): which convolutions can deliver the InstanceNorm statistics of their output
    from the epilogue, and how many partial entries per (image, channel) the caller must provide - 2 per 8 x 16 (or 4 x 16)
    output tile of the patch kernel and of the 7x7 stride-2 stem kernel, 1 per tile where conv_dma.hip's fp32-input route takes
    the layer (128 input channels and more), 0 for everything else."""
    from focusflow_official_amd import _hip
    lib = ctypes.CDLL(lib_path)
    lib.ff_conv2d_stats_parts.restype = ctypes.c_int
    lib.ff_conv2d_stats_parts.argtypes = [ctypes.POINTER(_hip.FFConvParams)]

    def parts(cin, cout, k, stride, b, h, w, fmt=_hip.W_F16X3, res2=False):
        p = _hip.FFConvParams()
        p.x[0], p.x_c[0], p.x_ld[0] = 4096, cin, cin          # fake, aligned pointers: nothing is dereferenced
        p.w, p.y, p.y_ld = 4096, 4096, cout
        p.groups, p.B, p.H, p.W, p.Cout = 1, b, h, w, cout
        p.KH = p.KW = k
        p.stride, p.pad_h, p.pad_w = stride, k // 2, k // 2
        p.Ho, p.Wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        p.w_format = fmt
        if res2:
            p.res2 = 4096
        return lib.ff_conv2d_stats_parts(ctypes.byref(p))

    assert parts(64, 64, 3, 1, 16, 192, 256) == 24 * 16 * 2          # 8-row tiles
    assert parts(128, 128, 3, 1, 1, 46, 62) == 12 * 4                # four 32-channel chunks: conv_dma.hip's fp32-input route, ONE entry per 4 x 16 tile (few blocks: 4-row tiles, ragged plane)
    assert parts(128, 128, 3, 1, 16, 48, 64) == 6 * 4                # ... 8-row tiles
    assert parts(96, 96, 3, 1, 1, 46, 62) == 12 * 4 * 2              # three chunks: the patch kernel, two entries per tile
    assert parts(4, 64, 7, 2, 16, 384, 512) == 24 * 16 * 2           # the stem: 192 x 256 output
    assert parts(64, 64, 3, 1, 16, 192, 256, fmt=_hip.W_F16) == 24 * 16 * 2
    assert parts(64, 64, 3, 1, 16, 192, 256, fmt=_hip.W_F32) == 0    # exact-fp32 rows: the generic kernel
    assert parts(64, 96, 3, 2, 16, 192, 256) == 0                    # stride 2: im2col kernel
    assert parts(64, 128, 1, 1, 16, 192, 256) == 0                   # 1x1
    assert parts(48, 64, 3, 1, 2, 32, 32) == 0                       # Cin % 32 != 0
    assert parts(64, 64, 3, 1, 16, 192, 256, res2=True) == 0


def test_pack_job_table_is_host_checked(lib_path):
    """cce._pack_job builds the FFPackJob a packed convolution contributes to ff_pack_weights_table; the launch cannot
    check a table that lives in device memory, so every job passes ff_pack_job_check (host logic, no GPU) first: a
    consistent job passes, and wrong item counts / member layouts / slices are refused with the field named."""
    import torch.nn as nn
    from focusflow_official_amd import _hip, cce
    z, r = nn.Conv2d(160, 48, (1, 5), padding=(0, 2)), nn.Conv2d(160, 80, (1, 5), padding=(0, 2))
    cpu = torch.device("cpu")
    for pc in (cce.PackedConv([z, r], cin_slices=[(96, 160), (0, 32)], use_bias=False), cce.PackedConv([z, r]),
               cce.PackedConv([nn.Conv2d(3, 64, 7, 2, 3)], 4), cce.PackedConv([nn.Conv2d(256, 2, 3, padding=1)])):
        J, fmt, dfmt = cce._pack_job(pc, True, True, cpu)
        _hip.call("ff_pack_job_check", ctypes.byref(J))
        kf, kd = pc.kh * pc.kw * pc.cin_pad, pc.kh * pc.kw * ((pc.cout + 3) // 4 * 4)
        assert J.items_fwd == pc.cout * ((kf + 31) // 32) * 4 and J.items_dgrad == pc.cin_pad * ((kd + 31) // 32) * 4
        assert (J.nmem, J.cout, J.cin, J.nslice) == (len(pc.convs), pc.cout, pc.cin, len(pc.cin_slices or ()))
        assert fmt == (0 if pc.cout <= 2 else dfmt)
    J, _, _ = cce._pack_job(cce.PackedConv([z, r], cin_slices=[(96, 160), (0, 32)]), True, True, cpu)
    for field, value, msg in (("items_fwd", J.items_fwd + 4, "items_fwd"), ("items_dgrad", 8, "items_dgrad"), ("cin", 95, "cin does not match the slices"),
                              ("nmem", 5, "members"), ("cout", J.cout + 1, "do not add up"), ("cout_pad", J.cout - 4, "dgrad rows")):
        keep = getattr(J, field)
        setattr(J, field, value)
        with pytest.raises(_hip.FocusFlowHipError, match=msg):
            _hip.call("ff_pack_job_check", ctypes.byref(J))
        setattr(J, field, keep)
    _hip.call("ff_pack_job_check", ctypes.byref(J))
    # nothing to pack without a recorded use: prepack is a no-op on a fresh module (and never touches a CPU model)
    assert cce.prepack(nn.Sequential(z, r), cpu) == 0


def test_invalid_arguments_are_rejected_without_a_gpu(lib_path):
    """Argument validation happens before any launch, so it is testable here."""
    from focusflow_official_amd import _hip
    lib = _hip.load()
    assert lib.ff_conv2d_fwd(None, None) == -1
    assert b"null params" in lib.ff_last_error()
    p = _hip.FFConvParams()
    p.x[0], p.w, p.y = 16, 16, 16
    p.x_c[0], p.x_ld[0] = 6, 8          # channels not a multiple of 4
    p.groups = p.B = p.H = p.W = p.Cout = p.KH = p.KW = p.stride = 1
    assert lib.ff_conv2d_fwd(ctypes.byref(p), None) == -1
    assert b"multiple of 4" in lib.ff_last_error()
    assert lib.ff_corr_pyramid(16, 16, 16, 16, 10, 4, 4, None) == -1   # plane too small for 4 levels
    arr = (ctypes.c_void_p * 4)(16, 16, 16, 16)
    assert lib.ff_corr_lookup_fwd(arr, 4, 4, 16, 10, 12, 16, 16, 324, None, None) == -1
    assert b"divides by (n-1)" in lib.ff_last_error()   # the reference's own H/8 >= 16 limit


@pytest.mark.parametrize("ft,spec", [("1x1conv", "state_dict_spec"), ("concat", "state_dict_spec_concat")])
def test_state_dict_matches_reference(lib_path, ft, spec):
    from focusflow_official_amd import FF_RAFT_FUSION
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg(ft))
    mine = {k: (tuple(v.shape), str(v.dtype)) for k, v in m.state_dict().items()}
    ref = {k: (s, d) for k, s, d in golden_spec(spec)}
    assert mine == ref
    if ft == "1x1conv":
        assert sum(p.numel() for p in m.parameters()) == 7662272
    # shared BN registration (extractor.py:25-26,44-45): norm3 IS downsample.1
    blk = m.flow_net.cnet.layer2[0]
    assert blk.norm3 is blk.downsample[1]
    # reference API surface used by train.py
    m.flow_net.freeze_bn()
    assert not m.flow_net.cnet.norm1.training
    m.train()
    m.freeze_self()
    assert not m.flow_net.fnet.conv1.weight.requires_grad and m.flow_net.fnet.mask_conv1.weight.requires_grad
    assert m.flow_net.update_block.flow_head.conv1.weight.requires_grad
    assert not m.flow_net.update_block.gru.convz1.weight.requires_grad


def test_unsupported_configurations_fail_loudly(lib_path):
    from focusflow_official_amd import FF_RAFT_FUSION
    with pytest.raises(NotImplementedError):
        FF_RAFT_FUSION(use_fusion="attention", cfg=_cfg())
    for ft in ("SA", "CA"):  # attention fusion units are built: state_dict must carry the reference's key names
        keys = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg(ft)).state_dict().keys()
        assert [k for k, _, _ in golden_spec(f"state_dict_spec_{ft.lower()}")] == list(keys)
    with pytest.raises(ValueError):
        FF_RAFT_FUSION(use_fusion="parallel", fuse_cnet=True, cfg=_cfg("bogus"))


def test_no_cpu_fallback(lib_path):
    """Calling the model with CPU tensors must raise, not silently compute on the host."""
    from focusflow_official_amd import FF_RAFT_FUSION
    from focusflow_official_amd._hip import FocusFlowHipError
    m = FF_RAFT_FUSION(use_fusion="parallel", fusion_channels=256, fuse_cnet=True, cfg=_cfg()).eval()
    x = torch.zeros(1, 3, 128, 128)
    with pytest.raises(FocusFlowHipError):
        m(x, x, torch.zeros(1, 1, 128, 128), torch.zeros(1, 1, 128, 128), raft_iters=1, test_mode=True)


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "focusflow_official_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("test oracle", ""), f"{f} mentions the oracle"


def test_the_product_library_carries_no_timing_only_ablations(lib_path):
    """the ablation instances (results WRONG by design) and their getenv()s live behind -DFF_LAB in
    the lab build (tools/build_lab.sh); libfocusflow_hip.so contains none of their switches, the loader refuses to come up
    while one is set in the environment, and bench.py refuses every lab / tuning switch."""
    import subprocess
    import sys
    blob = open(lib_path, "rb").read()
    for name in (b"FF_PATCH_ABLATE", b"FF_LOOKUP_ABLATE", b"FF_CORR_BUILD_ABLATE", b"FF_WS_ABLATE", b"FF_DMA_ABL"):
        assert name not in blob, f"{name.decode()} found in the product library"
    env = dict(os.environ, FF_LOOKUP_ABLATE3="4")
    env.pop("FF_LAB_LIB", None)
    r = subprocess.run([sys.executable, "-c", "from focusflow_official_amd import _hip; _hip.load()"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "FF_LOOKUP_ABLATE3" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], cwd=ROOT, env=dict(os.environ, FF_DMA_TILE="4"),
                       capture_output=True, text=True)
    assert r.returncode != 0 and "FF_DMA_TILE" in (r.stderr + r.stdout)
