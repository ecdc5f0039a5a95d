"""ctypes front-end of oracle/corr_oracle.c (TEST INFRASTRUCTURE)."""
import ctypes as C

import numpy as np

from . import build_oracle

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle.build())
    return _lib


def _f(a):
    return a.ctypes.data_as(C.c_void_p)


def corr_volume(f1: np.ndarray, f2: np.ndarray) -> np.ndarray:
    b, c, h, w = f1.shape
    f1 = np.ascontiguousarray(f1, np.float32)
    f2 = np.ascontiguousarray(f2, np.float32)
    vol = np.empty((b, h * w, h * w), np.float32)
    lib().orc_corr_volume(_f(f1), _f(f2), _f(vol), b, c, h * w)
    return vol


def pyramid(vol: np.ndarray, h: int, w: int, levels=4):
    out = [np.ascontiguousarray(vol.reshape(-1, h, w), np.float32)]
    for _ in range(levels - 1):
        hh, ww = out[-1].shape[1:]
        nxt = np.empty((out[-1].shape[0], hh // 2, ww // 2), np.float32)
        lib().orc_avg_pool2(_f(out[-1]), _f(nxt), C.c_long(out[-1].shape[0]), hh, ww)
        out.append(nxt)
    return out


def lookup(levels, coords: np.ndarray, radius=4):
    """levels: list of [B*Q][h][w]; coords (B,2,H,W) -> out (B,K,H,W), taps [B*Q][L][2][2r+1]."""
    b, _, h, w = coords.shape
    coords = np.ascontiguousarray(coords, np.float32)
    nl, win = len(levels), 2 * radius + 1
    out = np.empty((b, nl * win * win, h, w), np.float32)
    taps = np.empty((b * h * w, nl, 2, win), np.int32)
    arr = (C.c_void_p * nl)(*[lv.ctypes.data for lv in levels])
    lib().orc_corr_lookup(arr, nl, radius, _f(coords), b, h, w, _f(out), _f(taps))
    return out, taps


def div5_mismatches(xs: np.ndarray, d: int):
    """How many of xs give (2 x) / d != the five fused operations of csrc/corr_lookup_dma.hip (and the first index)."""
    xs = np.ascontiguousarray(xs, np.float32)
    first = C.c_long(-1)
    f = lib().orc_div5_mismatches
    f.restype = C.c_long
    return f(_f(xs), C.c_long(xs.size), int(d), C.byref(first)), first.value


def pwc_costvolume_kernel(one: np.ndarray, two: np.ndarray) -> np.ndarray:
    """FF-PWC cost volume by the loop-level transcription of correlation.py:7-102 (NCHW fp32 -> (B,81,H,W))."""
    b, c, h, w = one.shape
    one = np.ascontiguousarray(one, np.float32)
    two = np.ascontiguousarray(two, np.float32)
    top = np.empty((b, 81, h, w), np.float32)
    rc = lib().orc_pwc_costvolume_kernel(_f(one), _f(two), _f(top), b, c, h, w)
    assert rc == 0
    return top
