"""On-disk formats either side of the hot path: Middlebury ``.flo``, PFM, KITTI 16-bit PNG flow.

Same functions and semantics as the reference's ``core/utils/frame_utils.py:12-136`` (readFlow :12, readPFM :33,
writeFlow :70, readFlowKITTI :102, readDispKITTI :109, writeFlowKITTI :116, read_gen :123).  The reference goes
through OpenCV for the 16-bit PNGs; OpenCV is not in this image, so the 16-bit RGB / gray PNG codec is written
here on zlib (non-interlaced, 16-bit, colour types 0 and 2 — what KITTI ships and what writeFlowKITTI emits).
"""
import re
import struct
import zlib
from os.path import splitext

import numpy as np

TAG_FLOAT = 202021.25  # "PIEH" as float32


def readFlow(fn):
    """Middlebury .flo -> (H, W, 2) float32, or None when the magic number is wrong (frame_utils.py:12-31)."""
    with open(fn, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or magic[0] != np.float32(TAG_FLOAT):
            print("Magic number incorrect. Invalid .flo file")
            return None
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    return np.resize(data, (h, w, 2))


def writeFlow(filename, uv, v=None):
    """(H, W, 2) or separate u, v planes -> .flo with u,v interleaved per pixel (frame_utils.py:70-99)."""
    if v is None:
        uv = np.asarray(uv)
        assert uv.ndim == 3 and uv.shape[2] == 2
        u, v = uv[:, :, 0], uv[:, :, 1]
    else:
        u, v = np.asarray(uv), np.asarray(v)
    assert u.shape == v.shape
    h, w = u.shape
    with open(filename, "wb") as f:
        np.array([TAG_FLOAT], np.float32).tofile(f)
        np.array([w, h], np.int32).tofile(f)
        np.stack([u, v], axis=-1).astype(np.float32).tofile(f)


def readPFM(file):
    """PFM (colour 'PF' or gray 'Pf'), bottom-up rows, sign of the scale = endianness (frame_utils.py:33-68)."""
    with open(file, "rb") as f:
        header = f.readline().rstrip()
        if header == b"PF":
            color = True
        elif header == b"Pf":
            color = False
        else:
            raise Exception("Not a PFM file.")
        dims = re.match(rb"^(\d+)\s(\d+)\s$", f.readline())
        if not dims:
            raise Exception("Malformed PFM header.")
        width, height = map(int, dims.groups())
        scale = float(f.readline().rstrip())
        data = np.fromfile(f, ("<" if scale < 0 else ">") + "f")
    return np.flipud(np.reshape(data, (height, width, 3) if color else (height, width)))


# ----------------------------------------------------------------------------------------------------------------
# 16-bit PNG (what cv2.imread(..., IMREAD_ANYDEPTH | IMREAD_COLOR) / cv2.imwrite do for KITTI flow maps)
# ----------------------------------------------------------------------------------------------------------------
_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _unfilter(raw, h, stride, bpp):
    """PNG scan-line reconstruction.  The per-byte recurrences of filter types 1, 3 and 4 are native code
    (ff_png_unfilter in libfocusflow_hip.so, host side: a 1242x375 RGB16 KITTI map is 2.8 M dependent steps - seconds per
    sample in a Python loop, which would starve the GPU); the Python loop below is the fallback when the library has
    not been built."""
    try:
        from . import _hip
        lib = _hip.load()
    except Exception:  # noqa: BLE001  (library not built: keep the loader usable)
        lib = None
    if lib is not None:
        buf = np.frombuffer(raw, np.uint8)
        out = np.empty((h, stride), np.uint8)
        rc = lib.ff_png_unfilter(buf.ctypes.data, len(buf), h, stride, bpp, out.ctypes.data)
        if rc != 0:
            raise ValueError("PNG: " + lib.ff_last_error().decode())
        return out
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(h):
        ft = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32)
        pos += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:  # filters 1, 3, 4 depend on the already reconstructed left neighbour
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    c = prev[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                else:
                    raise ValueError(f"PNG filter type {ft}")
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out


def read_png16(filename):
    """PNG -> (H, W) or (H, W, C) array in file channel order (R, G, B); uint16 for 16-bit files, uint8 for 8-bit."""
    with open(filename, "rb") as f:
        blob = f.read()
    if blob[:8] != _PNG_SIG:
        raise ValueError(f"{filename}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(blob):
        n, kind = struct.unpack(">I4s", blob[pos:pos + 8])
        body = blob[pos + 8:pos + 8 + n]
        pos += 12 + n
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    chans = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if chans is None or depth not in (8, 16) or interlace:
        raise ValueError(f"{filename}: unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
    bpp = chans * depth // 8
    rows = _unfilter(zlib.decompress(b"".join(idat)), h, w * bpp, bpp)
    arr = rows.reshape(h, w, chans, depth // 8)
    if depth == 16:
        arr = (arr[..., 0].astype(np.uint16) << 8) | arr[..., 1]
    else:
        arr = arr[..., 0]
    return arr[..., 0] if chans == 1 else arr


def write_png16(filename, arr):
    """(H, W) or (H, W, 3) uint16 -> 16-bit gray / RGB PNG (filter 0, one IDAT)."""
    arr = np.asarray(arr, np.uint16)
    if arr.ndim == 2:
        arr = arr[..., None]
    h, w, c = arr.shape
    assert c in (1, 3)
    be = arr.astype(">u2").tobytes()
    stride = w * c * 2
    raw = b"".join(b"\x00" + be[y * stride:(y + 1) * stride] for y in range(h))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    with open(filename, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0 if c == 1 else 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def readFlowKITTI(filename):
    """KITTI flow PNG: R = 64 u + 2^15, G = 64 v + 2^15, B = valid (frame_utils.py:102-107)."""
    rgb = read_png16(filename).astype(np.float32)
    return (rgb[:, :, :2] - 2 ** 15) / 64.0, rgb[:, :, 2]


def readDispKITTI(filename):
    """KITTI disparity PNG (gray 16-bit / 256) as a horizontal flow (frame_utils.py:109-113)."""
    disp = read_png16(filename).astype(np.float64) / 256.0
    valid = disp > 0.0
    return np.stack([-disp, np.zeros_like(disp)], -1), valid


def writeFlowKITTI(filename, uv):
    """Inverse of readFlowKITTI with every pixel valid (frame_utils.py:116-120)."""
    uv = 64.0 * np.asarray(uv) + 2 ** 15
    valid = np.ones([uv.shape[0], uv.shape[1], 1])
    write_png16(filename, np.concatenate([uv, valid], axis=-1).astype(np.uint16))


def read_gen(file_name, pil=False):
    """Dispatch on the extension (frame_utils.py:123-136): images -> PIL image, .flo/.pfm -> float32 arrays."""
    ext = splitext(file_name)[-1]
    if ext in (".png", ".jpeg", ".ppm", ".jpg"):
        from PIL import Image
        return Image.open(file_name)
    if ext in (".bin", ".raw"):
        return np.load(file_name)
    if ext == ".flo":
        return readFlow(file_name).astype(np.float32)
    if ext == ".pfm":
        flow = readPFM(file_name).astype(np.float32)
        return flow if flow.ndim == 2 else flow[:, :, :-1]
    return []
