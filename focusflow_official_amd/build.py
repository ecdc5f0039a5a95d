"""Build libfocusflow_hip.so for gfx950.

In-tree output (git-ignored, shipped to the GPU box by gpurun):
  focusflow_official_amd/lib/libfocusflow_hip.so
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libfocusflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
         "-Wall", "-Wno-unused-function"]


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def build_hip(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(ROOT, "include", "focusflow_hip.h")]

    def compile_one(name):
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name[:-4] + ".o")
        extra = []
        with open(src) as f:
            # files that replay the reference's separately rounded fp32 chains (sampler coordinates, bilinear blend)
            # say so with a pragma; hipcc does not honour it inside templates / lambdas, so they also get the flag
            text = f.read()
        if "#pragma clang fp contract(off)" in text:
            extra = ["-ffp-contract=off"]
        # a file may name further flags of its own: a line "// hipcc-flags: <flags>"
        for line in text.splitlines():
            if line.startswith("// hipcc-flags:"):
                extra += line.split(":", 1)[1].split()
        # experiments: FF_HIPCC_EXTRA_<file stem>="<flags>"
        env_extra = os.environ.get("FF_HIPCC_EXTRA_" + name[:-4])
        if env_extra is not None:
            extra += env_extra.split()
        # the flags an object was built with are recorded beside it: an experimental object (FF_HIPCC_EXTRA_*, an edited
        # hipcc-flags line) is rebuilt as soon as the effective flags differ - also when the variable goes away again
        cmd = [HIPCC, *FLAGS, *extra, "-c", src, "-o", obj]
        stamp, want = obj + ".flags", " ".join(cmd[:-4])
        have = open(stamp).read() if os.path.exists(stamp) else None
        if force or _newer([src] + hdrs, obj) or have != want:
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            with open(stamp, "w") as f:
                f.write(want)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build_hip(force="--force" in sys.argv)
