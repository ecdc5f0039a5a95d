"""Registers, scratch and occupancy of every kernel in csrc/ as the compiler reports them (hipcc -Rpass-analysis=kernel-resource-usage).
A kernel with scratch bytes spills registers to memory - costvolume_fwd_kernel ran 4 x slower than its LDS traffic allows for
three rounds because of that (700 bytes per lane) without any test noticing.

    python tools/scan_spills.py [--all]        (default: kernels with scratch only)
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "focusflow_official_amd", "csrc")


def scan(path):
    flags = []
    with open(path) as f:
        for line in f:
            if "hipcc-flags:" in line:
                flags = line.split("hipcc-flags:")[1].split()
                break
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", *flags, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-c", path, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            out.append(cur)
        elif cur is not None:
            cur[k.split()[0]] = v
    return out


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    except OSError:
        return n


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    bad = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        for k in scan(path):
            sc = int(k.get("ScratchSize", "0"))
            if sc or show_all:
                bad += sc > 0
                print(f"{os.path.basename(path):24s} VGPR {k.get('VGPRs', '?'):>4s} AGPR {k.get('AGPRs', '?'):>4s} scratch {sc:5d} B  occupancy {k.get('Occupancy', '?')}  "
                      f"LDS {k.get('LDS', '?'):>6s}  {demangle(k['name'])[:110]}")
    print(f"{bad} kernel(s) with scratch")
