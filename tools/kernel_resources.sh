#!/bin/bash
# Registers, spills and occupancy of every kernel in one HIP source (compile-only, no GPU):
#   tools/kernel_resources.sh focusflow_official_amd/csrc/conv_patch.hip [extra hipcc flags]
src=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -I "$root/include" -I "$root/focusflow_official_amd/csrc" "$@" \
    --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "$src" -o /dev/null 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
print("%5s %5s %6s %6s %4s %6s  kernel" % ("VGPR", "AGPR", "vspill", "sspill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n); n = n.replace("void ", "")
    print("%5s %5s %6s %6s %4s %6s  %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]"), n[:150]))
'
