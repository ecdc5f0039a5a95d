#!/bin/bash
# Build and run tools/lookup_lab.cpp against the in-tree library.  Usage: tools/lookup_lab.sh [B H W half jitter reps]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -std=c++17 -I include -o gpurun_out/lookup_lab tools/lookup_lab.cpp \
    -L focusflow_official_amd/lib -lfocusflow_hip -Wl,-rpath,"$PWD/focusflow_official_amd/lib"
./gpurun_out/lookup_lab "$@"
