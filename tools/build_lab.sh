#!/bin/bash
# The LAB build of the library - NOT the product: every source compiled with -DFF_LAB, which adds the timing-only ablations
# (FF_PATCH_ABLATE, FF_LOOKUP_ABLATE, FF_LOOKUP_ABLATE3, FF_CORR_BUILD_ABLATE: WRONG results by design) and conv_dma.hip's
# in-kernel phase stamps; libfocusflow_hip.so contains none of them and refuses to load while one of those variables is set.
#   tools/build_lab.sh [extra hipcc flags, e.g. -DFF_DMA_ABL=3]  ->  focusflow_official_amd/lib/libfocusflow_lab.so
#   FF_LAB_LIB=libfocusflow_lab.so python tools/dma_stamps.py     (FF_LAB_LIB selects the library; bench.py refuses it)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/focusflow_official_amd/lib/lab_obj
mkdir -p "$out"
pids=()
for src in "$root"/focusflow_official_amd/csrc/*.hip; do
  name=$(basename "$src" .hip)
  extra=""
  grep -q "#pragma clang fp contract(off)" "$src" && extra="-ffp-contract=off"
  extra="$extra $(grep '^// hipcc-flags:' "$src" | cut -d: -f2-)"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I "$root/include" -I "$root/focusflow_official_amd/csrc" -DFF_LAB "$@" $extra \
      -c "$src" -o "$out/$name.o" &
  pids+=($!)
  if [ ${#pids[@]} -ge 6 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/focusflow_official_amd/lib/libfocusflow_lab.so" "$out"/*.o
ls -la "$root/focusflow_official_amd/lib/"
