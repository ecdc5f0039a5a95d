#!/bin/bash
# Lab variants of libfocusflow_hip.so (NOT the product): conv_dma.hip with in-kernel phase stamps and timing-only ablations.
#   tools/build_lab.sh "0 1 2 4 7"   ->  focusflow_official_amd/lib/libfocusflow_lab_abl<N>.so  (FF_LAB_LIB=<file> selects one)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/focusflow_official_amd/lib/obj
python -m focusflow_official_amd.build > /dev/null
for n in $1; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I "$root/include" -I "$root/focusflow_official_amd/csrc" -DFF_DMA_STAMPS -DFF_DMA_ABL=$n \
      -c "$root/focusflow_official_amd/csrc/conv_dma.hip" -o /tmp/conv_dma_lab_$n.o &
done
wait
for n in $1; do
  objs=$(ls $obj/*.o | grep -v conv_dma.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/focusflow_official_amd/lib/libfocusflow_lab_abl$n.so" $objs /tmp/conv_dma_lab_$n.o
done
ls -la "$root/focusflow_official_amd/lib/"
