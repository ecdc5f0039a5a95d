#!/bin/bash
# Register / LDS / spill report of one csrc file: tools/kernel_regs.sh conv_patch.hip
cd "$(dirname "$0")/../focusflow_official_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ../../include -I . -c -o /tmp/_regs.o "$1" -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: [^ ]* *//' | paste - - - - - | cut -c1-220
