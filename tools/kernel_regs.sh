#!/bin/bash
# VGPRs / spills / scratch of the FP64 degree-4 MFMA stage kernels for a set of -D flags:  tools/kernel_regs.sh [-DSG_...]
cd "$(dirname "$0")/../seigen_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c kernels_mfma.hip -o /tmp/km_regs_$$.o 2>&1 \
 | grep -E "Function Name|    VGPRs:|VGPR Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' | paste - - - - - \
 | grep -E "Id?Li4E" | sed -e 's/Function Name: _ZN2sg12mfma_stage_//' -e 's/EEvNS_9StageArgsE//'
rm -f /tmp/km_regs_$$.o
