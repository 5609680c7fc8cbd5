#!/bin/bash
# usage: tools/kernel_regs.sh <file.hip> [grep pattern on the demangled name]: VGPR / AGPR / spill / LDS / occupancy per kernel (compile only, no GPU)
cd "$(dirname "$0")/../hpfg_amd/csrc" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /tmp/kregs_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "remark: +(Function Name|VGPRs:|AGPRs:|VGPRs Spill|ScratchSize|Occupancy|LDS Size)" \
  | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - - - \
  | while IFS=$'\t' read -r name rest; do n=$(echo "${name#Function Name: }" | c++filt); echo "$n | $rest"; done \
  | grep -E "${2:-.}"
rm -f /tmp/kregs_$$.o
