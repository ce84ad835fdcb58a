#!/bin/bash
# VGPRs / scratch / waves per SIMD of every k_traverse_wide instantiation (compile-only, no GPU needed).
cd "$(dirname "$0")/../nanort_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero --cuda-device-only -Rpass-analysis=kernel-resource-usage -c ${1:-traverse.hip} -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - |
  sed -E 's/Function Name: //' | while IFS=$'\t' read -r name v s o; do printf '%-40s %-28s %-12s %s\n' "$o" "$s" "$v" "$(echo "$name" | c++filt | sed -E 's/\(.*//; s/^void //')"; done
