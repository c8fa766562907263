#!/bin/bash
# per-kernel VGPR / occupancy / LDS / scratch of one HIP source (compile-only): tools/kres.sh gemv.hip [filter]
cd "$(dirname "$0")/../vibevoice_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $VVHIP_CFLAGS -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|Occupancy|LDS Size|ScratchSize" | sed -e 's/.*remark: //' -e 's/ \[-Rpass.*//' | paste - - - - - \
 | sed -e 's/Function Name: //' -e 's/_ZN12_GLOBAL__N_1[0-9]*//' | grep -E "${2:-.}"
