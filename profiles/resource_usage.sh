#!/bin/bash
# Registers and spills of every trim_lds instantiation (compiler remarks; no GPU needed): bash profiles/resource_usage.sh [-DFOO ...]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" \
    -c faqcs_amd/csrc/faqcs_trim_lds_kernel.hip -o /tmp/faqcs_lds_dev.o 2>&1 | grep -E "Function Name|VGPRs:|Spill|Occupancy" | paste - - - - - | grep "trim_lds" |
    grep "Function Name: _Z8trim_lds" | sed -E 's/.*Function Name: _Z8trim_ldsILi([0-9]+)ELi([0-9]+)ELb([01])ELb([01])ELi([0-9]+)ELi([0-9]+)E.*VGPRs: ([0-9]+).*SGPRs Spill: ([0-9]+).*VGPRs Spill: ([0-9]+).*/trim_lds<C=\1, NW=\2, WINDOWED=\3, EXT=\4, LPR=\5, RPC=\6>: \7 VGPRs, \8 SGPR spills (to VGPR lanes), \9 VGPR spills (to scratch)/'
