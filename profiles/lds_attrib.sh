#!/bin/bash
# LDS-time attribution of trim_lds: the product build and the three diagnostic builds of profiles/build_variant.sh
# (no Q-B atomics / no S-B table look-ups / no S-A table look-ups), one PMC pass each.  bash profiles/lds_attrib.sh <tag>
set -u
tag=${1:-attrib}
mkdir -p gpurun_out/$tag
for v in "" _noqb _nosb _nosa; do
  lib=$PWD/faqcs_amd/libfaqcs_mi$v.so
  [ -f "$lib" ] || continue
  PMC="SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" bash profiles/pmc_insts.sh $tag/pmc$v FAQCS_MI_LIB=$lib > gpurun_out/$tag/pmc$v.txt 2>&1
  echo "variant '${v:-product}':"; grep -v "^$" gpurun_out/$tag/pmc$v.txt | tail -1
  FAQCS_MI_LIB=$lib python tools/ablate.py 0 16e6 | tail -1
done | tee gpurun_out/$tag/summary.txt
