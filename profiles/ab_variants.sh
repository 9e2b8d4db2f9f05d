#!/bin/bash
# Usage (GPU box): bash profiles/ab_variants.sh <name> [<name> ...]  -- tools/ablate.py (one 16 M-read launch per pass) under
# profiles/microbench/libfaqcs_mi_<name>.so ("base" = the product library), two rounds
for i in 1 2; do for v in "$@"; do
  lib=$PWD/profiles/microbench/libfaqcs_mi_$v.so; [ $v = base ] && lib=$PWD/faqcs_amd/libfaqcs_mi.so
  echo "$v: $(FAQCS_MI_LIB=$lib python3 tools/ablate.py 0 16e6 2>/dev/null | tail -1)"
done; done
