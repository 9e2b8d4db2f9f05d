#!/bin/bash
# same-box A/B of adapter_overlap builds: bash profiles/ab_adapter_variants.sh <name> ...  (profiles/microbench/libfaqcs_mi_<name>.so), 4 rounds
for i in 1 2 3 4; do for v in "$@"; do
  echo "$v $(FAQCS_MI_LIB=$PWD/profiles/microbench/libfaqcs_mi_$v.so FAQCS_ABLATE_ADAPTER_FRAC=0.05 python3 tools/ablate.py 0 8e6 --adapter --polyA 2>/dev/null | tail -1)"
done; done
