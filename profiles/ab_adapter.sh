#!/bin/bash
# same-box A/B of adapter_overlap: product library vs profiles/microbench/libfaqcs_mi_old.so, kernel-only time at 5 % read-through
for i in 1 2 3; do for v in old new; do
  lib=$PWD/faqcs_amd/libfaqcs_mi.so; [ $v = old ] && lib=$PWD/profiles/microbench/libfaqcs_mi_old.so
  echo "$v $(FAQCS_MI_LIB=$lib FAQCS_ABLATE_ADAPTER_FRAC=0.05 python3 tools/ablate.py 0 8e6 --adapter --polyA 2>/dev/null | tail -1)"
done; done
