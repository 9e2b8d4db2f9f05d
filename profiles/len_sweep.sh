#!/bin/bash
# Usage (GPU box): bash profiles/len_sweep.sh <tag> <len> [<len> ...]
# 2 x <len> plain shape on trim_lds and (FAQCS_TRIM_LDS=0) on the kernels it replaced: value and dominant kernel per length.
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for L in "$@"; do
  for lds in 1 0; do
    FAQCS_TRIM_LDS=$lds timeout 300 python bench.py --pairs 20e6 --read-len $L --steps 3 --warmup 1 --no-cpu-baseline --e2e-pairs 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('len $L lds=$lds', d['roofline'].get('kernel'), '%.0f' % d['value'], d['unit'], '(kernel %.3f ms)' % d['kernels_ms'].get(d['roofline'].get('kernel'), 0) if 'kernels_ms' in d else '')"
  done
done | tee $out/len_sweep.txt
