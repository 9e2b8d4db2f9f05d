#!/bin/bash
# A/B of library variants on the k-mer bench: bash tools/ab.sh <reps> <lib> [<lib> ...]   (prints ms/step per run, interleaved)
# FAQCS_SKM_DIAG etc. pass through from the environment.
reps=$1; shift
for r in $(seq $reps); do
  for lib in "$@"; do
    v=$(FAQCS_MI_LIB=$PWD/$lib timeout 250 python3 bench.py --config kmer --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$lib $v"
  done
done
