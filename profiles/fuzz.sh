#!/bin/bash
# Re-seeded runs of every oracle-comparison test (FAQCS_TEST_SEED): bash profiles/fuzz.sh <first seed> <last seed> [out file]
out=${3:-gpurun_out/fuzz.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
for s in $(seq $1 $2); do
  r=$(FAQCS_TEST_SEED=$s python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(oracle or kmer or padded_rows or equal_length) and not 4gib and not full_size and not two_devices and not golden and not at_scale and not rank_kmer" 2>&1 | tail -1)
  echo "seed $s: $r" >> "$out"
done
grep -c passed "$out"; grep -v " passed" "$out" | head
