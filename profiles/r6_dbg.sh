#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6d
mkdir -p $out
( for a in "2400 250 200 400 2" "2400 250 200 400 1" "2400 250 200 400 4"; do
  FAQCS_KMER_STATS=1 timeout 120 python tools/owner_repro.py $a 2>&1 | grep -v amdgpu.ids
done ) > $out/owner_repro.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
echo done
