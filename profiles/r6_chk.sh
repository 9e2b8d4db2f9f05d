#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6k
mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
echo done
