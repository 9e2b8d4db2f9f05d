#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6h
mkdir -p $out
./profiles/microbench/tmpfs_write 8 16 > $out/tmpfs_write_16threads.txt 2>&1
./profiles/microbench/tmpfs_write 8 32 > $out/tmpfs_write_32threads.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
timeout 900 python bench.py 2> $out/bench_default.err > $out/bench_default.json
FAQCS_E2E_GZ=1 timeout 900 python3 tools/e2e_big.py 8e6 > $out/e2e_gz.txt 2>&1
FAQCS_E2E_GZ=1 FAQCS_MI_PARGZ_TWO_PASS=1 timeout 900 python3 tools/e2e_big.py 8e6 > $out/e2e_gz_two_pass.txt 2>&1
rm -rf /dev/shm/faqcs_e2e_big
echo done
