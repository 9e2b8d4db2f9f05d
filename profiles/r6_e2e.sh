#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6j
mkdir -p $out
for i in 1 2 3; do
  for m in 0 1; do
    FAQCS_MI_OUT_PWRITE=$m timeout 600 python3 tools/e2e_big.py 14.3e6 2>&1 | grep -E "^mapped|formatters" | sed "s/^/pwrite=$m /"
  done
done > $out/e2e_pwrite_ab.txt 2>&1
rm -rf /dev/shm/faqcs_e2e_big
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dry_run or native_cli" > $out/pytest_cli.txt 2>&1
FAQCS_MI_OUT_PWRITE=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_cli" > $out/pytest_cli_pwrite.txt 2>&1
echo done
