#!/bin/bash
# compressed input end to end with the per-role thread accounting of FAQCS_MI_TIMING (which stage do the others wait for?)
export TMPDIR=/tmp
out=gpurun_out/${1:-r6k}
mkdir -p $out
FAQCS_E2E_GZ=1 timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 | grep -E "^mapped|^streaming|input|threads .|main thread:|parsers:" > $out/e2e_gz_threads.txt
rm -rf /dev/shm/faqcs_e2e_big
cat $out/e2e_gz_threads.txt
