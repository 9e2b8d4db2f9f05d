#!/bin/bash
# compressed input end to end on the GPU box: plain / BGZF / ordinary gzip, with the A/B switches of round 6's inflate work
export TMPDIR=/tmp
out=gpurun_out/${1:-r6k}
mkdir -p $out
FAQCS_E2E_GZ=1 timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 | grep -E "^mapped|^streaming|input|threads .|main thread:|parsers:" > $out/e2e_gz_8Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_cli" > $out/pytest_cli.txt 2>&1
tail -3 $out/pytest_cli.txt
cat $out/e2e_gz_8Mpairs.txt
