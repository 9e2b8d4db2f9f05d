#!/bin/bash
# the final command line: the whole GPU suite, compressed input end to end, the driver's line
export TMPDIR=/tmp
out=gpurun_out/${1:-r6s}
mkdir -p $out
timeout 900 python -m pytest tests -x -q -m gpu > $out/gpu_suite.txt 2>&1 < /dev/null
grep -E "passed|failed" $out/gpu_suite.txt | tail -1
FAQCS_E2E_GZ=1 FAQCS_E2E_MARKS=1 timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 < /dev/null | grep -E "^mapped|^streaming|input|faqcs_mi" > $out/e2e_gz_8Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
grep -E "^mapped|^streaming|^gzip input|^bgzf input" $out/e2e_gz_8Mpairs.txt
timeout 900 python3 bench.py > $out/bench_default_all_configs.json 2> $out/bench.err < /dev/null
tail -c 600 $out/bench_default_all_configs.json
