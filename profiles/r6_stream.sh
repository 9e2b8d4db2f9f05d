#!/bin/bash
# the streaming path's new output side (formatter pool + ordered committers) and the background allocation of the pinned buffers:
# the command-line tests of the GPU suite, then compressed input end to end with the thread accounting
export TMPDIR=/tmp
out=gpurun_out/${1:-r6q}
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_cli or golden_cli or cli" > $out/pytest_cli.txt 2>&1 < /dev/null
tail -2 $out/pytest_cli.txt
FAQCS_E2E_GZ=1 FAQCS_E2E_MARKS=1 FAQCS_E2E_GZ_TRY="${2:-FAQCS_MI_STREAM_FORMATTERS=2;FAQCS_MI_STREAM_FORMATTERS=6}" timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 < /dev/null | grep -E "^mapped|^streaming|input|faqcs_mi" > $out/e2e_gz_8Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
grep -E "^mapped|^streaming|^gzip input|^bgzf input" $out/e2e_gz_8Mpairs.txt | grep -v "TWO_PASS\|NO_PARGZ\|BGZF_ZLIB"
