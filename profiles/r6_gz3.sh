#!/bin/bash
# ordinary gzip end to end: stage marks, and the knobs between the inflate and the writers (pieces in flight, piece size, buffers per file, inflate threads)
export TMPDIR=/tmp
out=gpurun_out/${1:-r6m}
mkdir -p $out
FAQCS_E2E_GZ=1 FAQCS_E2E_MARKS=1 FAQCS_E2E_GZ_TRY="${2:-FAQCS_MI_PARGZ_WINDOW=18;FAQCS_MI_PARGZ_WINDOW=12;FAQCS_MI_PARGZ_PIECE=1048576;FAQCS_MI_PARGZ_PIECE=2097152 FAQCS_MI_PARGZ_WINDOW=18;FAQCS_MI_STREAM_BUFS=8;FAQCS_MI_PARGZ_THREADS=6 FAQCS_MI_PARGZ_WINDOW=14;FAQCS_MI_PARGZ_THREADS=5;FAQCS_MI_PARGZ_THREADS=6 FAQCS_MI_PARGZ_PIECE=2097152 FAQCS_MI_PARGZ_WINDOW=14 FAQCS_MI_STREAM_BUFS=8}" timeout 1700 python3 tools/e2e_big.py 8e6 2>&1 | grep -E "^mapped|^streaming|input|faqcs_mi" > $out/e2e_gz_knobs2.txt
rm -rf /dev/shm/faqcs_e2e_big
cat $out/e2e_gz_knobs2.txt
