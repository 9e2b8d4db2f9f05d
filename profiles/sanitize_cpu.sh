#!/bin/bash
# AddressSanitizer + UBSan on the CPU-side code (GPU sanitizers are not available on the pool): the oracle over the 60 golden cases and the
# host-only entry points of faqcs_mi (BGZF reader, report script) over tests/test_cli_host.py.  Run from the repo root after build().
set -e
gcc -O1 -g -std=c11 -fPIC -shared -fsanitize=address,undefined -ffp-contract=off -o /tmp/libfaqcs_oracle_asan.so oracle/faqcs_oracle.c -lm
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/faqcs_mi_asan faqcs_amd/csrc/faqcs_cli.cpp -Lfaqcs_amd -lfaqcs_mi -lz -Wl,-rpath,$PWD/faqcs_amd
cp oracle/libfaqcs_oracle.so /tmp/liboracle_orig.so; cp faqcs_amd/faqcs_mi /tmp/faqcs_mi_orig
trap 'cp /tmp/liboracle_orig.so oracle/libfaqcs_oracle.so; cp /tmp/faqcs_mi_orig faqcs_amd/faqcs_mi' EXIT
cp /tmp/libfaqcs_oracle_asan.so oracle/libfaqcs_oracle.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py -q -x
cp /tmp/faqcs_mi_asan faqcs_amd/faqcs_mi
ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_cli_host.py -q -x
