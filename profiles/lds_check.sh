#!/bin/bash
# Parity subset + PMC instruction mix + kernel time of the trim kernel: bash profiles/lds_check.sh <tag> [env assignments...]
set -u
tag=${1:-chk}; shift
out=gpurun_out/$tag
mkdir -p $out
for kv in "$@"; do export "$kv"; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random or edge or width or offset or quality_error" > $out/pytest.log 2>&1
tail -12 $out/pytest.log
bash profiles/pmc_insts.sh $tag/insts1 "$@" > $out/insts1.txt 2>&1
PMC="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH" bash profiles/pmc_insts.sh $tag/insts2 "$@" > $out/insts2.txt 2>&1
cat $out/insts1.txt $out/insts2.txt | grep -v "^$" | tail -4
python tools/ablate.py 0 16e6 | tee $out/ablate.txt
