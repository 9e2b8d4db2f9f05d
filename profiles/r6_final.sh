#!/bin/bash
# the last call of round 6: the counter files stamped for the final tree + the driver's line, compressed input end to end on the final
# command line, and three more re-seeded runs of the whole GPU suite
export TMPDIR=/tmp
tag=${1:-r6p}
out=gpurun_out/$tag
mkdir -p $out
bash profiles/collect_r6.sh $tag counters > $out/collect.log 2>&1
FAQCS_E2E_GZ=1 timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 < /dev/null | grep -E "^mapped|^streaming|input|threads .|main thread:|parsers:" > $out/e2e_gz_8Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
for s in 6101 6102 6103; do
  FAQCS_TEST_SEED=$s timeout 800 python -m pytest tests -x -q -m gpu > $out/fuzz_seed_$s.txt 2>&1 < /dev/null
  echo "seed $s rc=$? $(grep -E 'passed|failed' $out/fuzz_seed_$s.txt | tail -1)" >> $out/summary.txt
done
cat $out/summary.txt; grep -E "^gzip input|^bgzf input" $out/e2e_gz_8Mpairs.txt | head -8
