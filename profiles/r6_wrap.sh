#!/bin/bash
# the whole collection once more on the final tree, then three more re-seeded runs of the GPU suite
export TMPDIR=/tmp
tag=${1:-r6v}
bash profiles/collect_r6.sh $tag > gpurun_out/collect_$tag.log 2>&1
mv gpurun_out/collect_$tag.log gpurun_out/$tag/collect.log
for s in 6301 6302 6303; do
  FAQCS_TEST_SEED=$s timeout 800 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/fuzz_seed_$s.txt 2>&1 < /dev/null
  echo "seed $s rc=$? $(grep -E 'passed|failed' gpurun_out/$tag/fuzz_seed_$s.txt | tail -1)" >> gpurun_out/$tag/summary.txt
done
cat gpurun_out/$tag/summary.txt
