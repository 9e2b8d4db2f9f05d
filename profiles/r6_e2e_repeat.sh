#!/bin/bash
# how much the end-to-end figures of the driver's line move from run to run on one box (five runs of the e2e leg), and compressed input at
# the size of that leg (14.3 M pairs)
export TMPDIR=/tmp
out=gpurun_out/${1:-r6u}
mkdir -p $out
for i in 1 2 3 4 5; do
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs 2>> $out/bench.err < /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['e2e']
print('run $i: one process %.1f, as the caller sees it %.1f, pipeline %.1f M reads/s; marks %s' % (e['value'], e['value_as_the_caller_sees_it'], e['pipeline_value'], json.dumps(e['stage_marks_s'])))"
done > $out/e2e_five_runs.txt
cat $out/e2e_five_runs.txt
FAQCS_E2E_GZ=1 timeout 1700 python3 tools/e2e_big.py 14.3e6 2>&1 < /dev/null | grep -E "^mapped|^streaming|input" > $out/e2e_gz_14Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
grep -E "^mapped|^streaming|^gzip input|^bgzf input" $out/e2e_gz_14Mpairs.txt | grep -v "TWO_PASS\|NO_PARGZ"
