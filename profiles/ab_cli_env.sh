#!/bin/bash
# same-box A/B of a faqcs_mi environment switch through the bench line's e2e object: bash profiles/ab_cli_env.sh VAR [pairs]
var=$1; pairs=${2:-8e6}
for i in 1 2 3; do for v in "" 1; do
  env ${v:+$var=$v} python3 bench.py --pairs 20e6 --steps 1 --warmup 0 --no-cpu-baseline --e2e-pairs $pairs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); e=d['e2e']; print('$var=${v:-unset}', e['value'], e['pipeline_value'], e['seconds'], {k: e['stage_marks_s'][k] for k in ('first pair parsed','last pair submitted','outputs written')})"
done; done
