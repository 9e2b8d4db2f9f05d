#!/bin/bash
# Usage (GPU box): bash profiles/ab_env.sh VAR v1 v2 ...   -- the default bench line under VAR=v for each value, three rounds
var=$1; shift
for i in 1 2 3; do for v in "$@"; do
  env $var=$v python3 bench.py --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$var=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
