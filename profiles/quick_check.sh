#!/bin/bash
# Usage (GPU box): bash profiles/quick_check.sh <tag>   -- parity subset, kernel stats of a short bench run, one bench line
tag=${1:-quick}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "golden_cases_on_gpu or random_batches or full_size" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o plain -- python3 bench.py --pairs 25165824 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/prof.log 2>&1
python3 - $out/prof/plain_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:44], r['Calls'], 'avg %.3f ms min %.3f max %.3f' % (float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
PY
python3 bench.py --no-cpu-baseline --e2e-pairs 0 2>/dev/null | tee $out/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
