#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6e
mkdir -p $out
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-other-configs --e2e-pairs 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail fold on ', d['value'], d['ms_per_step'], d['roofline']['frac'])"
  FAQCS_TAIL_FOLD=0 timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-other-configs --e2e-pairs 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail fold off', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done > $out/ab_tail_fold.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_plain -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --no-cpu-baseline --no-other-configs --e2e-pairs 0 < /dev/null > $GRAFT_REPO_ROOT/$out/prof_plain.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py $out/prof_plain 8 > $out/kstats_plain.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
echo done
