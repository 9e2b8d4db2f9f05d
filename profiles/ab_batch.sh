for i in 1 2 3; do for b in 16777216 33554432; do
  python3 bench.py --batch-reads $b --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch $b', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'], d['roofline']['frac'])"
done; done
