for p in 8388608 25165824 50e6 100e6; do
  python3 bench.py --pairs $p --steps 5 --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pairs $p', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
done
python3 bench.py --pairs 100e6 --steps 5 --batch-reads 33554432 --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch 2^25', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
python3 bench.py --pairs 100e6 --steps 5 --batch-reads 8388608 --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('batch 2^23', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['launches'])"
