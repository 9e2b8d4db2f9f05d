#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6f
mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "adapter or random_batches or golden or phix or edge or full_size_adapter" > $out/pytest_adapter.txt 2>&1
echo "rc=$?" >> $out/pytest_adapter.txt
for i in 1 2; do
  timeout 600 python bench.py --config adapter --steps 5 --no-cpu-baseline --no-other-configs --e2e-pairs 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair on ', d['value'], d['ms_per_step'])"
  FAQCS_ADAPTER_PAIR=0 timeout 600 python bench.py --config adapter --steps 5 --no-cpu-baseline --no-other-configs --e2e-pairs 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair off', d['value'], d['ms_per_step'])"
done > $out/ab_adapter_pair.txt 2>&1
echo done
