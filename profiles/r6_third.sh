#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r6c
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmer" > $out/pytest_kmer.txt 2>&1
echo "pytest kmer rc=$?" >> $out/pytest_kmer.txt
FAQCS_KMER_STATS=1 timeout 900 python bench.py --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer.json 2> $out/bench_kmer.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_kmer -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config kmer --steps 3 --no-cpu-baseline < /dev/null > $GRAFT_REPO_ROOT/$out/prof_kmer.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py $out/prof_kmer 12 > $out/kstats_kmer.txt 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
echo done
