#!/bin/bash
# first GPU run of round 6: k-mer tests, k-mer bench A/B (count-in-one-piece vs round 5's table path), request width microbenchmark
export TMPDIR=/tmp
out=gpurun_out/r6a
mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kmer" > $out/pytest_kmer.txt 2>&1
echo "pytest kmer rc=$?" >> $out/pytest_kmer.txt
timeout 600 python bench.py --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer.json 2> $out/bench_kmer.err
FAQCS_KMER_FINAL=0 timeout 600 python bench.py --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer_nofinal.json 2> $out/bench_kmer_nofinal.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_kmer -o s -- python3 $GRAFT_REPO_ROOT/bench.py --config kmer --steps 3 --no-cpu-baseline < /dev/null > $GRAFT_REPO_ROOT/$out/prof_kmer.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/kstats.py $out/prof_kmer 16 > $out/kstats_kmer.txt 2>&1
./profiles/microbench/req_width 32 268435456 > $out/req_width.txt 2>&1
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE WRITE_SIZE" "TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/reqw_$n -o p -- ./profiles/microbench/req_width 32 268435456 > $out/reqw_$n.log 2>&1
done
python3 - $out <<'PY' > $out/req_width_counters.txt 2>&1
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(out + "/reqw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:20]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc:
    print(k)
    for n, v in sorted(acc[k].items()):
        print("   %-26s %14.0f  = %.4f per access (2 launches of 2^28 accesses)" % (n, v, v / (2 * 268435456)))
PY
echo done
