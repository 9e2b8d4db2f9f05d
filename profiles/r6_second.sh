#!/bin/bash
# second GPU run of round 6: the whole GPU suite, the bench (all configs), k-mer stats, request width (pair variant)
export TMPDIR=/tmp
out=gpurun_out/r6b
mkdir -p $out
FAQCS_KMER_STATS=1 timeout 900 python bench.py --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer.json 2> $out/bench_kmer.err
./profiles/microbench/req_width 32 268435456 > $out/req_width.txt 2>&1
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/reqw_$n -o p -- ./profiles/microbench/req_width 32 268435456 > $out/reqw_$n.log 2>&1
done
python3 - $out <<'PY' > $out/req_width_counters.txt 2>&1
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(out + "/reqw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:20]][r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc:
    print(k)
    for n, v in sorted(acc[k].items()):
        print("   %-26s %14.0f  (2 launches; gather16 / scatter16: 2^28 accesses each, gather_pair: 2^27 lines each)" % (n, v))
PY
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $out/pytest_gpu.txt
bash profiles/pmc_skm.sh r6b/pmc_skm 16e6 31 > $out/pmc_skm.txt 2>&1
echo done
