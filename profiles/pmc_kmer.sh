#!/bin/bash
# What bounds kmer_count: L2 atomic requests per second.  rocprofv3 --pmc over tools/kmer_bench.py (8 M reads of 250 bases
# sampled from a 50 Mbp synthetic genome, 2^30-slot table): bash profiles/pmc_kmer.sh <tag>
set -u
tag=${1:-kmer}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/kmer_bench.py 8e6 250 30 50e6 > $out/kmer_bench.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $out/pmc -o pmc -- python3 tools/kmer_bench.py 8e6 250 30 50e6 > $out/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_MISS_sum --output-format csv -d $out/pmc2 -o pmc -- python3 tools/kmer_bench.py 8e6 250 30 50e6 > $out/pmc2.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(out + "/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][:40]] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
print(open(out + "/kmer_bench.txt").read().strip().splitlines()[-1])
for k in acc:
    if "kmer_count" in k:
        c = acc[k]; t = dur.get(k, 0.0)
        print(k, {n: int(v) for n, v in sorted(c.items())}, "kernel seconds %.4f" % t)
        if t > 0 and "TCC_ATOMIC_sum" in c:
            print("  L2 atomic requests: %.2f G/s ; L2 requests of all kinds: %.2f G/s ; EA (fabric) atomics: %.2f G/s ; fabric reads %.2f G x 64 B/s"
                  % (c["TCC_ATOMIC_sum"] / t / 1e9, c.get("TCC_REQ_sum", 0) / t / 1e9, c.get("TCC_EA0_ATOMIC_sum", 0) / t / 1e9, c.get("TCC_EA0_RDREQ_sum", 0) / t / 1e9))
PY
