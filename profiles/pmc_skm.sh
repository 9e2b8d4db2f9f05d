#!/bin/bash
# Counters of the combine-before-insert k-mer kernels (skm_extract / skm_split / skm_combine (round 5: super-k-mers)), per k-mer occurrence:
#   bash profiles/pmc_skm.sh <tag> [reads] [log2 slots]
# rocprofv3 --pmc passes over tools/kmer_bench.py (reads of 250 bases sampled from a 50 Mbp synthetic genome); one pass per
# counter set (SQ slots are limited; TCC counters in passes of their own, as MI355X_MICROARCH.md prescribes).
set -u
tag=${1:-kmer_group}; n=${2:-8e6}; slots=${3:-30}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/kmer_bench.py $n 250 $slots 50e6 > $out/kmer_bench.txt 2>&1
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o pmc -- python3 tools/kmer_bench.py $n 250 $slots 50e6 < /dev/null > $out/pmc$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
line = open(out + "/kmer_bench.txt").read().strip().splitlines()[-1]
print(line)
total = float(re.search(r"total (\d+)", line).group(1)); distinct = float(re.search(r"distinct (\d+)", line).group(1))
def name(k):
    m = re.search(r"(skm_\w+|kmer_\w+|trim_\w+|composition_\w+)", k)
    return m.group(1) if m else k[:40]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[name(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(out + "/pmc1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = name(r["Kernel_Name"])
        dur[k] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9; calls[k] += 1
print("occurrences %.0f, distinct keys %.0f (%.3f per occurrence)" % (total, distinct, distinct / total))
for k in sorted(acc):
    if "kmer" not in k and "skm" not in k: continue
    c = acc[k]; t = dur.get(k, 0.0)
    print("%s: %d launches, %.2f ms (under the profiler), %.1f G occurrences/s" % (k, calls[k], t * 1e3, total / t / 1e9 if t else 0))
    for name, v in sorted(c.items()):
        print("   %-26s %12.4f per occurrence" % (name, v / total))
PY
