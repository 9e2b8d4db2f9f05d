#!/bin/bash
# LDS bank-conflict attribution of trim_lds at a given read length: the product library and the diagnostic builds of profiles/build_variant.sh
# (lin = staged-data loads made linear, qbnc = Q-B adds on private banks, umask = one mask row for all lanes, uslk = one S table entry for all lanes;
# all four give WRONG results, they exist to price the conflicts).  bash profiles/lds_attrib16.sh <tag> [L] [reads]
set -u
tag=${1:-attrib16}; L=${2:-250}; n=${3:-8388608}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
for v in product lin qbnc umask uslk; do
  lib=profiles/microbench/libfaqcs_mi_$v.so; [ $v = product ] && lib=faqcs_amd/libfaqcs_mi.so
  [ -f $lib ] || continue
  TRIM_AB_REPS=2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/$v -o pmc -- ./profiles/microbench/trim_ab $n $L 1 $lib > $out/$v.log 2>&1
  python3 - $out/$v $n $v <<'PY'
import csv, glob, sys, collections
d, n, v = sys.argv[1], float(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(float); cnt = collections.Counter(); dur = []
for f in glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trim_lds" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for f in glob.glob(d + "/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "trim_lds" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("%-8s %s ms | " % (v, " ".join("%.3f" % x for x in dur)) + "  ".join("%s %.2f" % (c.replace("SQ_", ""), acc[c] / cnt[c] / n) for c in sorted(acc)))
PY
done | tee $out/summary.txt
