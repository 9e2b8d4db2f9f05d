#!/bin/bash
# PMC counters of the trim kernel per read, without Python: bash profiles/pmc_trim.sh <tag> [lib.so] [reads] [L]
# (rocprofv3 --pmc passes over profiles/microbench/trim_ab; one pass per counter set: SQ slots are limited)
set -u
tag=${1:-pmc}; lib=${2:-faqcs_amd/libfaqcs_mi.so}; n=${3:-16777216}; L=${4:-150}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN"; do
    i=$((i+1))
    TRIM_AB_REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o pmc -- ./profiles/microbench/trim_ab $n $L 1 $lib < /dev/null > $out/pmc$i.log 2>&1
done
python3 - "$out" "$n" <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/pmc*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    if "trim" in k or "fold" in k:
        print(k, "launches", max(cnt[(k, c)] for c in acc[k]))
        for c, v in sorted(acc[k].items()):
            print("   %-26s %10.3f per read" % (c, v / cnt[(k, c)] / n))
PY
