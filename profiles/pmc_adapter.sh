#!/bin/bash
# Instruction mix of adapter_overlap per read (rocprofv3 --pmc, two passes): bash profiles/pmc_adapter.sh <tag> [adapter fraction]
set -u
tag=${1:-adapter}; frac=${2:-0.05}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp FAQCS_ABLATE_ADAPTER_FRAC=$frac
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -o pmc -- python3 tools/ablate.py 0 8e6 --adapter --polyA < /dev/null > $out/pmc$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "adapter" in k:
        print(k, {c: round(v / n[(k, c)] / 8e6, 1) for c, v in sorted(acc[k].items())}, "per read")
PY
