#!/bin/bash
# Instruction mix of the trim kernel (rocprofv3 --pmc, own pass): bash profiles/pmc_insts.sh <tag> [env assignments...]
set -u
tag=${1:-insts}; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES} --output-format csv -d $out/pmc -o pmc -- python3 tools/ablate.py 0 16e6 > $out/pmc.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k in acc:
    if "trim" in k:
        print(k, {c: round(v / n[(k, c)] / 16e6, 2) for c, v in acc[k].items()}, "per read; launches", max(n[(k, c)] for c in acc[k]))
PY
