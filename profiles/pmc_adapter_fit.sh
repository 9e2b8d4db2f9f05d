#!/bin/bash
# Fixed per-read cost vs per-adapter cost of adapter_overlap: SQ_INSTS_* per read with 1 (--polyA), 9 (--adapter) and 10 adapters, no read-through
out=gpurun_out/${1:-adfit}; mkdir -p $out; export TMPDIR=/tmp FAQCS_ABLATE_ADAPTER_FRAC=0
for cfg in "--polyA" "--adapter" "--adapter --polyA"; do
  tag=$(echo $cfg | tr -d ' -')
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_BRANCH --output-format csv -d $out/$tag -o pmc -- python3 tools/ablate.py 0 8e6 $cfg > $out/$tag.log 2>&1
  python3 - "$out/$tag" "$cfg" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "adapter_overlap" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[2], {c: round(v / n[c] / 8e6, 1) for c, v in sorted(acc.items())})
PY
  tail -1 $out/$tag.log
done
