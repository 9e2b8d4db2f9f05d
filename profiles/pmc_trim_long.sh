#!/bin/bash
# Counters of trim_long / long_accumulate on reads of 10 000 bases: bash profiles/pmc_trim_long.sh <tag>
# (rocprofv3 --pmc over bench.py --read-len 10000 --pairs 4e5: one launch = 429 496 reads = 4.29 G positions)
set -u
tag=${1:-trim_long}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python3 bench.py --read-len 10000 --pairs 1e6 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_10000bp_1Mpairs_trim_long.json 2>> $out/bench.err
python3 bench.py --read-len 2000 --pairs 5e6 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_2000bp_5Mpairs_trim_long.json 2>> $out/bench.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH --output-format csv -d $out/pmc -o pmc -- python3 bench.py --read-len 10000 --pairs 4e5 --steps 2 --no-cpu-baseline --e2e-pairs 0 > $out/pmc.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(list)
def name(k):
    m = re.search(r"(trim_long|long_accumulate|composition_\w+|synth_fill)", k)
    return m.group(1) if m else None
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = name(r["Kernel_Name"])
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for f in glob.glob(out + "/pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = name(r["Kernel_Name"])
        if k: dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
pos = 429496 * 10000.0
print("# bench.py --read-len 10000 --pairs 4e5: one launch = 429 496 reads of 10 000 bases = 4.29 G positions; counters per 64 positions")
for k in ("trim_long", "long_accumulate"):
    if k in acc:
        print(k, "launches", len(dur[k]), "avg %.2f ms (under the profiler)" % (sum(dur[k]) / max(1, len(dur[k]))))
        for c, v in sorted(acc[k].items()):
            print("   %-20s %10.2f per 64 positions" % (c, v / n[(k, c)] / pos * 64))
for f in sorted(glob.glob(out + "/bench_plain_*.json")):
    import json
    try:
        j = json.load(open(f)); print(f.split("/")[-1], j["value"], "M reads/s", j["roofline"]["kernels_ms"])
    except Exception as e:
        print(f, "unreadable", e)
PY
