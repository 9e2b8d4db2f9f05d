#!/bin/bash
# HBM traffic of one trim-kernel launch, without Python in the profiled process: bash profiles/pmc_traffic2.sh <tag> [lib.so] [reads]
# Separate rocprofv3 --pmc passes (never combined with other trace domains) over profiles/microbench/trim_ab:
#   FETCH_SIZE ; WRITE_SIZE ; the request counters behind them, split by request width
# (MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B, i.e. a 128-byte request of a wide streaming read is tallied at 64 bytes;
#  the x2 correction therefore belongs to the requests that are NOT 32-byte requests only)
set -u
tag=${1:-traffic}; lib=${2:-faqcs_amd/libfaqcs_mi.so}; n=${3:-16777216}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    TRIM_AB_REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/t$i -o pmc -- ./profiles/microbench/trim_ab $n 150 1 $lib < /dev/null > $out/t$i.log 2>&1
done
python3 - "$out" "$n" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, n, tag = sys.argv[1], float(sys.argv[2]), sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in set(glob.glob(out + "/t*/**/*counter_collection.csv", recursive=True)) | set(glob.glob(out + "/t*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, c in acc.items():
    if not ("trim" in k or "fold" in k or "composition" in k):
        continue
    m = {name: sum(v) / len(v) for name, v in c.items()}
    d = {"launches_seen": max(len(v) for v in c.values()), "raw_counters_per_launch": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["fetch_bytes_x2"] = m["FETCH_SIZE"] * 1024 * 2
        d["write_bytes"] = m["WRITE_SIZE"] * 1024
    if "TCC_EA0_RDREQ_sum" in m:
        wide = m["TCC_EA0_RDREQ_sum"] - m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        d["fetch_bytes_by_request_width"] = wide * 128 + m.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32
        d["fetch_requests"] = {"not_32B": wide, "32B": m.get("TCC_EA0_RDREQ_32B_sum", 0.0)}
    if "TCC_EA0_WRREQ_sum" in m:
        w64 = m.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["write_bytes_by_request_width"] = w64 * 64 + (m["TCC_EA0_WRREQ_sum"] - w64) * 32
    res[k] = d
trim = next((k for k in res if k.startswith("trim")), None)
summary = {"tag": tag, "reads_per_launch": n, "kernels": res}
if trim:
    t = res[trim]
    fb = t.get("fetch_bytes_by_request_width", t.get("fetch_bytes_x2", 0.0))
    wb = t.get("write_bytes", t.get("write_bytes_by_request_width", 0.0))
    summary.update({"kernel": trim.split("<")[0], "kernel_full": trim, "hbm_bytes_per_launch": fb + wb, "hbm_bytes_per_read": (fb + wb) / n,
                    "fetch_bytes_per_read": fb / n, "write_bytes_per_read": wb / n, "fetch_bytes_per_read_x2_rule": t.get("fetch_bytes_x2", 0.0) / n,
                    "algorithmic_bytes_per_read": 312,
                    "note": "rocprofv3 --pmc in separate passes over profiles/microbench/trim_ab (no other trace domain); fetch = (RDREQ - RDREQ_32B) x 128 B + RDREQ_32B x 32 B "
                            "(the x2 rule of MI355X_MICROARCH.md applied to the wide requests only), write = WRITE_SIZE"})
print(json.dumps(summary, indent=1))
PY
