#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of profiles/collect.sh into the per-launch HBM
traffic of the trim kernel (trim_lds / trim_tpr / trim_filter_accumulate), with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE
is in KB and tallies 128-byte requests as 64 bytes -> x2; WRITE_SIZE in KB as is).
Usage: pmc_traffic.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <reads per launch>"""
import csv
import glob
import json
import os
import sys

d, reads = sys.argv[1], float(sys.argv[2])
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.path.normpath(d))
raw = {}
name = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in ("trim_filter_accumulate", "trim_tpr", "trim_lds")):
                if row["Counter_Name"] == c:
                    vals.append(float(row["Counter_Value"]))
                    name = row["Kernel_Name"]
    raw[c] = sum(vals) / max(1, len(vals))
    raw[c + "_launches"] = len(vals)
fetch = raw["FETCH_SIZE"] * 1024 * 2
write = raw["WRITE_SIZE"] * 1024
print(json.dumps({
    "kernel": (name or "").split("<")[0].replace("void ", ""), "kernel_full": name, "tag": tag, "reads_per_launch": reads, "raw_counters_per_launch": raw,
    "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
    "hbm_bytes_per_read": (fetch + write) / reads, "algorithmic_bytes_per_read": 312,
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 tools/ablate.py 0 16e6`; FETCH_SIZE (KB) "
            "doubled per the gfx950 correction in MI355X_MICROARCH.md (128-B requests tallied at 64 B)"}, indent=1))
