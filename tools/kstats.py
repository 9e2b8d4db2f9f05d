"""Prints a rocprofv3 --stats kernel_stats.csv as name / calls / total ms / average ms.  Usage: python tools/kstats.py <dir or csv> [rows]"""
import csv
import glob
import os
import sys

p = sys.argv[1]
if os.path.isdir(p):
    p = (glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True) or [""])[0]
if not p:
    sys.exit("no kernel_stats.csv")
for i, r in enumerate(csv.DictReader(open(p))):
    if i >= int(sys.argv[2]) if len(sys.argv) > 2 else 16:
        break
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    print("%-44s %5s calls %10.3f ms total %9.3f ms avg" % (name[:44], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
