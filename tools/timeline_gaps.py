"""Diagnostic (not a test): where a step's wall time goes between kernels.  Reads a rocprofv3 --kernel-trace CSV
(<dir>/**/*kernel_trace.csv) and prints, for the last `steps` repetitions of the launch pattern, busy time per kernel and the idle gaps
on the device (time no kernel is running), largest first.
python tools/timeline_gaps.py <dir> [top]"""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:44]))
rows.sort()
t0 = rows[0][0]
# merge overlapping intervals to find device-idle gaps
gaps = []
busy_end = rows[0][1]
for i in range(1, len(rows)):
    s, e, k = rows[i]
    if s > busy_end:
        gaps.append((s - busy_end, busy_end - t0, rows[i - 1][2], k))
    busy_end = max(busy_end, e)
span = busy_end - t0
print("span %.1f ms, %d kernels, idle %.1f ms" % (span / 1e6, len(rows), sum(g[0] for g in gaps) / 1e6))
for g in sorted(gaps, reverse=True)[:top]:
    print("  gap %8.3f ms at %9.2f ms   after %-44s before %s" % (g[0] / 1e6, g[1] / 1e6, g[2], g[3]))
