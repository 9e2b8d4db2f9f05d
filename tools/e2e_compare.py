"""Diagnostic (not a test): whole-process wall clock of the reference binary vs the reference driver linked against
libfaqcs_mi.so (integration/trim_shim.cpp) on the same FASTQ pair.  Usage: python tools/e2e_compare.py [pairs] [threads]"""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import make_fixtures  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500000
threads = sys.argv[2] if len(sys.argv) > 2 else "8"
tmp = tempfile.mkdtemp(prefix="faqcs_e2e_")
for mate in (1, 2):
    s, q = make_fixtures.headline_arrays(n, 150, mate=mate)
    with open(os.path.join(tmp, "r%d.fq" % mate), "wb") as f:
        f.write(b"".join(b"@SYN:%d/%d\n" % (i, mate) + s[i].tobytes() + b"\n+\n" + q[i].tobytes() + b"\n" for i in range(n)))
BIN = {"FaQCs_ref": os.path.join(ROOT, "oracle", "_ref", "FaQCs_ref"), "FaQCs_hip": os.path.join(ROOT, "oracle", "_ref", "FaQCs_hip"),
       "faqcs_mi": os.path.join(ROOT, "faqcs_amd", "faqcs_mi")}
for name in ("FaQCs_ref", "FaQCs_hip", "faqcs_mi"):
    for extra in ([], ["--adapter", "--polyA"]):
        out = os.path.join(tmp, name + "_" + str(len(extra)))
        cmd = [BIN[name], "-1", os.path.join(tmp, "r1.fq"), "-2", os.path.join(tmp, "r2.fq"),
               "-d", out, "-t", threads, "--ascii", "33", "--trim_only"] + extra
        t0 = time.perf_counter()
        rc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode
        dt = time.perf_counter() - t0
        print("%-10s %-18s rc=%d wall %.2f s -> %.3f M reads/s" % (name, " ".join(extra) or "default", rc, dt, 2 * n / dt / 1e6))
same = subprocess.run(["cmp", os.path.join(tmp, "FaQCs_ref_0", "QC.stats.txt"), os.path.join(tmp, "FaQCs_hip_0", "QC.stats.txt")]).returncode
same2 = subprocess.run(["cmp", os.path.join(tmp, "FaQCs_ref_2", "QC.1.trimmed.fastq"), os.path.join(tmp, "FaQCs_hip_2", "QC.1.trimmed.fastq")]).returncode
same3 = subprocess.run(["cmp", os.path.join(tmp, "FaQCs_ref_2", "QC.2.trimmed.fastq"), os.path.join(tmp, "faqcs_mi_2", "QC.2.trimmed.fastq")]).returncode
same4 = subprocess.run(["cmp", os.path.join(tmp, "FaQCs_ref_2", "QC.stats.txt"), os.path.join(tmp, "faqcs_mi_2", "QC.stats.txt")]).returncode
print("stats identical:", same == 0, " adapter-run trimmed FASTQ identical:", same2 == 0, " faqcs_mi adapter run identical:", same3 == 0 and same4 == 0)
subprocess.run(["rm", "-rf", tmp])
