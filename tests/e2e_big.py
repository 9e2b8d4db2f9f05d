"""Diagnostic: steady-state end-to-end throughput of faqcs_amd/faqcs_mi on a larger uncompressed FASTQ pair.
Usage: python tests/e2e_big.py [pairs] [extra faqcs flags...]"""
import os
import resource
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_fixtures  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
extra = sys.argv[2:]
tmp = tempfile.mkdtemp(prefix="faqcs_big_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
L = 150
t0 = time.perf_counter()
for mate in (1, 2):
    s, q = make_fixtures.headline_arrays(n, L, mate=mate)
    ids = np.char.add(np.char.add("@SYN:", np.char.zfill(np.arange(n).astype(str), 9)), "/%d" % mate).astype("S16")
    rec = np.empty((n, 16 + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :16] = ids.view(np.uint8).reshape(n, 16)
    rec[:, 16] = 10
    rec[:, 17:17 + L] = s
    rec[:, 17 + L:20 + L] = np.frombuffer(b"\n+\n", np.uint8)
    rec[:, 20 + L:20 + 2 * L] = q
    rec[:, 20 + 2 * L] = 10
    rec.tofile(os.path.join(tmp, "r%d.fq" % mate))
    del s, q, rec, ids
print("generated %d pairs in %.1f s (%s)" % (n, time.perf_counter() - t0, tmp))
cmd = [os.path.join(ROOT, "faqcs_amd", "faqcs_mi"), "-1", os.path.join(tmp, "r1.fq"), "-2", os.path.join(tmp, "r2.fq"), "-d",
       os.path.join(tmp, "out"), "--ascii", "33", "--trim_only"] + extra
for rep in range(2):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    rc = subprocess.run(cmd, stderr=subprocess.DEVNULL).returncode
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    print("faqcs_mi %s rc=%d wall %.2f s (user %.1f sys %.1f) -> %.2f M reads/s end to end, %.2f GB/s of FASTQ text" % (
        " ".join(extra), rc, dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, 2 * n / dt / 1e6,
        2 * n * (37 + 2 * L) / dt / 1e9))
print(open(os.path.join(tmp, "out", "QC.stats.txt")).read()[:300])
subprocess.run(["rm", "-rf", tmp])
