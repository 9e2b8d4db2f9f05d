"""Diagnostic (not a test): faqcs_mi end to end on synthetic 2x150 FASTQ in /dev/shm for several parser / formatter / prefaulter thread
counts, forked and in one process, best of two runs each.  python tools/e2e_threads.py [pairs]   -> profiles/r5*/e2e_threads.txt"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_fixtures  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 14_300_000
L = 150
base = "/dev/shm/faqcs_e2e_threads"
os.makedirs(base, exist_ok=True)
blk = 500_000
seqs, quals = make_fixtures.headline_arrays(2 * blk, L)
paths = []
for mate in (1, 2):
    p = os.path.join(base, "r%d.fq" % mate)
    paths.append(p)
    with open(p, "wb") as f:
        done = 0
        while done < n:
            m = min(blk, n - done)
            head, tail = np.frombuffer(b"@SYN:", np.uint8), np.frombuffer(b"/%d\n" % mate, np.uint8)
            a = np.empty((m, len(head) + 9 + len(tail) + L + 3 + L + 1), np.uint8)
            c = 0
            a[:, c:c + len(head)] = head; c += len(head)
            ids = np.arange(done, done + m, dtype=np.int64)
            for k in range(9):
                a[:, c + 8 - k] = 48 + (ids // 10 ** k) % 10
            c += 9
            a[:, c:c + len(tail)] = tail; c += len(tail)
            lo = (mate - 1) * blk
            a[:, c:c + L] = seqs[lo:lo + m]; c += L
            a[:, c:c + 3] = np.frombuffer(b"\n+\n", np.uint8); c += 3
            a[:, c:c + L] = quals[lo:lo + m]; c += L
            a[:, c] = 10
            a.tofile(f)
            done += m
cli = os.path.join(ROOT, "faqcs_amd", "faqcs_mi")
print("host threads: %d; %d pairs 2x%d (%.1f GB in)" % (os.cpu_count(), n, L, 2 * os.path.getsize(paths[0]) / 1e9))
for par, fmt, pre in ((16, 16, 0), (16, 16, 16), (40, 40, 0), (40, 40, 16), (24, 40, 16), (40, 64, 32), (0, 0, -1)):
    for nofork in (0, 1):
        best, marks = None, ""
        for rep in range(2):
            out = os.path.join(base, "out")
            subprocess.run(["rm", "-rf", out])
            env = dict(os.environ, FAQCS_MI_TIMING="1")
            if par:
                env.update(FAQCS_MI_PARSERS=str(par), FAQCS_MI_FORMATTERS=str(fmt), FAQCS_MI_PREFAULTERS=str(pre))
            if nofork:
                env["FAQCS_MI_NO_FORK"] = "1"
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-1", paths[0], "-2", paths[1], "-d", out, "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"],
                               env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            if r.returncode == 0 and (best is None or dt < best):
                best = dt
                marks = " | ".join(ln.strip() for ln in r.stderr.decode(errors="replace").splitlines() if "parsers:" in ln or "first pair" in ln or "outputs written" in ln)
        label = "parsers %d formatters %d prefaulters %d" % (par, fmt, pre) if par else "defaults"
        print("%-46s %s: %.3f s = %5.1f M reads/s   %s" % (label, "one process" if nofork else "forked     ", best or -1, 2 * n / (best or 1e9) / 1e6, marks[:260]))
subprocess.run(["rm", "-rf", base])
