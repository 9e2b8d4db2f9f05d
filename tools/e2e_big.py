"""Diagnostic (not a test): end-to-end timing of faqcs_mi on synthetic 2x150 FASTQ files in /dev/shm with stage marks.
python tools/e2e_big.py [pairs] [extra faqcs_mi args...]      FAQCS_E2E_GZ=1: also as bgzip (BGZF) and as plain gzip input"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_fixtures  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
L = 150
base = "/dev/shm/faqcs_e2e_big"
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
            idw = 9
            head = np.frombuffer(b"@SYN:", np.uint8)
            tail = np.frombuffer(b"/%d\n" % mate, np.uint8)
            rec = len(head) + idw + len(tail) + L + 3 + L + 1
            a = np.empty((m, rec), np.uint8)
            c = 0
            a[:, c:c + len(head)] = head; c += len(head)
            ids = np.arange(done, done + m, dtype=np.int64)
            for k in range(idw):
                a[:, c + idw - 1 - k] = 48 + (ids // 10 ** k) % 10
            c += idw
            a[:, c:c + len(tail)] = tail; c += len(tail)
            lo = (mate - 1) * blk
            a[:, c:c + L] = seqs[lo:lo + m]; c += L
            a[:, c:c + 3] = np.frombuffer(b"\n+\n", np.uint8); c += 3
            a[:, c:c + L] = quals[lo:lo + m]; c += L
            a[:, c] = 10
            a.tofile(f)
            done += m
cli = os.path.join(ROOT, "faqcs_amd", "faqcs_mi")
for env in ({}, {"FAQCS_MI_STREAMING": "1"}):
    out = os.path.join(base, "out")
    subprocess.run(["rm", "-rf", out])
    t0 = time.perf_counter()
    r = subprocess.run([cli, "-1", paths[0], "-2", paths[1], "-d", out, "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + sys.argv[2:],
                       env=dict(os.environ, FAQCS_MI_TIMING="1", **env), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    print("%s: %d pairs in %.3f s = %.1f M reads/s (rc %d)" % ("streaming" if env else "mapped", n, dt, 2 * n / dt / 1e6, r.returncode))
    print(r.stderr.decode(errors="replace"))
if os.environ.get("FAQCS_E2E_GZ"):
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    def member(raw):  # one BGZF member (zlib releases the GIL: the pool compresses in parallel)
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        body = c.compress(raw) + c.flush()
        return struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw))

    gz_sets = {"bgzf": [], "gzip": []}
    with ThreadPoolExecutor(32) as pool:
        for p in paths:
            data = open(p, "rb").read()
            blocks = [data[o:o + 65280] for o in range(0, len(data), 65280)] + [b""]
            with open(p + ".bgzf.gz", "wb") as f:
                for m in pool.map(member, blocks, chunksize=64):
                    f.write(m)
            gz_sets["bgzf"].append(p + ".bgzf.gz")
            c = zlib.compressobj(1, zlib.DEFLATED, 31)
            with open(p + ".plain.gz", "wb") as f:
                for o in range(0, len(data), 1 << 24):
                    f.write(c.compress(data[o:o + (1 << 24)]))
                f.write(c.flush())
            gz_sets["gzip"].append(p + ".plain.gz")
    import resource
    variants = {"bgzf": [("", {}), (" (members through zlib: FAQCS_MI_BGZF_ZLIB=1)", {"FAQCS_MI_BGZF_ZLIB": "1"})],
                "gzip": [("", {})] + [(" (%s)" % v, dict(kv.split("=") for kv in v.split())) for v in os.environ.get("FAQCS_E2E_GZ_TRY", "").split(";") if v] +
                        [(" (round 5's two zlib passes per piece: FAQCS_MI_PARGZ_TWO_PASS=1)", {"FAQCS_MI_PARGZ_TWO_PASS": "1"}),
                         (" (through gzread, one thread per file: FAQCS_MI_NO_PARGZ=1)", {"FAQCS_MI_NO_PARGZ": "1"})]}
    for tag, gp in gz_sets.items():
        for what, extra in variants[tag]:
            for rep in range(1 if "NO_PARGZ" in "".join(extra) else 2):
                out = os.path.join(base, "out")
                subprocess.run(["rm", "-rf", out])
                t0 = time.perf_counter()
                r = subprocess.run([cli, "-1", gp[0], "-2", gp[1], "-d", out, "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + sys.argv[2:],
                                   env=dict(os.environ, FAQCS_MI_TIMING="1", **extra), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                dt = time.perf_counter() - t0
                print("%s input%s (%.2f GB compressed per file): %d pairs in %.3f s = %.1f M reads/s (rc %d)" % (tag, what, os.path.getsize(gp[0]) / 1e9, n, dt, 2 * n / dt / 1e6, r.returncode))
            for line in r.stderr.decode(errors="replace").splitlines():  # (the last repeat's threads, by role: FAQCS_MI_TIMING)
                if "threads '" in line or "main thread:" in line or "parsers:" in line or (os.environ.get("FAQCS_E2E_MARKS") and "[faqcs_mi" in line):
                    print("    " + line)
            # the same in ONE process, so that its CPU time and page faults can be read (the default mode leaves them in a detached worker)
            subprocess.run(["rm", "-rf", out])
            r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-1", gp[0], "-2", gp[1], "-d", out, "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + sys.argv[2:],
                               env=dict(os.environ, FAQCS_MI_NO_FORK="1", **extra), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
            print("%s input%s, one process: %.3f s wall, %.1f s user, %.1f s system, %d minor faults (rc %d)"
                  % (tag, what, dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime, r1.ru_minflt - r0.ru_minflt, r.returncode))
if os.environ.get("FAQCS_E2E_GZ") and os.environ.get("FAQCS_E2E_UNPAIRED"):  # one file as unpaired input (process_unpaired's streaming path)
    for tag, gp in gz_sets.items():
        for rep in range(2):
            out = os.path.join(base, "out")
            subprocess.run(["rm", "-rf", out])
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-u", gp[0], "-d", out, "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + sys.argv[2:],
                               env=dict(os.environ, FAQCS_MI_TIMING="1"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            print("%s input, ONE file as unpaired reads: %d reads in %.3f s = %.1f M reads/s (rc %d)" % (tag, n, dt, n / dt / 1e6, r.returncode))
subprocess.run(["rm", "-rf", base])
