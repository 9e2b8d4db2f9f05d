"""Diagnostic (not a test): repeat one equal-length batch of test_gpu_parity (L, option set, seed) and list EVERY counter that differs
from the oracle.  python tools/flaky_probe.py <L> <reps> <seed> [faqcs options...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, d)
from oracle_engine import OracleEngine  # noqa: E402

from faqcs_amd import _capi as capi  # noqa: E402
from faqcs_amd import driver  # noqa: E402
from faqcs_amd.engine import HipEngine  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

L, reps, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
args = sys.argv[4:]
rng = np.random.Generator(np.random.PCG64([91, L, seed]))
opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1"] + args)
reads = []
for i in range(900):
    s = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, L, p=[.235, .235, .235, .235, .02, .01, .01, .01, .01])].copy()
    q = (rng.integers(28, 41, L) + 33).astype(np.uint8)
    head = int(rng.integers(0, min(L, 12))) if i % 3 else 0
    q[:head] = 33 + rng.integers(0, 4, head)
    if i % 7 == 0 and L > 20:
        q[L - int(rng.integers(1, 9)):] = 34
    reads.append((b"@x", s.tobytes(), q.tobytes()))
bufs = [reads[i:i + 300] for i in range(0, len(reads), 300)]
seq, qual, offset, seg = driver.pack_segments(bufs)
ora = OracleEngine(opt, 256, 33)
r2 = ora.process(seq, qual, offset, seg)
c2 = ora.counters()
lay = capi.python_layout(256, 0)
bad_runs = 0
for rep in range(reps):
    hip = HipEngine(opt, 256, 33, device=0)
    r1 = hip.process(seq, qual, offset, seg)
    c1 = hip.counters()
    nb = np.nonzero(r1 != r2)[0]
    ks = np.nonzero(c1 != c2)[0]
    if len(nb) or len(ks):
        bad_runs += 1
        print("rep %d: %d results differ, %d counters differ" % (rep, len(nb), len(ks)))
        for k in ks[:12]:
            name = [nm for nm, v in lay.items() if nm != "total" and v[0] <= k < v[0] + v[1]][0]
            j = int(k - lay[name][0])
            extra = " (pos %d, q %d)" % (j // 42, j % 42) if name.endswith("_qual") else ""
            print("    %s[%d]%s hip=%d oracle=%d" % (name, j, extra, c1[k], c2[k]))
    hip.close() if hasattr(hip, "close") else None
print("bad runs: %d of %d" % (bad_runs, reps))
