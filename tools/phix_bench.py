"""Diagnostic: adapter pre-pass on reads that ARE PhiX (stage 2 runs for every read).  Usage: python tools/phix_bench.py [reads] [hit fraction]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch  # noqa: E402

torch.cuda.init()
import make_fixtures  # noqa: E402

from faqcs_amd import driver, options  # noqa: E402
from faqcs_amd.engine import HipEngine  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
L = 150
rng = np.random.Generator(np.random.PCG64(3))
phix = np.frombuffer(options.phix_sequence().encode(), np.uint8)
seq = make_fixtures.ACGT[rng.integers(0, 4, (n, L))]
hit = rng.random(n) < frac
p0 = rng.integers(0, len(phix) - L, n)
idx = p0[:, None] + np.arange(L)[None, :]
seq[hit] = phix[idx[hit]]
mut = rng.random((n, L)) < 0.02
seq[mut] = make_fixtures.ACGT[rng.integers(0, 4, int(mut.sum()))]
qual = np.full((n, L), 33 + 38, np.uint8)
off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
seg[-1] = n
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "--phiX"])
eng = HipEngine(opt, 256, 33, device=0)
s1, q1 = np.concatenate([seq.ravel(), np.zeros(64, np.uint8)]), np.concatenate([qual.ravel(), np.zeros(64, np.uint8)])
eng.process(s1, q1, off, seg)
t0 = time.perf_counter()
res = eng.process(s1, q1, off, seg)
dt = time.perf_counter() - t0
print("phiX, %.0f %% hits: %d reads in %.1f ms -> %.2f M reads/s (host submission), %d reads flagged" % (
    100 * frac, n, dt * 1e3, n / dt / 1e6, int(((res["flags"] & 0x20) != 0).sum())))
