"""Diagnostic (run by tests/test_gpu_slot_rule.py with --quick): with a library built by `bash profiles/build_variant.sh qbchk -DFAQCS_LDS_DIAG_CHECK_QB_ADDR` (FAQCS_MI_LIB=...), count
the Q-B adds of trim_lds -- zero increments included -- whose address lies outside the position x quality matrix (the slot's
validity rule) over batches of every lane geometry: adversarial and ragged reads, equal-length batches with most reads taken back, padded rows,
adapters.  python tools/qb_rule_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, d)
import make_fixtures  # noqa: E402

from faqcs_amd import driver  # noqa: E402
from faqcs_amd.engine import HipEngine  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

quick = "--quick" in sys.argv  # (tests/test_gpu_slot_rule.py: every lane geometry, fewer option sets and reads)
seed = int(os.environ.get("FAQCS_TEST_SEED", "0"))
rng = np.random.Generator(np.random.PCG64(2026 + seed))
total_bad = total_steps = 0
for L in ((50, 75, 128, 150, 224, 250, 300) if quick else (36, 50, 64, 75, 100, 128, 150, 152, 160, 200, 224, 250, 288, 300)):
    for args in (([], ["--avg_q", "20", "-n", "1"]) if quick else
                 ([], ["--avg_q", "20", "-n", "1"], ["--adapter", "--polyA", "--min_L", "20"], ["--mode", "HARD", "-q", "10", "--5end", "3"])):
        for kind in ("adv", "equal", "ragged"):
            reads = []
            for i in range(3000 if quick else 6000):
                if kind == "adv":
                    s, q = make_fixtures._adv_read(rng, L)
                elif kind == "equal":
                    s = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, L, p=[.2, .2, .2, .2, .02, .045, .045, .045, .045])].copy()
                    q = (rng.integers(28, 41, L) + 33).astype(np.uint8)
                else:
                    l = int(rng.integers(0, L + 1))
                    s = np.frombuffer(b"ACGTNacgtnRY", np.uint8)[rng.integers(0, 12, l)].copy()
                    q = (rng.integers(0, 42, l) + 33).astype(np.uint8)
                reads.append((b"@x", s.tobytes(), q.tobytes()))
            opt = parse_args(["-u", "x", "-d", "y"] + args)
            eng = HipEngine(opt, 320 if L > 256 else 256, 33, device=0)
            seq, qual, offset, seg = driver.pack_segments([reads[i:i + 1500] for i in range(0, len(reads), 1500)])
            eng.process(seq, qual, offset, seg)
            eng.counters()
            dbg = (C.c_uint64 * 6)()
            eng.lib.faqcs_debug_words(eng.ctx, dbg, 6)
            bad, steps = int(dbg[3]), int(dbg[4])
            total_bad += bad
            total_steps += steps
            if bad:
                print("L %d %s %s: %d adds outside the matrix (%d steps checked)" % (L, kind, " ".join(args) or "default", bad, steps))
print("adds outside the quality matrix: %d ; lane-steps checked: %d" % (total_bad, total_steps))
sys.exit(0 if total_bad == 0 and total_steps > 0 else 1)
