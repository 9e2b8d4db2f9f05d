"""Developer tool (GPU box): rebuild the batch of a failing seeded parity test and shrink it to a small failing batch.

    python tools/fuzz_shrink.py width <seed> <maxlen> <kind> <option-set index>
    python tools/fuzz_shrink.py qoff  <seed> <maxlen> <kind> <in_off>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

which, seed, maxlen, kind, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
os.environ["FAQCS_TEST_SEED"] = str(seed)
import test_gpu_parity as T  # noqa: E402
from faqcs_amd import _capi as capi  # noqa: E402
from faqcs_amd import driver  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402
from oracle_engine import OracleEngine  # noqa: E402

if which == "width":
    args = T.OPTION_SETS[last]
    rng = np.random.Generator(np.random.PCG64([3, len(kind), maxlen, last, seed]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    reads = T.random_batch(rng, 500 if "--adapter" in args else 1500, maxlen, kind)
    in_off, seg_size = 33, 389
else:
    rng = np.random.Generator(np.random.PCG64([11, last, maxlen, seed]))
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", str(last), "--min_L", "20"])
    reads = T.random_batch(rng, 1200, maxlen, kind)
    in_off, seg_size = last, 500


def diff(reads, seg_size, verbose=False):
    bufs = [reads[i:i + seg_size] for i in range(0, len(reads), seg_size)] or [[]]
    seq, qual, offset, seg = driver.pack_segments(bufs)
    hip = T.hip_factory(opt, 256, in_off)
    ora = OracleEngine(opt, 256, in_off)
    r1 = hip.process(seq, qual, offset, seg)
    r2 = ora.process(seq, qual, offset, seg)
    c1, c2 = hip.counters(), ora.counters()
    bad_r = np.nonzero(r1 != r2)[0]
    bad_c = np.nonzero(c1 != c2)[0]
    if verbose:
        lay = capi.python_layout(256, hip.holder.n_adapters)
        for i in bad_r[:5]:
            print("read", i, "hip", r1[i], "oracle", r2[i], reads[i][1], reads[i][2])
        for k in bad_c[:40]:
            name = [nm for nm, v in lay.items() if nm != "total" and v[0] <= k < v[0] + v[1]][0]
            print("counter %s[%d]: hip=%d oracle=%d" % (name, k - lay[name][0], c1[k], c2[k]))
    return len(bad_r) + len(bad_c) > 0


print("full batch fails:", diff(reads, seg_size, True))
cur = list(reads)
S = (1 << 30) if diff(reads, 1 << 30) else seg_size
print("shrinking with segment size", S)
# drop whole prefixes / suffixes, then single reads (keeps neighbours that share a wave)
step = len(cur) // 2
while step >= 1:
    changed = True
    while changed and len(cur) > step:
        changed = False
        if diff(cur[:-step], S):
            cur, changed = cur[:-step], True
        elif diff(cur[step:], S):
            cur, changed = cur[step:], True
    step //= 2
print("shrunk to", len(cur), "reads (one segment)")
i = 0
while i < len(cur) and len(cur) > 1:
    trial = cur[:i] + cur[i + 1:]
    if diff(trial, S):
        cur = trial
    else:
        i += 1
print("minimal:", len(cur))
diff(cur, S, True)
for r in cur[:20]:
    print(len(r[1]), r[1], r[2])
