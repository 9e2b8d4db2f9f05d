"""Diagnostic: the owner-partitioned k-mer path inside ONE process (world = 1, the rank is its own owner): outbox -> insert_device -> finish.
Usage: python tools/owner_repro.py <n_reads> <maxlen> <subset> <split_size> <parts>"""
import os
import sys

import numpy as np

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in (here, os.path.join(here, "tests"), os.path.join(here, "tests", "golden")):
    sys.path.insert(0, d)
import torch  # noqa: E402

torch.cuda.init()
from oracle_engine import OracleEngine  # noqa: E402
from test_gpu_parity import random_batch  # noqa: E402

from faqcs_amd import driver, parallel  # noqa: E402
from faqcs_amd.engine import HipEngine  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

n_reads, maxlen, subset, split, parts = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", split, "--subset", subset])
rng = np.random.Generator(np.random.PCG64(4242))
reads = random_batch(rng, n_reads, maxlen, "adv")
segs = [reads[i:i + 333] for i in range(0, n_reads, 333)]
epochs, points = parallel.rarefaction_schedule([len(s) for s in segs], opt.split_size, opt.num_subsample)
eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 22)
n_epochs = opt.num_subsample + 1
eng.kmer_partition(0, 1, n_epochs)
cuts = [len(segs) * i // parts for i in range(parts + 1)]
for a, b in zip(cuts[:-1], cuts[1:]):
    seq, qual, offset, seg = driver.pack_segments(segs[a:b])
    eng.kmer_set_epochs(epochs[a:b])
    eng.process(seq, qual, offset, seg)
    ptr, counts = eng.kmer_outbox()
    n = int(counts.sum())
    got = torch.empty(2 * n, dtype=torch.int64, device="cuda")
    got.copy_(torch.as_tensor(parallel._DevArray(ptr, 2 * n), device="cuda"))
    torch.cuda.synchronize()
    eng.kmer_insert_device(got.data_ptr(), n)
eng.kmer_finish_pass()
d, t = eng.kmer_epoch_counts()
eng.kmer_end_table()
c, k = eng.kmer_histogram()
ora = OracleEngine(opt, 256, 33)
s2, q2, o2, g2 = driver.pack_segments(segs)
ora.process(s2, q2, o2, g2)
ora.kmer_end_table()
hc, hk = ora.kmer_histogram()
want = {int(a): int(b) for a, b in zip(hc, hk)}
mine = {int(a): int(b) for a, b in zip(c, k)}
diff = sorted((x, mine.get(x, 0), want.get(x, 0)) for x in set(mine) | set(want) if mine.get(x, 0) != want.get(x, 0))
pts = ora.kmer_points()
print("args", sys.argv[1:], "distinct", int(np.sum(d)), "total", int(np.sum(t)), "oracle last point", pts[-1] if len(pts) else None, "hist diff", diff[:8], "OK" if not diff else "MISMATCH")
