"""Diagnostic: owner-partitioned k-mer path on ONE rank (world = 1: every bucket is local): extraction + insertion rates.
Usage: python tools/kmer_exchange_bench.py [reads] [L] [world for bucketing only]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from faqcs_amd import _capi as capi  # noqa: E402
from faqcs_amd.engine import HipEngine, _check  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 250
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "--kmer_rarefaction", "--split_size", "1000000", "--subset", "1000"])
eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 30)
eng.kmer_partition(0, world, 1001)
lib = eng.lib
dev = torch.device("cuda:0")
seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
off = torch.empty(n + 1, dtype=torch.int32, device=dev)
res = torch.empty((n, 4), dtype=torch.int16, device=dev)
_check(lib, lib.faqcs_synth_fill_genome(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, 50_000_000))
seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
seg[-1] = n
b = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
for _ in range(2):  # the first submission pays for code-object loading and the outbox allocation
    eng.kmer_set_epochs(np.minimum(np.arange(len(seg) - 1) // 31, 1000).astype(np.uint32))
    t0 = time.perf_counter()
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    ptr, counts = eng.kmer_outbox()
    t1 = time.perf_counter()
tot = int(counts.sum())
if world == 1:
    eng.kmer_insert_device(ptr, tot)
t2 = time.perf_counter()
print("trim + extract (count, prefix, fill) of %d k-mers into %d bucket(s): %.1f ms -> %.2f G k-mers/s; insert: %.1f ms -> %.2f G inserts/s" % (
    tot, world, (t1 - t0) * 1e3, tot / (t1 - t0) / 1e9, (t2 - t1) * 1e3, tot / max(t2 - t1, 1e-9) / 1e9))
