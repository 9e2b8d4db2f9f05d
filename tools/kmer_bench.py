"""Diagnostic: throughput of the k-mer path (trim + kmer_count).  Usage: python tools/kmer_bench.py [reads] [L] [log2 slots] [genome length]"""
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
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "--kmer_rarefaction", "--split_size", "1000000", "--subset", "100000"])
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 30
eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << slots)
lib = eng.lib
dev = torch.device("cuda:0")
seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
off = torch.empty(n + 1, dtype=torch.int32, device=dev)
res = torch.empty((n, 4), dtype=torch.int16, device=dev)
genome = int(float(sys.argv[4])) if len(sys.argv) > 4 else 0  # 0 = all-random reads (worst case), else genome length
if genome:
    _check(lib, lib.faqcs_synth_fill_genome(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, genome))
else:
    _check(lib, lib.faqcs_synth_fill(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, 0.0))
seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
seg[-1] = n
b = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
t0 = time.perf_counter()
_check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
if os.environ.get("KMER_BENCH_TABLE") == "1":  # the curve so far: the open group goes into the table (round 5's path)
    d, t = eng.kmer_totals()
eng.kmer_end_table()  # the pass ends: counted in one piece
dt = time.perf_counter() - t0
d, t = eng.kmer_totals()
print("k-mer path: %d reads x %d bp in %.1f ms -> %.1f M reads/s, %.2f G k-mer inserts/s (distinct %d, total %d, points %d)" % (
    n, L, dt * 1e3, n / dt / 1e6, t / dt / 1e9, d, t, len(eng.kmer_points())))
