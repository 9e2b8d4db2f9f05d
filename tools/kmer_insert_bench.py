"""Diagnostic: raw insert rate of the device k-mer table (faqcs_kmer_insert_device) vs table size and key reuse.
Usage: python tools/kmer_insert_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from faqcs_amd.engine import HipEngine  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction"])
dev = torch.device("cuda:0")
n = 200_000_000
for slots_log2, distinct in ((30, n), (27, 30_000_000), (30, 30_000_000), (24, 4_000_000), (30, 4_000_000), (30, 100_000)):
    eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << slots_log2)
    eng.kmer_partition(0, 1, 4)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    keys = torch.randint(0, distinct, (n,), device=dev, dtype=torch.int64, generator=g)
    keys = (keys * 0x9E3779B97F4A7C15 % (1 << 61)) & 0x7FFFFFFF7FFFFFFF  # spread; bits 31 and 63 clear like real keys
    items = torch.stack([keys, torch.zeros_like(keys)], dim=1).contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.kmer_insert_device(items.data_ptr(), n)
    dt = time.perf_counter() - t0
    d, t = eng.kmer_epoch_counts()
    print("table 2^%d slots (%5.1f GB), %9d distinct of %d inserts: %.2f G inserts/s (distinct counted %d)" % (
        slots_log2, 16 * 2.0 ** slots_log2 / 1e9, distinct, n, n / dt / 1e9, int(d.sum())))
    eng.close()
    del items, keys
