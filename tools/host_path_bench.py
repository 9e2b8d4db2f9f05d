"""Diagnostic: PCIe-inclusive rate of the host-buffer path (faqcs_submit_async from pinned arenas, results copied back).
Usage: python tools/host_path_bench.py [reads per batch] [batches]"""
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

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = 150
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "-q", "5", "--min_L", "50"])
eng = HipEngine(opt, 256, 33, device=0)
lib = eng.lib
dev = torch.device("cuda:0")
seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
off = torch.empty(n + 1, dtype=torch.int32, device=dev)
_check(lib, lib.faqcs_synth_fill(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, 0.0))


def pinned(nbytes):
    p = lib.faqcs_host_alloc(nbytes)
    assert p
    return p, (C.c_uint8 * nbytes).from_address(p)


bufs = []
for _ in range(2):
    ps, vs = pinned(n * L + 64)
    pq, vq = pinned(n * L + 64)
    po, vo = pinned(4 * (n + 1))
    pr, vr = pinned(8 * n)
    np.frombuffer(vs, np.uint8)[:] = seq.cpu().numpy()
    np.frombuffer(vq, np.uint8)[:] = qual.cpu().numpy()
    np.frombuffer(vo, np.uint8)[:] = off.cpu().numpy().view(np.uint8)
    bufs.append((ps, pq, po, pr))
seg = np.array([0, n], dtype=np.uint32)
tick = C.c_uint64()
t0 = time.perf_counter()
tickets = []
for i in range(nb):
    ps, pq, po, pr = bufs[i & 1]
    if i >= 2:
        _check(lib, lib.faqcs_wait(eng.ctx, tickets[i - 2]))  # the buffer pair is free again
    b = capi.Batch(ps, pq, po, n, 1, seg.ctypes.data, L)
    _check(lib, lib.faqcs_submit_async(eng.ctx, C.byref(b), pr, C.byref(tick)))
    tickets.append(tick.value)
eng.sync()
dt = time.perf_counter() - t0
print("host-buffer path: %d batches x %d reads of %d bp from pinned memory in %.1f ms -> %.1f M reads/s, %.1f GB/s over PCIe (in) + %.1f GB/s (results out)" % (
    nb, n, L, dt * 1e3, nb * n / dt / 1e6, nb * n * (2 * L + 4) / dt / 1e9, nb * n * 8 / dt / 1e9))
