"""Diagnostic (not a test): times trim_filter_accumulate with FAQCS_DBG ablation bits, one process per setting
so every run builds its own context.  Usage: python tools/ablate.py <dbg> [pairs]"""
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

dbg = sys.argv[1]
os.environ["FAQCS_DBG"] = dbg
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 8_000_000
extra = sys.argv[3:]
L = int(__import__("os").environ.get("FAQCS_ABLATE_L", "150"))
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"] + extra)
eng = HipEngine(opt, 256 if L <= 256 else 1024, 33, device=0)
lib = eng.lib
dev = torch.device("cuda:0")
seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
off = torch.empty(n + 1, dtype=torch.int32, device=dev)
res = torch.empty((n, 4), dtype=torch.int16, device=dev)
_check(lib, lib.faqcs_synth_fill(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, float(os.environ.get("FAQCS_ABLATE_ADAPTER_FRAC", "0.05")) if extra else 0.0))
seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
seg[-1] = n
b = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
for _ in range(2):
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
eng.sync()
ms, nl = C.c_double(), C.c_uint64()
lib.faqcs_kernel_time_ms(eng.ctx, C.byref(ms), C.byref(nl))
t0 = time.perf_counter()
for _ in range(5):
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
eng.sync()
dt = (time.perf_counter() - t0) / 5
lib.faqcs_kernel_time_ms(eng.ctx, C.byref(ms), C.byref(nl))
kt = capi.KernelTimes()
print("dbg=%s %s: wall %.3f ms/pass, trim kernel %.3f ms -> %.1f M reads/s (kernel), %.1f GB/s algorithmic" % (
    dbg, " ".join(extra), dt * 1e3, ms.value, n / ms.value / 1e3, n * 312 / ms.value / 1e6))
if extra:
    for _ in range(2):
        _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    lib.faqcs_kernel_report(eng.ctx, C.byref(kt))
    print("adapter_overlap %.3f ms -> %.1f M reads/s (kernel)" % (kt.adapter_ms, n / max(kt.adapter_ms, 1e-9) / 1e3))
