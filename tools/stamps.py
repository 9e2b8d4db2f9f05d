"""Diagnostic (not a test): section clocks of trim_lds from a -DFAQCS_LDS_STAMPS build of the library
(FAQCS_MI_LIB=profiles/microbench/libfaqcs_mi_stamps.so python tools/stamps.py [reads])."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
from faqcs_amd import _capi as capi  # noqa: E402
from faqcs_amd.engine import HipEngine, _check  # noqa: E402
from faqcs_amd.options import parse_args  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 16_000_000
L = 150
opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"] + sys.argv[2:])
eng = HipEngine(opt, 256, 33, device=0)
lib = eng.lib
dev = torch.device("cuda:0")
seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
off = torch.empty(n + 1, dtype=torch.int32, device=dev)
res = torch.empty((n, 4), dtype=torch.int16, device=dev)
_check(lib, lib.faqcs_synth_fill(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, 0.0))
seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
seg[-1] = n
b = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
w = np.zeros(16, np.uint64)
_check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
_check(lib, lib.faqcs_debug_words(eng.ctx, w.ctypes.data, 16))
for _ in range(3):
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
_check(lib, lib.faqcs_debug_words(eng.ctx, w.ctypes.data, 16))
names = ["load Q (offsets + DMA)", "terminal-N + sum pass", "3' walk", "5' walk + filters", "Q-B", "load S (DMA)", "S (fused pass)", "verdicts / dinucleotide", "undo / epilogue / flush"]
tot = float(w[:9].sum())
for i, nm in enumerate(names):
    print("%-26s %7.1f clocks/read-chunk-wave = %5.1f %%" % (nm, w[i] / (3 * n / 64), 100.0 * w[i] / tot))
print("total %.0f clocks per 64-read chunk per wave" % (tot / (3 * n / 64)))
