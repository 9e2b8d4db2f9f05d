"""OracleEngine -- TEST INFRASTRUCTURE.  Same surface as faqcs_amd.engine.HipEngine but backed by the
plain-C CPU restatement in oracle/ (the checker).  Lives under tests/ so the product package can never
import it."""
import ctypes as C
import os
import subprocess

import numpy as np

from faqcs_amd import _capi as capi
from faqcs_amd.engine import FaqcsError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libfaqcs_oracle.so")
_LIB = None


def load_oracle():
    global _LIB
    if _LIB is not None:
        return _LIB
    src = os.path.join(ROOT, "oracle", "faqcs_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    lib = C.CDLL(ORACLE_SO)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    lib.faqcs_oracle_create.restype = vp
    lib.faqcs_oracle_create.argtypes = [C.POINTER(capi.Params)]
    lib.faqcs_oracle_destroy.argtypes = [vp]
    lib.faqcs_oracle_trim.restype = i32
    lib.faqcs_oracle_trim.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    lib.faqcs_oracle_kmer_active.restype = i32
    lib.faqcs_oracle_kmer_active.argtypes = [vp]
    lib.faqcs_oracle_kmer_points.restype = u32
    lib.faqcs_oracle_kmer_points.argtypes = [vp, vp, u32]
    lib.faqcs_oracle_kmer_totals.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    lib.faqcs_oracle_kmer_end_table.argtypes = [vp, u64]
    lib.faqcs_oracle_kmer_histogram.restype = u64
    lib.faqcs_oracle_kmer_histogram.argtypes = [vp, vp, vp, u64]
    lib.faqcs_oracle_quality_trim.restype = u32
    lib.faqcs_oracle_quality_trim.argtypes = [i32, vp, u32, i32, i32, i32, C.POINTER(u32)]
    lib.faqcs_oracle_align.restype = i32
    lib.faqcs_oracle_align.argtypes = [vp, u32, vp, u32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.faqcs_oracle_find_mask_range.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    _LIB = lib
    return lib


class OracleEngine:
    name = "oracle"

    def __init__(self, opt, max_read_length, input_quality_offset=None, **_):
        self.lib = load_oracle()
        self.holder = capi.ParamsHolder(opt, max_read_length, input_quality_offset)
        self.o = self.lib.faqcs_oracle_create(C.byref(self.holder.p))
        assert self.o
        self.layout = capi.python_layout(max_read_length, self.holder.n_adapters)
        self.n_counters = self.layout["total"]
        self.block = np.zeros(self.n_counters, dtype=np.uint64)

    def process(self, seq, qual, offset, segment_start, terminal_n=None):  # (the oracle looks at the bases itself)
        offset = np.ascontiguousarray(offset, dtype=np.uint32)
        n = len(offset) - 1
        res = np.zeros(n, dtype=capi.RESULT_DTYPE)
        for s in range(len(segment_start) - 1):
            a, b = int(segment_start[s]), int(segment_start[s + 1])
            rc = self.lib.faqcs_oracle_trim(self.o, seq.ctypes.data, qual.ctypes.data, offset[a:].ctypes.data,
                                            b - a, res[a:].ctypes.data, self.block.ctypes.data)
            if rc:
                raise FaqcsError(rc, capi.ERR_TEXT.get(rc, "oracle error %d" % rc))
        return res

    def set_quality(self, q):
        # Options::quality is mutable (NextSeq bump): rebuild the params in place
        C.cast(self.o, C.POINTER(capi.Params)).contents.quality = int(q)  # params is the first member

    def sync(self):
        pass

    def counters(self):
        return self.block.copy()

    def kmer_active(self):
        return bool(self.lib.faqcs_oracle_kmer_active(self.o))

    def kmer_points(self):
        n = self.lib.faqcs_oracle_kmer_points(self.o, None, 0)
        out = np.zeros(n, dtype=capi.RAREFACTION_DTYPE)
        if n:
            self.lib.faqcs_oracle_kmer_points(self.o, out.ctypes.data, n)
        return out

    def kmer_totals(self):
        d, t = C.c_uint64(), C.c_uint64()
        self.lib.faqcs_oracle_kmer_totals(self.o, C.byref(d), C.byref(t))
        return int(d.value), int(t.value)

    def kmer_end_table(self):
        tn = int(self.block[self.layout["filter_stats"][0] + capi.TOTAL_NUMBER])
        self.lib.faqcs_oracle_kmer_end_table(self.o, tn)

    def kmer_histogram(self):
        n = self.lib.faqcs_oracle_kmer_histogram(self.o, None, None, 0)
        c = np.zeros(n, dtype=np.uint64)
        k = np.zeros(n, dtype=np.uint64)
        if n:
            self.lib.faqcs_oracle_kmer_histogram(self.o, c.ctypes.data, k.ctypes.data, n)
        return c, k

    def close(self):
        if self.o:
            self.lib.faqcs_oracle_destroy(self.o)
            self.o = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def oracle_factory(opt, max_read_length, in_off):
    return OracleEngine(opt, max_read_length, in_off)
