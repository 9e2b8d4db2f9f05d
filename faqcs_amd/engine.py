"""HipEngine: the product engine -- a thin numpy wrapper over the C ABI of libfaqcs_mi.so.

An *engine* is what the host driver (faqcs_amd/driver.py) calls where the reference calls ``trim()``
(FaQCs.h:245-248).  ``process()`` takes a structure-of-arrays batch whose segments are the reference's
individual trim() calls.  There is exactly one engine in the product: this one.  (tests/ has an
OracleEngine with the same surface that wraps the CPU checker; it is never imported from here.)
"""
import ctypes as C

import numpy as np

from . import _capi as capi


class FaqcsError(RuntimeError):
    """Raised with the text the reference would print after ``Caught the error``."""

    def __init__(self, code, text):
        super().__init__(text)
        self.code = code


def _check(lib, rc):
    if rc != 0:
        msg = capi.ERR_TEXT.get(rc) or (lib.faqcs_last_error() or b"").decode() or "faqcs error %d" % rc
        raise FaqcsError(rc, msg)


class HipEngine:
    name = "hip"

    def __init__(self, opt, max_read_length, input_quality_offset=None, device=-1, kmer_table_slots=0):
        self.lib = capi.load_library()
        self.holder = capi.ParamsHolder(opt, max_read_length, input_quality_offset, kmer_table_slots)
        self.ctx = C.c_void_p()
        _check(self.lib, self.lib.faqcs_create(C.byref(self.holder.p), device, C.byref(self.ctx)))
        lay = capi.Layout()
        _check(self.lib, self.lib.faqcs_counters_layout(max_read_length, self.holder.n_adapters, C.byref(lay)))
        self.layout = lay
        self.n_counters = int(lay.total)

    # -- the trim() seam ---------------------------------------------------------------------------
    def process(self, seq, qual, offset, segment_start, terminal_n=None):
        """seq/qual: uint8 arenas (host memory: the library copies them into padded device buffers), offset: uint32[n+1],
        segment_start: uint32[n_segments+1]; terminal_n: optional uint8[n] (faqcs_batch.terminal_n, see terminal_n_flags()).
        Returns the per-read result array."""
        offset = np.ascontiguousarray(offset, dtype=np.uint32)
        segment_start = np.ascontiguousarray(segment_start, dtype=np.uint32)
        n = len(offset) - 1
        res = np.zeros(n, dtype=capi.RESULT_DTYPE)
        if terminal_n is not None:
            terminal_n = np.ascontiguousarray(terminal_n, dtype=np.uint8)
            assert len(terminal_n) >= n
        b = capi.Batch(seq.ctypes.data, qual.ctypes.data, offset.ctypes.data, n, len(segment_start) - 1,
                       segment_start.ctypes.data, 0, terminal_n.ctypes.data if terminal_n is not None and n else None)
        _check(self.lib, self.lib.faqcs_submit(self.ctx, C.byref(b), res.ctypes.data))
        _check(self.lib, self.lib.faqcs_sync(self.ctx))
        return res

    def set_quality(self, q):
        _check(self.lib, self.lib.faqcs_set_quality(self.ctx, int(q)))

    def sync(self):
        _check(self.lib, self.lib.faqcs_sync(self.ctx))

    # -- accumulators ------------------------------------------------------------------------------
    def counters_device(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        _check(self.lib, self.lib.faqcs_counters_device(self.ctx, C.byref(ptr), C.byref(n)))
        return ptr.value, int(n.value)

    def counters_export(self, d_dst, n_u64):
        """Device-to-device copy of the block into a caller-owned buffer (the buffer the collective runs on)."""
        _check(self.lib, self.lib.faqcs_counters_export(self.ctx, d_dst, int(n_u64)))

    def counters_import(self, d_src, n_u64):
        _check(self.lib, self.lib.faqcs_counters_import(self.ctx, d_src, int(n_u64)))

    def comm_init(self, comm_id, rank, world):
        """The library's own RCCL communicator (include/faqcs_mi.h): comm_id = the FAQCS_COMM_ID_BYTES bytes rank 0 got from comm_id()."""
        buf = C.create_string_buffer(bytes(comm_id), 128)
        _check(self.lib, self.lib.faqcs_comm_init(self.ctx, buf, int(rank), int(world)))
        self.has_comm = True

    def comm_id(self):
        buf = C.create_string_buffer(128)
        _check(self.lib, self.lib.faqcs_comm_id(buf))
        return buf.raw

    def comm_allreduce_counters(self):
        """All-reduce(sum) of the counter block in place, enqueued on the context's compute stream (no copy, no host sync)."""
        _check(self.lib, self.lib.faqcs_comm_allreduce_counters(self.ctx))

    def counters(self):
        out = np.zeros(self.n_counters, dtype=np.uint64)
        _check(self.lib, self.lib.faqcs_finish(self.ctx, out.ctypes.data, self.n_counters))
        return out

    def kmer_active(self):
        return bool(self.lib.faqcs_kmer_active(self.ctx))

    def kmer_points(self):
        n = C.c_uint32()
        _check(self.lib, self.lib.faqcs_kmer_points(self.ctx, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=capi.RAREFACTION_DTYPE)
        if n.value:
            _check(self.lib, self.lib.faqcs_kmer_points(self.ctx, out.ctypes.data, n.value, C.byref(n)))
        return out

    def kmer_end_table(self):
        """process_paired / process_unpaired epilogue (FaQCs.cpp:518-537, :737-756)."""
        _check(self.lib, self.lib.faqcs_kmer_end_table(self.ctx))

    def kmer_finish_pass(self):
        """The pass is complete: its k-mers are counted (in one piece, without the device table, when they fit the group buffers)."""
        _check(self.lib, self.lib.faqcs_kmer_finish_pass(self.ctx))

    def kmer_totals(self):
        d, t = C.c_uint64(), C.c_uint64()
        _check(self.lib, self.lib.faqcs_kmer_totals(self.ctx, C.byref(d), C.byref(t)))
        return int(d.value), int(t.value)

    def kmer_histogram(self):
        n = C.c_uint64()
        _check(self.lib, self.lib.faqcs_kmer_histogram(self.ctx, None, None, 0, C.byref(n)))
        c = np.zeros(n.value, dtype=np.uint64)
        k = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            _check(self.lib, self.lib.faqcs_kmer_histogram(self.ctx, c.ctypes.data, k.ctypes.data, n.value, C.byref(n)))
        return c, k

    # -- k-mers across GPUs (owner-partitioned tables; faqcs_amd/parallel.py drives the exchange) --------------
    def kmer_partition(self, rank, world, n_epochs):
        self._part = (rank, world, n_epochs)
        _check(self.lib, self.lib.faqcs_kmer_partition(self.ctx, rank, world, n_epochs))

    def kmer_set_epochs(self, epochs):
        e = np.ascontiguousarray(epochs, dtype=np.uint32)
        _check(self.lib, self.lib.faqcs_kmer_set_epochs(self.ctx, e.ctypes.data, len(e)))

    def kmer_outbox(self):
        """(device pointer to the (key, epoch) pairs grouped by destination, pairs per destination)."""
        ptr = C.c_void_p()
        counts = np.zeros(self._part[1], dtype=np.uint64)
        _check(self.lib, self.lib.faqcs_kmer_outbox(self.ctx, C.byref(ptr), counts.ctypes.data))
        return ptr.value, counts

    def kmer_insert_device(self, d_items, n_items):
        _check(self.lib, self.lib.faqcs_kmer_insert_device(self.ctx, d_items, int(n_items)))

    def kmer_epoch_counts(self):
        n = self._part[2]
        d = np.zeros(n, dtype=np.uint64)
        t = np.zeros(n, dtype=np.uint64)
        _check(self.lib, self.lib.faqcs_kmer_epoch_counts(self.ctx, d.ctypes.data, t.ctypes.data, n))
        return d, t

    def close(self):
        if self.ctx:
            self.lib.faqcs_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
