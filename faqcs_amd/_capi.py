"""ctypes view of include/faqcs_mi.h (the C ABI of libfaqcs_mi.so) and the data model shared with the
test oracle.  No torch types cross this boundary: numpy arrays / raw pointers only."""
import ctypes as C
import os

import numpy as np

ABI_VERSION = 2
NUM_STAT = 25
NQ = 42
NBASE = 5
NCOMP_BIN = 10001
NCOMP_KIND = 6
SEGMENT_READS = 32768
MAX_READ_LENGTH = 32767       # FAQCS_MAX_READ_LENGTH: longest read the library takes (batches with a read > 1 024 bases run on trim_long)
FAST_READ_LENGTH = 1024       # longest read of the chunked kernels

(TOTAL_COUNT, TOTAL_NUMBER, TOTAL_LENGTH, TOTAL_TRIMMED_NUMBER, TOTAL_TRIMMED_LENGTH, PAIRED_READ_NUMBER,
 PAIRED_BASE_LENGTH, READ_LENGTH, BASE_LENGTH, READ_NN, BASE_NN, READ_PHIX, BASE_PHIX, READ_ADAPTER,
 BASE_ADAPTER, READ_AVG_Q, BASE_AVG_Q, READ_QUAL_TRIM, BASE_QUAL_TRIM, READ_LOW_COMPLEXITY,
 BASE_LOW_COMPLEXITY, N_TO_A, N_TO_T, N_TO_G, N_TO_C) = range(NUM_STAT)

F_VALID, F_FILTER_MASK, F_FILTER_SHIFT = 0x1, 0xE, 1
F_QUAL_TRIMMED, F_ADAPTER, F_POLY_N_SEEN, F_ERR_QUALITY, F_ERR_BASE = 0x10, 0x20, 0x40, 0x100, 0x200

E_INVAL, E_NODEVICE, E_QUALITY, E_BASE, E_NOMEM, E_KMER_FULL = -1, -2, -3, -4, -5, -6
EPOCH_NONE = 0xFFFFFFFF

# the messages the reference throws at the corresponding sites (fastq.h:32, seq_overlap.cpp:409)
ERR_TEXT = {
    E_QUALITY: "fastq.h:quality_score: Found a quality score value that is greater than the maximum allowed quality score",
    E_BASE: "seq_overlap.cpp:na_to_bits: Unknown base!",
}


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("mode", C.c_int32), ("quality", C.c_int32),
        ("input_quality_offset", C.c_int32), ("output_quality_offset", C.c_int32),
        ("min_read_length", C.c_uint32), ("max_num_poly_N", C.c_uint32), ("trim_5", C.c_uint32),
        ("trim_3", C.c_uint32), ("replace_to_N_q", C.c_uint32), ("average_quality", C.c_float),
        ("low_complexity_cutoff_ratio", C.c_float), ("filterAdapterMismatchRate", C.c_float),
        ("protect_5", C.c_uint32), ("qc_only", C.c_uint32), ("kmer_rarefaction", C.c_uint32),
        ("kmer", C.c_uint32), ("split_size", C.c_uint32), ("num_subsample", C.c_uint32),
        ("max_read_length", C.c_uint32), ("n_adapters", C.c_uint32),
        ("adapter_seq", C.POINTER(C.c_char_p)), ("kmer_table_slots", C.c_uint64),
    ]


class Batch(C.Structure):
    _fields_ = [
        ("seq", C.c_void_p), ("qual", C.c_void_p), ("offset", C.c_void_p), ("n_reads", C.c_uint32),
        ("n_segments", C.c_uint32), ("segment_start", C.c_void_p), ("max_read_len", C.c_uint32),
        ("terminal_n", C.c_void_p),  # optional per-read flags (faqcs_submit_device only); None = the kernels look themselves
    ]


class Layout(C.Structure):
    _fields_ = [("max_read_length", C.c_uint32), ("n_adapters", C.c_uint32)] + [
        (n, C.c_uint64) for n in (
            "filter_stats", "pre_read_qhist", "pre_base_qhist", "post_read_qhist", "post_base_qhist",
            "pre_len_hist", "post_len_hist", "pre_qual", "post_qual", "pre_base", "post_base", "pre_comp",
            "post_comp", "adapter_stats", "total")
    ]


class KernelTimes(C.Structure):
    _fields_ = [("trim_ms", C.c_double), ("adapter_ms", C.c_double), ("n_launches", C.c_uint64), ("trim_kernel", C.c_char_p),
                ("kmer_ms", C.c_double), ("kmer_insert_ms", C.c_double)]


RESULT_DTYPE = np.dtype([("start", "<u2"), ("len", "<u2"), ("flags", "<u2"), ("adapter", "<u2")])
RAREFACTION_DTYPE = np.dtype([("num_seq", "<u8"), ("distinct_kmer", "<u8"), ("total_kmer", "<u8")])


def python_layout(R, n_adapters):
    """Pure-python statement of the counter-block layout (tests assert the C library and the oracle agree)."""
    o, out = 0, {}
    for name, size in (
        ("filter_stats", 32), ("pre_read_qhist", NQ), ("pre_base_qhist", NQ), ("post_read_qhist", NQ),
        ("post_base_qhist", NQ), ("pre_len_hist", R + 1), ("post_len_hist", R + 1), ("pre_qual", R * NQ),
        ("post_qual", R * NQ), ("pre_base", R * NBASE), ("post_base", R * NBASE),
        ("pre_comp", NCOMP_BIN * NCOMP_KIND), ("post_comp", NCOMP_BIN * NCOMP_KIND),
        ("adapter_stats", 2 * n_adapters),
    ):
        out[name] = (o, size)
        o += size
    out["total"] = o
    return out


class ParamsHolder:
    """Keeps the ctypes Params struct and the adapter string array alive together."""

    def __init__(self, opt, max_read_length, input_quality_offset=None, kmer_table_slots=0):
        seqs = [a[1].encode() for a in opt.adapter] if opt.adapters_active() else []
        self._arr = (C.c_char_p * max(1, len(seqs)))(*seqs)
        off = opt.input_quality_offset if input_quality_offset is None else input_quality_offset
        self.p = Params(
            abi_version=ABI_VERSION, mode=opt.mode, quality=opt.quality, input_quality_offset=off,
            output_quality_offset=opt.output_quality_offset, min_read_length=opt.min_read_length,
            max_num_poly_N=opt.max_num_poly_N, trim_5=opt.trim_5, trim_3=opt.trim_3,
            replace_to_N_q=opt.replace_to_N_q, average_quality=opt.average_quality,
            low_complexity_cutoff_ratio=opt.low_complexity_cutoff_ratio,
            filterAdapterMismatchRate=opt.filterAdapterMismatchRate, protect_5=int(opt.protect_5),
            qc_only=int(opt.qc_only), kmer_rarefaction=int(opt.kmer_rarefaction), kmer=opt.kmer,
            split_size=opt.split_size, num_subsample=opt.num_subsample, max_read_length=max_read_length,
            n_adapters=len(seqs), adapter_seq=C.cast(self._arr, C.POINTER(C.c_char_p)),
            kmer_table_slots=kmer_table_slots,
        )
        self.n_adapters = len(seqs)
        self.max_read_length = max_read_length


_LIB = None


def lib_path():
    # FAQCS_MI_LIB lets a tuning run point at an alternative build of the same library
    return os.environ.get("FAQCS_MI_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfaqcs_mi.so")


def load_library():
    """Loads libfaqcs_mi.so (built in-tree by __graft_entry__.build()).  Fails loudly: there is no CPU
    fallback for the product path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            "libfaqcs_mi.so is missing (%s): build the HIP extension with `python -c 'import __graft_entry__ as g; "
            "g.build()'` -- the FaQCs MI355X hot path has no CPU fallback" % path)
    lib = C.CDLL(path)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "faqcs_abi_version": (i32, []),
        "faqcs_counters_layout": (i32, [u32, u32, C.POINTER(Layout)]),
        "faqcs_counter_rows": (u32, [vp, u32, u32]),
        "faqcs_apply_edits": (i32, [C.POINTER(Params), vp, vp, u32, vp, vp, vp]),
        "faqcs_auto_detect_quality_offset": (i32, [vp, vp, u32]),
        "faqcs_last_error": (C.c_char_p, []),
        "faqcs_create": (i32, [C.POINTER(Params), i32, C.POINTER(vp)]),
        "faqcs_destroy": (None, [vp]),
        "faqcs_submit": (i32, [vp, C.POINTER(Batch), vp]),
        "faqcs_submit_device": (i32, [vp, C.POINTER(Batch), vp]),
        "faqcs_sync": (i32, [vp]),
        "faqcs_submit_async": (i32, [vp, C.POINTER(Batch), vp, C.POINTER(u64)]),
        "faqcs_wait": (i32, [vp, u64]),
        "faqcs_host_alloc": (vp, [C.c_size_t]),
        "faqcs_host_free": (None, [vp]),
        "faqcs_counters_device": (i32, [vp, C.POINTER(vp), C.POINTER(u64)]),
        "faqcs_counters_export": (i32, [vp, vp, u64]),
        "faqcs_counters_import": (i32, [vp, vp, u64]),
        "faqcs_finish": (i32, [vp, vp, u64]),
        "faqcs_reset_counters": (i32, [vp]),
        "faqcs_set_quality": (i32, [vp, i32]),
        "faqcs_kmer_points": (i32, [vp, vp, u32, C.POINTER(u32)]),
        "faqcs_kmer_histogram": (i32, [vp, vp, vp, u64, C.POINTER(u64)]),
        "faqcs_kmer_totals": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "faqcs_kmer_active": (i32, [vp]),
        "faqcs_kmer_end_table": (i32, [vp]),
        "faqcs_kmer_finish_pass": (i32, [vp]),
        "faqcs_kmer_memory_plan": (i32, [vp, u64, vp, u32]),
        "faqcs_kmer_partition": (i32, [vp, u32, u32, u32]),
        "faqcs_kmer_set_epochs": (i32, [vp, vp, u32]),
        "faqcs_kmer_outbox": (i32, [vp, C.POINTER(vp), vp]),
        "faqcs_kmer_outbox_host": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "faqcs_kmer_insert_device": (i32, [vp, vp, u64]),
        "faqcs_kmer_epoch_counts": (i32, [vp, vp, vp, u32]),
        "faqcs_kmer_forward": (i32, [vp, vp, u32]),
        "faqcs_comm_id": (i32, [vp]),
        "faqcs_comm_init": (i32, [vp, vp, u32, u32]),
        "faqcs_comm_allreduce_counters": (i32, [vp]),
        "faqcs_comm_init_all": (i32, [vp, u32]),
        "faqcs_comm_allreduce_counters_all": (i32, [vp, u32]),
        "faqcs_synth_fill": (i32, [i32, vp, vp, vp, u32, u32, u64, u64, C.c_float]),
        "faqcs_synth_fill_genome": (i32, [i32, vp, vp, vp, u32, u32, u64, u64, u64]),
        "faqcs_terminal_n_flags": (i32, [i32, vp, vp, u32, vp]),
        "faqcs_kernel_time_ms": (i32, [vp, C.POINTER(C.c_double), C.POINTER(u64)]),
        "faqcs_debug_words": (i32, [vp, vp, u32]),
        "faqcs_kernel_report": (i32, [vp, C.POINTER(KernelTimes)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError == a symbol the header declares is missing
        fn.restype = res
        fn.argtypes = args
    lib._faqcs_symbols = sorted(sig)
    _LIB = lib
    return lib


def declared_symbols():
    """Every `faqcs_*(` function include/faqcs_mi.h declares (parsed from the header itself)."""
    import re

    hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "faqcs_mi.h")
    text = open(hdr).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(faqcs_[a-z0-9_]+)\s*\(", text)))
