"""Byte-exact writers for ``<prefix>.stats.txt`` (FaQCs.cpp:759-1034) and the ``--debug`` report tables
(plot.cpp:540-733).  Pure formatting of the counter block produced by the engine; no compute."""
import math
import os

import numpy as np

from . import _capi as capi


def _f2(x, prec=2):
    """C++ ``fixed << setprecision(prec)`` of a double (glibc prints -nan for 0.0/0.0)."""
    if isinstance(x, float) and math.isnan(x):
        return "-nan"
    if isinstance(x, float) and math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return "%.*f" % (prec, x)


def _ddiv(a, b):
    a, b = float(a), float(b)
    if b == 0.0:
        return float("nan") if a == 0.0 else math.copysign(float("inf"), a)
    return a / b


def _fdiv(a, b):
    """float(a)/b evaluated in binary32 as the reference does for the read-length lines (FaQCs.cpp:784,857,870)."""
    a32, b32 = np.float32(a), np.float32(b)
    if b32 == 0:
        return float("nan") if a32 == 0 else float("inf")
    return float(a32 / b32)


def _pct(a, b):
    return _f2(_ddiv(100.0 * float(a), b))


class Counters:
    """Named views into the additive counter block (layout: include/faqcs_mi.h faqcs_layout)."""

    def __init__(self, block, R, n_adapters):
        self.block = np.asarray(block, dtype=np.uint64)
        self.R = R
        self.n_adapters = n_adapters
        self.lay = capi.python_layout(R, n_adapters)
        assert self.lay["total"] == len(self.block), (self.lay["total"], len(self.block))

    def view(self, name):
        o, n = self.lay[name]
        return self.block[o:o + n]

    @property
    def fs(self):
        return self.view("filter_stats")[: capi.NUM_STAT]

    def matrix(self, name, ncol, rows_from=None):
        """Rows = the reference's grow-on-demand row count = 1 + last non-zero row of the QUALITY matrix
        (every covered position increments exactly one quality column; the base matrix is resized to the
        same full_len, trim.cpp:801-803,817-819, but may hold an all-zero last row)."""
        m = self.view(name).reshape(self.R, ncol)
        src = m if rows_from is None else self.view(rows_from).reshape(self.R, capi.NQ)
        nz = np.nonzero(src.any(axis=1))[0]
        rows = int(nz[-1]) + 1 if len(nz) else 0
        return m[:rows]

    def length_hist(self, name):
        h = self.view(name)
        nz = np.nonzero(h)[0]
        return h[: int(nz[-1]) + 1] if len(nz) else h[:0]


def merged_adapter_stats(opt, counters):
    """name -> [reads, bases] like MAP<string,pair> adapter_stats (entries exist only once credited)."""
    out = {}
    if counters.n_adapters == 0:
        return out
    a = counters.view("adapter_stats").reshape(-1, 2)
    for (name, _), (r, b) in zip(opt.adapter, a):
        if r:
            e = out.setdefault(name, [0, 0])
            e[0] += int(r)
            e[1] += int(b)
    return out


def fold_phix_and_adapters(opt, fs, adapter_stats):
    """FaQCs.cpp:89-127: PhiX entries move into READ_PHIX/BASE_PHIX, the rest sum into READ/BASE_ADAPTER."""
    from .options import PHI_X, PHI_X_COMPLEMENT

    fs = fs.copy()
    if opt.filter_phiX:
        for k in (PHI_X, PHI_X_COMPLEMENT):
            if k in adapter_stats:
                r, b = adapter_stats.pop(k)
                fs[capi.READ_PHIX] += r
                fs[capi.BASE_PHIX] += b
    if opt.filter_adapter:
        for r, b in adapter_stats.values():
            fs[capi.READ_ADAPTER] += r
            fs[capi.BASE_ADAPTER] += b
    return fs


def _adapter_lines(fs, adapter_stats):
    out = []
    # sort ascending on (reads, name) and print from the back (FaQCs.cpp:816-848)
    for reads, name in sorted(((v[0], k) for k, v in adapter_stats.items()), reverse=True):
        bases = adapter_stats[name][1]
        out.append("    %s %d reads (%s %%) %d bases (%s %%)\n" % (
            name, reads, _pct(reads, fs[capi.TOTAL_NUMBER]), bases, _pct(bases, fs[capi.TOTAL_LENGTH])))
    return out


def stats_text(opt, fs, adapter_stats, quality):
    """fs: the 25 FilterStat values after fold_phix_and_adapters; quality: Options::quality at exit
    (possibly bumped by the NextSeq check)."""
    c = capi
    T = lambda k: int(fs[k])  # noqa: E731
    L = []
    if opt.qc_only:
        L.append("\n")
        L.append("Reads #: %d\n" % T(c.TOTAL_COUNT))
        L.append("Total bases: %d\n" % T(c.TOTAL_LENGTH))
        L.append("Reads Length: %s\n" % _f2(_fdiv(T(c.TOTAL_LENGTH), T(c.TOTAL_COUNT))))
        L.append("Processed %d reads for quality check only\n" % T(c.TOTAL_NUMBER))
        L.append("  Reads length < %d bp: %d (%s %%)\n" % (opt.min_read_length, T(c.READ_LENGTH), _pct(T(c.READ_LENGTH), T(c.TOTAL_NUMBER))))
        L.append('  Reads have %d continuous base "N": %d (%s %%)\n' % (opt.max_num_poly_N, T(c.READ_NN), _pct(T(c.READ_NN), T(c.TOTAL_NUMBER))))
        L.append("  Low complexity Reads  (>%s%% mono/di-nucleotides): %d (%s %%)\n" % (
            _f2(float(np.float32(opt.low_complexity_cutoff_ratio)) * 100.0), T(c.READ_LOW_COMPLEXITY),
            _pct(T(c.READ_LOW_COMPLEXITY), T(c.TOTAL_NUMBER))))
        L.append("  Reads < average quality %s: %d (%s %%)\n" % (
            _f2(float(np.float32(opt.average_quality))), T(c.READ_AVG_Q), _pct(T(c.READ_AVG_Q), T(c.TOTAL_NUMBER))))
        if opt.filter_phiX:
            L.append("  Reads hits to phiX sequence: %d (%s %%)\n" % (T(c.READ_PHIX), _pct(T(c.READ_PHIX), T(c.TOTAL_NUMBER))))
        if opt.filter_adapter:
            L.append("  Reads with Adapters/Primers: %d (%s %%)\n" % (T(c.READ_ADAPTER), _pct(T(c.READ_ADAPTER), T(c.TOTAL_NUMBER))))
            L += _adapter_lines(fs, adapter_stats)
        return "".join(L)

    tn, tl = T(c.TOTAL_NUMBER), T(c.TOTAL_LENGTH)
    ttn, ttl = T(c.TOTAL_TRIMMED_NUMBER), T(c.TOTAL_TRIMMED_LENGTH)
    L.append("Before Trimming\n")
    L.append("Reads #: %d\n" % tn)
    L.append("Total bases: %d\n" % tl)
    L.append("Reads Length: %s\n" % _f2(_fdiv(tl, tn)))
    L.append("\nAfter Trimming\n")
    L.append("Reads #: %d (%s %%)\n" % (ttn, _pct(ttn, tn)))
    L.append("Total bases: %d (%s %%)\n" % (ttl, _pct(ttl, tl)))
    if ttn > 0:
        L.append("Mean Reads Length: %s\n" % _f2(_fdiv(ttl, ttn)))
    else:
        L.append("Mean Reads Length: 0\n")
    if opt.has_paired():
        prn, pbl = T(c.PAIRED_READ_NUMBER), T(c.PAIRED_BASE_LENGTH)
        L.append("  Paired Reads #: %d (%s %%)\n" % (prn, _pct(prn, ttn)))
        L.append("  Paired total bases: %d (%s %%)\n" % (pbl, _pct(pbl, ttl)))
        L.append("  Unpaired Reads #: %d (%s %%)\n" % (ttn - prn, _pct(ttn - prn, ttn)))
        L.append("  Unpaired total bases: %d (%s %%)\n" % (ttl - pbl, _pct(ttl - pbl, ttl)))
    L.append("\nDiscarded reads #: %d (%s %%)\n" % (tn - ttn, _pct(tn - ttn, tn)))
    L.append("Trimmed bases: %d (%s %%)\n" % (tl - ttl, _pct(tl - ttl, tl)))
    L.append("  Reads Filtered by length cutoff (%d bp): %d (%s %%)\n" % (opt.min_read_length, T(c.READ_LENGTH), _pct(T(c.READ_LENGTH), tn)))
    L.append("  Bases Filtered by length cutoff: %d (%s %%)\n" % (T(c.BASE_LENGTH), _pct(T(c.BASE_LENGTH), tl)))
    L.append('  Reads Filtered by continuous base "N" (%d): %d (%s %%)\n' % (opt.max_num_poly_N, T(c.READ_NN), _pct(T(c.READ_NN), tn)))
    L.append('  Bases Filtered by continuous base "N": %d (%s %%)\n' % (T(c.BASE_NN), _pct(T(c.BASE_NN), tl)))
    L.append("  Reads Filtered by low complexity ratio (%s): %d (%s %%)\n" % (
        _f2(float(np.float32(opt.low_complexity_cutoff_ratio)), 1), T(c.READ_LOW_COMPLEXITY), _pct(T(c.READ_LOW_COMPLEXITY), tn)))
    L.append("  Bases Filtered by low complexity ratio: %d (%s %%)\n" % (T(c.BASE_LOW_COMPLEXITY), _pct(T(c.BASE_LOW_COMPLEXITY), tl)))
    if np.float32(opt.average_quality) > 0.0:
        L.append("  Reads Filtered by avg quality (%s): %d (%s %%)\n" % (
            _f2(float(np.float32(opt.average_quality))), T(c.READ_AVG_Q), _pct(T(c.READ_AVG_Q), tn)))
        L.append("  Bases Filtered by avg quality: %d (%s %%)\n" % (T(c.BASE_AVG_Q), _pct(T(c.BASE_AVG_Q), tl)))
    if opt.filter_phiX:
        L.append("  Reads Filtered by phiX sequence: %d (%s %%)\n" % (T(c.READ_PHIX), _pct(T(c.READ_PHIX), tn)))
        L.append("  Bases Filtered by phiX sequence: %d (%s %%)\n" % (T(c.BASE_PHIX), _pct(T(c.BASE_PHIX), tl)))
    L.append("  Reads Trimmed by quality (%s): %d (%s %%)\n" % (_f2(float(quality), 1), T(c.READ_QUAL_TRIM), _pct(T(c.READ_QUAL_TRIM), tn)))
    L.append("  Bases Trimmed by quality: %d (%s %%)\n" % (T(c.BASE_QUAL_TRIM), _pct(T(c.BASE_QUAL_TRIM), tl)))
    if opt.trim_5 > 0:
        L.append("  Reads Trimmed with %d bp from 5' end\n" % opt.trim_5)
    if opt.trim_3 > 0:
        L.append("  Reads Trimmed with %d bp from 3' end\n" % opt.trim_3)
    if opt.filter_adapter:
        L.append("  Reads Trimmed with Adapters/Primers: %d (%s %%)\n" % (T(c.READ_ADAPTER), _pct(T(c.READ_ADAPTER), tn)))
        L.append("  Bases Trimmed with Adapters/Primers: %d (%s %%)\n" % (T(c.BASE_ADAPTER), _pct(T(c.BASE_ADAPTER), tl)))
        L += _adapter_lines(fs, adapter_stats)
    if opt.replace_N:
        L.append("\nN base random substitution: A %d, T %d, C %d, G %d\n" % (T(c.N_TO_A), T(c.N_TO_T), T(c.N_TO_C), T(c.N_TO_G)))
    return "".join(L)


# ---- report tables (plot.cpp) ------------------------------------------------------------------------
def matrix_text(m):  # plot.cpp:613-640 ; caller skips the file when there are no rows
    return "".join("\t".join(str(int(v)) for v in row) + "\n" for row in m)


def quality_histogram_text(read_hist, base_hist):  # plot.cpp:642-663
    out = ["Score\treadsNum\treadsBases\n"]
    for i in range(capi.NQ - 1, -1, -1):
        out.append("%d\t%d\t%d\n" % (i, int(read_hist[i]), int(base_hist[i])))
    return "".join(out)


def base_content_text(comp):  # plot.cpp:540-611 ; comp: [10001][6]
    out = []
    comp = np.asarray(comp).reshape(capi.NCOMP_BIN, capi.NCOMP_KIND)
    for k, name in enumerate(("A", "T", "C", "G", "N", "GC")):
        col = comp[:, k]
        for i in np.nonzero(col)[0]:
            out.append("%s\t%.2f\t%d\n" % (name, int(i) * 0.01, int(col[i])))
    return "".join(out)


def length_histogram_text(h):  # plot.cpp:665-681
    return "".join("%d\t%d\n" % (i, int(h[i])) for i in range(1, len(h)))


def kmer_histogram_text(count, nkeys):  # plot.cpp:683-714
    return "".join("%d %d\n" % (int(c), int(k)) for c, k in zip(count, nkeys))


def rarefaction_text(points):  # plot.cpp:716-733
    out, last = [], 0
    for p in points:
        out.append("%d\t%d\t%d\n" % (int(p["num_seq"]) - last, int(p["distinct_kmer"]), int(p["total_kmer"])))
        last = int(p["num_seq"])
    return "".join(out)


def write_debug_tables(opt, counters, kmer_hist=None, kmer_points=None):
    """The files plot() leaves behind with --debug (plot.cpp:31-91, :517-537).  The R/PDF step is out of
    scope; the reference prints `sh: 1: R: not found` when R is absent."""
    d, p = opt.output_dir, opt.prefix

    def put(name, text, skip_if_empty=False):
        if skip_if_empty and not text:
            return
        with open(os.path.join(d, name), "w") as f:
            f.write(text)

    put("qa.%s.quality.matrix" % p, matrix_text(counters.matrix("pre_qual", capi.NQ)), True)
    put("%s.quality.matrix" % p, matrix_text(counters.matrix("post_qual", capi.NQ)), True)
    put("qa.%s.base.matrix" % p, matrix_text(counters.matrix("pre_base", capi.NBASE, "pre_qual")), True)
    put("%s.base.matrix" % p, matrix_text(counters.matrix("post_base", capi.NBASE, "post_qual")), True)
    put("qa.%s.for_qual_histogram.txt" % p, quality_histogram_text(counters.view("pre_read_qhist"), counters.view("pre_base_qhist")))
    put("%s.for_qual_histogram.txt" % p, quality_histogram_text(counters.view("post_read_qhist"), counters.view("post_base_qhist")))
    put("qa.%s.base_content.txt" % p, base_content_text(counters.view("pre_comp")))
    put("%s.base_content.txt" % p, base_content_text(counters.view("post_comp")))
    put("qa.%s.length_count.txt" % p, length_histogram_text(counters.length_hist("pre_len_hist")))
    put("%s.length_count.txt" % p, length_histogram_text(counters.length_hist("post_len_hist")))
    if kmer_hist is not None and len(kmer_hist[0]):
        put("%s.kmerH.txt" % p, kmer_histogram_text(*kmer_hist))
        put("%s.Kmercount.txt" % p, rarefaction_text(kmer_points))
