"""Host driver: the FaQCs process contract (FaQCs.cpp main / process_paired / process_unpaired) around
an *engine* that stands where the reference calls ``trim()``.

Architecture differs from the reference on purpose: instead of one ``trim()`` per 32 768-read buffer the
driver packs several buffers (``segments``) into one structure-of-arrays batch and submits them together;
the segment table keeps every 32 768-granular semantic of the reference (adapter groups of 8, k-mer
rarefaction points, TOTAL_NUMBER ordering mate-1 buffer / mate-2 buffer).  Observable outputs -- the
trimmed FASTQs, ``<prefix>.stats.txt``, the --debug tables, exit codes and messages -- follow the
reference byte for byte (SURVEY.md section 8b "outer seam").

``run(argv)`` uses the HIP engine.  ``engine_factory`` exists so the tests can drive the same host logic
with the CPU checker; nothing in this package constructs any other engine.
"""
import gzip
import os
import sys

import numpy as np

from . import _capi as capi
from . import report
from .options import AUTO_DETECT_QUALITY_OFFSET, DEFAULT_NEXTSEQ_QUALITY_SCORE, Options, parse_args

BUFFER_SIZE = capi.SEGMENT_READS  # FaQCs.cpp:232
PAD = 32  # readable slack the kernels' aligned 16-byte loads may touch


class FatalError(Exception):
    """An error the reference reports as ``Caught the error <msg>`` + EXIT_FAILURE (FaQCs.cpp:136-147)."""


def parse_id(d: bytes) -> bytes:
    """trim.cpp:188-222"""
    loc = d.find(b" ")
    if loc < 0:
        loc = len(d)
    if loc > 1 and d[loc - 1:loc].isdigit() and d[loc - 2:loc - 1] in (b".", b"/"):
        loc -= 2
    return d[:loc]


def auto_detect_quality_offset(quals) -> int:
    """trim.cpp:599-617 over a buffer of quality strings (first decisive byte wins)."""
    for q in quals:
        if not q:
            continue
        a = np.frombuffer(q, dtype=np.uint8).view(np.int8)
        hit = np.nonzero((a > 74) | (a < 59))[0]
        if len(hit):
            return 64 if a[hit[0]] > 74 else 33
    raise FatalError("trim.cpp:auto_detect_quality_offset: Unknown quality format!")


def auto_detect_next_seq(defs) -> bool:
    """trim.cpp:619-626"""
    return bool(defs) and defs[0].startswith(b"@NS")


class FastqReader:
    """fastq.cpp:8-125 ``next_read`` semantics on a (possibly gzip'd) file; line terminators \\n or \\r."""

    def __init__(self, path, what):
        try:
            f = open(path, "rb")
        except OSError:
            raise FatalError("I/O error")
        magic = f.read(2)
        f.seek(0)
        self.f = gzip.open(f, "rb") if magic == b"\x1f\x8b" else f
        self.what = what

    def next_read(self):
        d = self.f.readline()
        if not d:
            return None
        s = self.f.readline()
        if not s:
            raise FatalError("fastq.cpp:next_read: Unable to read sequence")
        p = self.f.readline()
        if not p:
            raise FatalError("fastq.cpp:next_read: Unable to read '+'")
        q = self.f.readline()
        if not q and not p.endswith((b"\n", b"\r")):
            raise FatalError("fastq.cpp:next_read: Error reading '+' delimiter")
        d, s, q = (x.split(b"\n")[0].split(b"\r")[0] for x in (d, s, q))
        if len(s) != len(q):
            raise FatalError("fastq.cpp:next_read: |Sequence| != |Quality|")
        return d, s, q

    def close(self):
        self.f.close()


def pack_segments(buffers):
    """buffers: list of lists of (def, seq, qual).  -> (seq arena, qual arena, offset, segment_start)"""
    lens, seg = [], [0]
    for b in buffers:
        lens.extend(len(r[1]) for r in b)
        seg.append(len(lens))
    offset = np.zeros(len(lens) + 1, dtype=np.uint32)
    if lens:
        offset[1:] = np.cumsum(np.asarray(lens, dtype=np.int64)).astype(np.uint32)
    pad = b"\0" * PAD
    seq = np.frombuffer(b"".join([pad] + [r[1] for b in buffers for r in b] + [pad]), dtype=np.uint8)
    qual = np.frombuffer(b"".join([pad] + [r[2] for b in buffers for r in b] + [pad]), dtype=np.uint8)
    # arenas start PAD bytes in so the aligned loads may under-read; hand the engine the shifted views
    return seq[PAD:], qual[PAD:], offset, np.asarray(seg, dtype=np.uint32)


def terminal_n_flags(seq, offset):
    """faqcs_batch.terminal_n for a packed host arena: bit 0 = a read's first base is 'N', bit 1 = its last (what
    mask_quality_terminal_N, trim.cpp:1191-1216, tests first); 0 for an empty read."""
    off = offset.astype(np.int64)
    n = len(off) - 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint8)
    nonempty = off[1:] > off[:-1]
    first = seq[np.minimum(off[:-1], max(len(seq) - 1, 0))] == ord("N")
    last = seq[np.maximum(off[1:] - 1, 0)] == ord("N")
    return ((first & nonempty).astype(np.uint8) | ((last & nonempty).astype(np.uint8) << 1))


def edited_arenas(opt, in_off, seq, qual, offset):
    """The rule-based byte edits of include/faqcs_mi.h faqcs_read_result, vectorised over a whole arena:
    terminal-N quality masking (trim.cpp:1191-1216), G->N (trim.cpp:390-403), offset re-encode
    (trim.cpp:516-525).  Only bytes inside a valid read's kept window are ever written out."""
    n = len(offset) - 1
    total = int(offset[-1])
    s = np.array(seq[:total], dtype=np.uint8)
    q = np.array(qual[:total], dtype=np.uint8)
    if n:
        lens = np.diff(offset.astype(np.int64))
        nz = lens > 0
        first = offset[:-1][nz].astype(np.int64)
        last = offset[1:][nz].astype(np.int64) - 1
        idx = np.nonzero(nz)[0]
        for i in idx[(s[first] == 78) | (s[last] == 78)]:
            a, b = int(offset[i]), int(offset[i + 1])
            k = a
            while k < b and s[k] == 78:
                q[k] = in_off & 0xFF
                k += 1
            k = b
            while k > a and s[k - 1] == 78:
                q[k - 1] = in_off & 0xFF
                k -= 1
    qs = np.maximum(q.view(np.int8).astype(np.int16) - in_off, 0)
    if opt.replace_to_N_q > 0:
        s[(s == 71) & (qs < opt.replace_to_N_q)] = 78
    if in_off != opt.output_quality_offset:
        q = (qs + opt.output_quality_offset).astype(np.uint8)
    return s, q


class Run:
    """State the reference keeps in main(): filter_stats, adapter_stats, PlotInfo, Options (FaQCs.cpp:67-69)."""

    def __init__(self, opt: Options, engine_factory, err, batch_buffers=8, max_read_length=capi.MAX_READ_LENGTH):
        self.opt = opt
        self.engine_factory = engine_factory
        self.err = err
        self.engine = None
        self.batch_buffers = batch_buffers
        self.max_read_length = max_read_length
        self.in_off = opt.input_quality_offset
        self.quality = opt.quality
        self.paired_read_number = 0
        self.paired_base_length = 0

    def say(self, msg):
        print(msg, file=self.err)

    def ensure_engine(self):
        if self.engine is None:
            self.engine = self.engine_factory(self.opt, self.max_read_length, self.in_off)
            if self.quality != self.opt.quality:
                self.engine.set_quality(self.quality)
        return self.engine

    def nextseq_check(self, defs):
        if self.quality < DEFAULT_NEXTSEQ_QUALITY_SCORE and auto_detect_next_seq(defs):
            self.say("The input looks like NextSeq data and the quality level (-q) is adjusted to %d for trimming."
                     % DEFAULT_NEXTSEQ_QUALITY_SCORE)
            self.quality = DEFAULT_NEXTSEQ_QUALITY_SCORE
            if self.engine is not None:
                self.engine.set_quality(self.quality)

    # ---- FaQCs.cpp:153-538 ----------------------------------------------------------------------
    def process_paired(self):
        opt = self.opt
        try:
            fin1 = FastqReader(opt.input_read1_file, "read one")
        except FatalError:
            self.say("Unable to open %s for loading read one sequences" % opt.input_read1_file)
            raise
        try:
            fin2 = FastqReader(opt.input_read2_file, "read two")
        except FatalError:
            self.say("Unable to open %s for loading read two sequences" % opt.input_read2_file)
            raise
        fout1 = fout2 = foutu = foutd = None
        if not opt.qc_only:
            fout1 = open(opt.trimmed_read1_file, "wb")
            fout2 = open(opt.trimmed_read2_file, "wb")
            foutu = open(opt.trimmed_unpaired_file, "wb")
            if opt.trimmed_discard_file:
                foutd = open(opt.trimmed_discard_file, "wb")
        pending = []  # list of (buffer1, buffer2)
        check_for_next_seq = True
        b1, b2 = [], []

        def flush():
            if not pending:
                return
            eng = self.ensure_engine()
            bufs = [b for pair in pending for b in pair]  # reference call order: trim(buffer1), trim(buffer2)
            seq, qual, offset, seg = pack_segments(bufs)
            res = eng.process(seq, qual, offset, seg, terminal_n_flags(seq, offset))
            es, eq = edited_arenas(opt, self.in_off, seq, qual, offset) if not opt.qc_only else (None, None)
            valid = (res["flags"] & capi.F_VALID) != 0
            for k, (x1, x2) in enumerate(pending):
                o1, o2 = int(seg[2 * k]), int(seg[2 * k + 1])
                for i in range(len(x1)):
                    i1, i2 = o1 + i, o2 + i
                    v1, v2 = valid[i1], valid[i2]
                    if v1 and v2:
                        self.paired_read_number += 2
                        self.paired_base_length += int(res["len"][i1]) + int(res["len"][i2])
                    if opt.qc_only:
                        continue
                    if v1 and v2:
                        _write(fout1, x1[i][0], es, eq, offset, res, i1)
                        _write(fout2, x2[i][0], es, eq, offset, res, i2)
                    else:
                        if v1:
                            _write(foutu, x1[i][0], es, eq, offset, res, i1)
                        elif v2:
                            _write(foutu, x2[i][0], es, eq, offset, res, i2)
                        if foutd is not None:
                            if not v1:
                                _write_raw(foutd, x1[i])
                            if not v2:
                                _write_raw(foutd, x2[i])
            pending.clear()

        while True:
            r1 = fin1.next_read()
            r2 = fin2.next_read()
            if r1 is None and r2 is None:
                if self.in_off == AUTO_DETECT_QUALITY_OFFSET:
                    self.in_off = auto_detect_quality_offset([r[2] for r in b1])
                    if self.in_off != auto_detect_quality_offset([r[2] for r in b2]):
                        self.say("Inconsistent quality offset detection between reads one and two")
                        raise FatalError("FaQCs.cpp:process_paired: I/O Error")
                flush()  # buffers trimmed with the pre-bump quality must go first
                self.nextseq_check([r[0] for r in b1])  # Q16: runs on the last partial buffer regardless
                pending.append((b1, b2))
                flush()
                break
            if (r1 is None) != (r2 is None):
                if r1 is not None:
                    self.say("Did not find a match to read one: " + r1[0].decode("latin-1"))
                else:
                    self.say("Did not find a match to read two: " + r2[0].decode("latin-1"))
                raise FatalError("FaQCs.cppI/O error")
            if parse_id(r1[0]) != parse_id(r2[0]):
                self.say("Read one id (%s)\ndoes not match\nread two id (%s)" % (
                    parse_id(r1[0]).decode("latin-1"), parse_id(r2[0]).decode("latin-1")))
                raise FatalError("FaQCs.cpp:trim: I/O error")
            b1.append(r1)
            b2.append(r2)
            if len(b1) == BUFFER_SIZE:
                if self.in_off == AUTO_DETECT_QUALITY_OFFSET:
                    self.in_off = auto_detect_quality_offset([r[2] for r in b1])
                    if self.in_off != auto_detect_quality_offset([r[2] for r in b2]):
                        self.say("Inconsistent quality offset detection between reads one and two")
                        raise FatalError("FaQCs.cpp:process_paired: I/O Error")
                if check_for_next_seq:
                    self.nextseq_check([r[0] for r in b1])
                    check_for_next_seq = False
                pending.append((b1, b2))
                b1, b2 = [], []
                if len(pending) >= self.batch_buffers:
                    flush()
        fin1.close()
        fin2.close()
        for f in (fout1, fout2, foutu, foutd):
            if f is not None:
                f.close()
        if self.engine is not None:
            self.engine.kmer_end_table()  # FaQCs.cpp:518-537

    # ---- FaQCs.cpp:540-757 ----------------------------------------------------------------------
    def process_unpaired(self):
        opt = self.opt
        try:
            fin = FastqReader(opt.input_unpaired_file, "unpaired")
        except FatalError:
            self.say("Unable to open %s for loading unpaired read sequences" % opt.input_unpaired_file)
            raise
        fout = foutd = None
        if not opt.qc_only:
            fout = open(opt.trimmed_unpaired_file, "wb")  # "wT": truncates what process_paired wrote (Q17)
            if opt.trimmed_discard_file:
                foutd = open(opt.trimmed_discard_file, "wb")
        pending, buf = [], []
        check_for_next_seq = True

        def flush():
            if not pending:
                return
            eng = self.ensure_engine()
            seq, qual, offset, seg = pack_segments(pending)
            res = eng.process(seq, qual, offset, seg, terminal_n_flags(seq, offset))
            if not opt.qc_only:
                es, eq = edited_arenas(opt, self.in_off, seq, qual, offset)
                valid = (res["flags"] & capi.F_VALID) != 0
                i = 0
                for b in pending:
                    for r in b:
                        if valid[i]:
                            _write(fout, r[0], es, eq, offset, res, i)
                        elif foutd is not None:
                            _write_raw(foutd, r)
                        i += 1
            pending.clear()

        while True:
            r = fin.next_read()
            if r is None:
                if self.in_off == AUTO_DETECT_QUALITY_OFFSET:
                    self.in_off = auto_detect_quality_offset([x[2] for x in buf])
                flush()
                self.nextseq_check([x[0] for x in buf])
                pending.append(buf)
                flush()
                break
            buf.append(r)
            if len(buf) == BUFFER_SIZE:
                if self.in_off == AUTO_DETECT_QUALITY_OFFSET:
                    self.in_off = auto_detect_quality_offset([x[2] for x in buf])
                if check_for_next_seq:
                    self.nextseq_check([x[0] for x in buf])
                    check_for_next_seq = False
                pending.append(buf)
                buf = []
                if len(pending) >= 2 * self.batch_buffers:
                    flush()
        fin.close()
        for f in (fout, foutd):
            if f is not None:
                f.close()
        if self.engine is not None:
            self.engine.kmer_end_table()

    def finish(self):
        opt = self.opt
        eng = self.ensure_engine()
        block = eng.counters()
        counters = report.Counters(block, self.max_read_length, eng.holder.n_adapters)
        fs = counters.fs.copy()
        fs[capi.PAIRED_READ_NUMBER] += self.paired_read_number
        fs[capi.PAIRED_BASE_LENGTH] += self.paired_base_length
        adapter_stats = report.merged_adapter_stats(opt, counters)
        fs = report.fold_phix_and_adapters(opt, fs, adapter_stats)
        try:
            with open(opt.stats_file, "w") as f:
                f.write(report.stats_text(opt, fs, adapter_stats, self.quality))
        except OSError:
            self.say("Unable to open %s for writing filtering statistics" % opt.stats_file)
        if not opt.trim_only and opt.debug:
            report.write_debug_tables(opt, counters, eng.kmer_histogram(), eng.kmer_points())
        self.filter_stats = fs
        self.counters = counters
        return fs


def _write(f, d, es, eq, offset, res, i):
    a = int(offset[i]) + int(res["start"][i])
    b = a + int(res["len"][i])
    f.write(d + b"\n" + es[a:b].tobytes() + b"\n+\n" + eq[a:b].tobytes() + b"\n")  # fastq.cpp:127-138


def _write_raw(f, r):
    f.write(r[0] + b"\n" + r[1] + b"\n+\n" + r[2] + b"\n")


def _remove_file(path, err):
    if path and os.path.isfile(path):  # FaQCs.cpp:1046-1053, file_util.cpp:11-20 (regular files only)
        print("The output %s file exists and will be overwritten." % path, file=err)
        os.unlink(path)


def _hip_engine(opt, max_read_length, in_off):
    from .engine import HipEngine

    return HipEngine(opt, max_read_length, in_off)


def run(argv, engine_factory=None, err=None, batch_buffers=8, max_read_length=capi.MAX_READ_LENGTH):
    """FaQCs.cpp:36-151.  Returns the process exit code (0 / 1)."""
    err = err or sys.stderr
    engine_factory = engine_factory or _hip_engine
    try:
        try:
            opt = parse_args(list(argv))
        except ValueError as e:
            raise FatalError(str(e))
        for m in opt.messages:
            print(m, file=err)
        if opt.print_usage:
            if not opt.version:
                print("FaQCs version 2.10 (MI355X engine): see the reference usage text for the flag list", file=err)
            return 1
        if not os.path.isdir(opt.output_dir):
            try:
                os.mkdir(opt.output_dir, 0o700)  # file_util.cpp:33-36
            except OSError:
                print('Unable to create requested output directory: "%s"' % opt.output_dir, file=err)
                return 1
        for p in (opt.plots_file, opt.stats_file, opt.trimmed_read1_file, opt.trimmed_read2_file,
                  opt.trimmed_unpaired_file, opt.trimmed_discard_file):
            _remove_file(p, err)
        r = Run(opt, engine_factory, err, batch_buffers, max_read_length)
        if opt.has_paired():
            r.process_paired()
        if opt.has_unpaired():
            r.process_unpaired()
        r.finish()
        run.last = r
        return 0
    except FatalError as e:
        print("Caught the error %s" % e, file=err)
        return 1
    except Exception as e:  # engine errors carry the reference's message text
        from .engine import FaqcsError

        if isinstance(e, FaqcsError):
            print("Caught the error %s" % e, file=err)
            return 1
        raise
