"""Flag surface, defaults and built-in adapter table of the FaQCs command line.

Host-side mirror of the reference's ``Options`` (FaQCs.h:77-144, options.cpp:72-774) restricted to what
the hot path and the output contract need.  The adapter sequences are *data* the drop-in must reproduce
(options.cpp:583-625); the parsing code is our own (argparse-free, getopt_long_only-like: long flags are
accepted with one or two dashes).
"""
import os
from dataclasses import dataclass, field
from typing import List, Tuple

FAQCS_VERSION = "2.10"

MODE_HARD, MODE_BWA, MODE_BWA_PLUS = 0, 1, 2
AUTO_DETECT_QUALITY_OFFSET = -128  # SCHAR_MIN, FaQCs.h:13
DEFAULT_NEXTSEQ_QUALITY_SCORE = 20  # FaQCs.h:15
PHI_X = "__PhiX174_NC_001422__"  # FaQCs.h:29-30
PHI_X_COMPLEMENT = "__PhiX174_NC_001422_complement__"

# options.cpp:583-617 (name, sequence) in list order -- the order decides ties (strict '>' keeps the
# earlier adapter, trim.cpp:1036).
BUILTIN_ADAPTERS: List[Tuple[str, str]] = [
    ("cre-loxp-forward", "TCGTATAACTTCGTATAATGTATGCTATACGAAGTTATTACG"),
    ("cre-loxp-reverse", "AGCATATTGAAGCATATTACATACGATATGCTTCAATAATGC"),
    ("TruSeq-adapter-1", "GGGGTAGTGTGGATCCTCCTCTAGGCAGTTGGGTTATTCTAGAAGCAGATGTGTTGGCTGTTTCTGAAACTCTGGAAAA"),
    ("TruSeq-adapter-3", "CAACAGCCGGTCAAAACATCTGGAGGGTAAGCCATAAACACCTCAACAGAAAA"),
    ("PCR-primer-1", "CGATAACTTCGTATAATGTATGCTATACGAAGTTATTACG"),
    ("PCR-primer-2", "GCATAACTTCGTATAGCATACATTATACGAAGTTATACGA"),
    ("Nextera-primer-adapter-1", "GATCGGAAGAGCACACGTCTGAACTCCAGTCAC"),
    ("Nextera-primer-adapter-2", "GATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT"),
    ("Nextera-junction-adapter-1", "CTGTCTCTTATACACATCTAGATGTGTATAAGAGACAG"),
]
POLYA: Tuple[str, str] = ("polyA", "A" * 20)  # options.cpp:620-625

_COMPLEMENT = bytes.maketrans(b"ATGCatgcMRSVWYHKDBNmrsvwyhkdbn", b"TACGtacgKYSBWRDMHVNkysbwrdmhvn")


def reverse_complement(seq: str) -> str:
    """options.cpp:894-996 ``complement()`` (case preserving, IUPAC aware, then reversed)."""
    return seq.encode().translate(_COMPLEMENT)[::-1].decode()


def phix_sequence() -> str:
    """PhiX174 NC_001422 (5 386 bp), the sequence the reference embeds at options.cpp:633-680."""
    path = os.path.join(os.path.dirname(__file__), "data", "phix174_nc_001422.txt")
    with open(path) as f:
        return "".join(line.strip() for line in f if not line.startswith("#"))


def parse_artifact_file(path: str) -> List[Tuple[str, str]]:
    """FASTA (optionally gz) -> [(defline, sequence)]; behaviour of options.cpp:820-891."""
    import gzip

    with open(path, "rb") as f:
        magic = f.read(2)
    op = gzip.open if magic == b"\x1f\x8b" else open
    out, defline, data = [], "", []
    with op(path, "rb") as f:
        for raw in f:
            line = raw.decode("latin-1")
            k = line.find(">")
            if k >= 0:
                if data:
                    out.append((defline, "".join(data)))
                data = []
                defline = line[k + 1:].split("\n")[0].split("\r")[0]
            else:
                data.append("".join(line.split()))
    if data and "".join(data):
        out.append((defline, "".join(data)))
    return out


class UsageError(Exception):
    """The reference prints usage and exits 1 (options.cpp:439-494)."""


@dataclass
class Options:
    # booleans, options.cpp:98-108
    print_usage: bool = False
    protect_5: bool = False
    replace_N: bool = False
    kmer_rarefaction: bool = False
    discard_output: bool = False
    qc_only: bool = False
    trim_only: bool = False
    filter_adapter: bool = False
    filter_phiX: bool = False
    debug: bool = False
    mode: int = MODE_BWA_PLUS
    prefix: str = "QC"
    plots_file: str = ""
    stats_file: str = ""
    input_read1_file: str = ""
    input_read2_file: str = ""
    input_unpaired_file: str = ""
    trimmed_read1_file: str = ""
    trimmed_read2_file: str = ""
    trimmed_unpaired_file: str = ""
    trimmed_discard_file: str = ""
    output_dir: str = ""
    artifact_file: str = ""
    average_quality: float = 0.0
    low_complexity_cutoff_ratio: float = 0.85
    filterAdapterMismatchRate: float = 0.2
    input_quality_offset: int = AUTO_DETECT_QUALITY_OFFSET
    output_quality_offset: int = 33
    num_thread: int = 0
    quality: int = 5
    min_read_length: int = 50
    max_num_poly_N: int = 2
    kmer: int = 31
    num_subsample: int = 10
    trim_5: int = 0
    trim_3: int = 0
    split_size: int = 1000000
    replace_to_N_q: int = 0
    adapter: List[Tuple[str, str]] = field(default_factory=list)
    version: bool = False
    messages: List[str] = field(default_factory=list)  # what the reference prints on stderr

    def has_paired(self) -> bool:
        return bool(self.input_read1_file)  # FaQCs.h:135-138 tests read1 twice

    def has_unpaired(self) -> bool:
        return bool(self.input_unpaired_file)

    def adapters_active(self) -> bool:
        return self.filter_adapter or self.filter_phiX  # trim.cpp:86, :270


def _strtou(s: str) -> int:
    if not s.isdigit() and s != "":
        raise ValueError("options.cpp:strtou: Invalid character")
    return int(s) if s else 0


def _c_float(x: str) -> float:
    import numpy as np

    return float(np.float32(float(x)))  # Options stores these as `float`


_LONG = {  # name -> takes argument (options.cpp:145-178)
    "mode": True, "5end": True, "3end": True, "adapter": False, "rate": True, "polyA": False,
    "artifactFile": True, "min_L": True, "avg_q": True, "lc": True, "phiX": False, "ascii": True,
    "out_ascii": True, "prefix": True, "stats": True, "split_size": True, "qc_only": False,
    "kmer_rarefaction": False, "subset": True, "discard": False, "substitute": False, "trim_only": False,
    "5trim_off": False, "debug": False, "version": False, "R1": True, "R2": True, "Ru": False, "Rd": False,
    "QRpdf": False, "replace_to_N_q": True,
}
_SHORT = {"d": True, "t": True, "n": True, "1": True, "2": True, "p": True, "q": True, "u": True, "m": True,
          "h": False, "?": False}


def parse_args(argv: List[str]) -> Options:
    """argv without the program name.  Mirrors options.cpp:72-774 including the documented quirks that
    matter for a drop-in; unlike the reference ``-m K`` is accepted (quirk a)."""
    o = Options()
    o.print_usage = len(argv) == 0
    trim_polyA = False
    i = 0
    while i < len(argv):
        a = argv[i]
        i += 1
        if not a.startswith("-") or a == "-":
            continue
        name = a.lstrip("-")
        val = None
        if "=" in name:
            name, val = name.split("=", 1)
        if name in _LONG:
            need = _LONG[name]
        elif name in _SHORT:
            need = _SHORT[name]
        elif len(name) > 1 and name[0] in _SHORT and _SHORT[name[0]] and not a.startswith("--"):
            name, val = name[0], name[1:]  # -q5
            need = True
        else:
            cand = [k for k in _LONG if k.startswith(name)]
            if len(cand) == 1:
                name, need = cand[0], _LONG[cand[0]]
            else:
                o.print_usage = True
                continue
        if need and val is None:
            if i >= len(argv):
                o.print_usage = True
                break
            val = argv[i]
            i += 1
        if name == "mode":
            o.mode = {"hard": MODE_HARD, "bwa": MODE_BWA, "bwa_plus": MODE_BWA_PLUS}.get(val.lower(), -1)
        elif name == "5end":
            o.trim_5 = _strtou(val)
        elif name == "3end":
            o.trim_3 = _strtou(val)
        elif name == "adapter":
            o.filter_adapter = True
        elif name == "rate":
            o.filterAdapterMismatchRate = _c_float(val)
        elif name == "polyA":
            trim_polyA = True
        elif name == "artifactFile":
            o.artifact_file = val
            o.filter_adapter = True
        elif name == "min_L":
            o.min_read_length = _strtou(val)
        elif name == "avg_q":
            o.average_quality = _c_float(val)
        elif name == "lc":
            o.low_complexity_cutoff_ratio = _c_float(val)
        elif name == "phiX":
            o.filter_phiX = True
        elif name == "ascii":
            o.input_quality_offset = int(val)
        elif name == "out_ascii":
            o.output_quality_offset = int(val)
        elif name == "prefix":
            o.prefix = val
        elif name == "stats":
            o.stats_file = val
        elif name == "split_size":
            o.split_size = _strtou(val)
        elif name == "qc_only":
            o.qc_only = True
        elif name == "kmer_rarefaction":
            o.kmer_rarefaction = True
        elif name == "subset":
            o.num_subsample = _strtou(val)
        elif name == "discard":
            o.discard_output = True
        elif name == "substitute":
            o.replace_N = True
        elif name == "trim_only":
            o.trim_only = True
        elif name == "5trim_off":
            o.protect_5 = True
        elif name == "debug":
            o.debug = True
        elif name == "version":
            o.version = True
        elif name == "R1":
            o.input_read1_file = val
        elif name == "R2":
            o.input_read2_file = val
        elif name == "replace_to_N_q":
            o.replace_to_N_q = _strtou(val)
        elif name == "1":
            o.input_read1_file = val
        elif name == "2":
            o.input_read2_file = val
        elif name == "u":
            o.input_unpaired_file = val
        elif name == "d":
            o.output_dir = val
        elif name == "m":
            o.kmer = _strtou(val)
        elif name == "n":
            o.max_num_poly_N = _strtou(val)
        elif name == "q":
            o.quality = int(val)
        elif name == "t":
            o.num_thread = _strtou(val)
        elif name in ("h", "?"):
            o.print_usage = True
    if o.print_usage:
        return o
    if o.version:
        o.messages.append("Version: " + FAQCS_VERSION)
        o.print_usage = True
        return o
    if bool(o.input_read1_file) != bool(o.input_read2_file):  # options.cpp:506-518
        o.print_usage = True
        return o
    o.num_subsample *= 2  # options.cpp:519-523 (also for unpaired-only runs: quirk b)
    if not o.input_unpaired_file and not o.input_read1_file:
        o.print_usage = True
        return o
    if not (2 <= o.kmer <= 31) or not (0.0 <= o.low_complexity_cutoff_ratio <= 1.0) or \
            not (0.0 <= o.filterAdapterMismatchRate <= 1.0) or o.split_size == 0 or o.num_subsample == 0:
        o.print_usage = True
        return o
    if o.replace_N:
        o.messages.append('**Warning** "-substitue" is not currently implemented')
    if o.filter_adapter:
        o.adapter.extend(BUILTIN_ADAPTERS)
    if trim_polyA:
        o.adapter.append(POLYA)
    if o.filter_phiX:
        px = phix_sequence()
        o.adapter.append((PHI_X, px))
        o.adapter.append((PHI_X_COMPLEMENT, reverse_complement(px)))
    if o.artifact_file:
        o.adapter.extend(parse_artifact_file(o.artifact_file))
    d, p = o.output_dir, o.prefix
    if o.input_read1_file and o.input_read2_file:  # options.cpp:696-714
        o.trimmed_read1_file = o.trimmed_read1_file or d + "/" + p + ".1.trimmed.fastq"
        o.trimmed_read2_file = o.trimmed_read2_file or d + "/" + p + ".2.trimmed.fastq"
        o.trimmed_unpaired_file = o.trimmed_unpaired_file or d + "/" + p + ".unpaired.trimmed.fastq"
        o.trimmed_discard_file = o.trimmed_discard_file or d + "/" + p + ".discard.trimmed.fastq"
    if o.input_unpaired_file:
        o.trimmed_unpaired_file = o.trimmed_unpaired_file or d + "/" + p + ".unpaired.trimmed.fastq"
        o.trimmed_discard_file = o.trimmed_discard_file or d + "/" + p + ".discard.trimmed.fastq"
    o.plots_file = o.plots_file or d + "/" + p + "_qc_report.pdf"
    if not o.discard_output:
        o.trimmed_discard_file = ""
    elif not o.trimmed_discard_file:
        o.trimmed_discard_file = d + "/" + p + ".discard.fastq"
    o.stats_file = o.stats_file or d + "/" + p + ".stats.txt"
    banner = {MODE_HARD: "Hard trimming is used.", MODE_BWA: "Bwa trimming is used.",
              MODE_BWA_PLUS: "Bwa extension trimming is used."}
    if o.mode == -1:
        o.mode = MODE_BWA_PLUS
        if not o.qc_only:
            o.messages.append("Not recognized mode. Bwa extension trimming algorithm is used.")
    elif not o.qc_only:
        o.messages.append(banner[o.mode])
    return o
