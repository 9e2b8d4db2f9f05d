#!/usr/bin/env python3
"""Deterministic FASTQ fixture generators used by the golden vectors and the parity tests.

TEST INFRASTRUCTURE.  Nothing here reads /root/reference; the generators are pure numpy with a
fixed PCG64 seed so the same bytes come out here and on the GPU box.

Two recipes:

* ``adversarial(n_pairs, seed)`` -- SURVEY.md Appendix A recipe: variable length reads, N runs at the
  ends / in the middle, lower-case bases, poly-A / AT / ACG / poly-G reads, adapter read-through with
  substitutions, low-quality heads and tails, ids of the form ``@R<i>/<mate> extra``.
* ``headline(n_pairs, L, seed, adapter_frac)`` -- SURVEY.md section 8(d) recipe for the BASELINE
  configs: fixed length L, iid ACGT with 0.2 % N, Q-plateau with a Q2 tail, optional 5 % adapter
  read-through.
"""
import gzip
import os
import sys

import numpy as np

# The built-in adapter table of the reference CLI (options.cpp:583-625) is *data* the drop-in must
# reproduce; it lives in faqcs_amd/options.py.  The fixture recipe only needs a few of them.
ADV_ADAPTERS = [
    "TCGTATAACTTCGTATAATGTATGCTATACGAAGTTATTACG",  # cre-loxp-forward
    "GATCGGAAGAGCACACGTCTGAACTCCAGTCAC",  # Nextera-primer-adapter-1
    "GATCGGAAGAGCGTCGTGTAGGGAAAGAGTGT",  # Nextera-primer-adapter-2
    "CTGTCTCTTATACACATCTAGATGTGTATAAGAGACAG",  # Nextera-junction-adapter-1
    "A" * 30,
    "GGGGTAGTGTGGATCCTCCTCTAGGCAGTTGGGTTATTCTAGAAGCAGATGTGTTGGCTGTTTCTGAAACTCTGGAAAA",  # TruSeq-adapter-1
]
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, s, rate):
    s = s.copy()
    hit = rng.random(len(s)) < rate
    s[hit] = ACGT[rng.integers(0, 4, hit.sum())]
    return s


def _adv_read(rng, maxlen=150):
    u = rng.random()
    if u < 0.006:
        L = int(rng.integers(1, 13))
    elif u < 0.006 + 1.0 / 7.0:
        L = int(rng.integers(20, maxlen + 1))
    else:
        L = maxlen
    c = rng.random()
    if c < 0.03:
        seq = np.full(L, ord("A"), np.uint8)
    elif c < 0.06:
        seq = np.resize(np.frombuffer(b"AT", np.uint8), L).copy()
    elif c < 0.08:
        seq = np.resize(np.frombuffer(b"ACG", np.uint8), L).copy()
    elif c < 0.10:
        seq = np.full(L, ord("G"), np.uint8)
    else:
        seq = ACGT[rng.integers(0, 4, L)]
    # adapter read-through
    if L > 30 and rng.random() < 0.25:
        ad = np.frombuffer(ADV_ADAPTERS[int(rng.integers(0, len(ADV_ADAPTERS)))].encode(), np.uint8)
        ad = _mutate(rng, ad, 0.08)
        pos = int(rng.integers(0, L))
        n = min(len(ad), L - pos)
        seq[pos:pos + n] = ad[:n]
    # per-base N
    seq[rng.random(L) < 0.004] = ord("N")
    if rng.random() < 0.05 and L > 8:
        k = int(rng.integers(1, 5))
        p = int(rng.integers(0, L - k))
        seq[p:p + k] = ord("N")
    if rng.random() < 0.04:
        seq[: int(rng.integers(1, 4))] = ord("N")
    if rng.random() < 0.04:
        seq[L - int(rng.integers(1, 4)):] = ord("N")
    if rng.random() < 0.02:
        low = rng.random(L) < 0.3
        seq[low] |= 0x20
    # quality
    if rng.random() < 0.8:
        q = rng.integers(10, 42, L)
    else:
        q = rng.integers(0, 42, L)
    if rng.random() < 0.3:
        h = int(rng.integers(0, 7))
        q[:h] = rng.integers(0, 9, min(h, L))
    cut = int(rng.integers(0, L + 31))
    if cut < L:
        tail = np.array([2, 2, 2, 3, 5, 7, 12, 30])
        q[cut:] = tail[rng.integers(0, 8, L - cut)]
    return seq, (q + 33).astype(np.uint8)


def adversarial(n_pairs, seed=7, maxlen=150, id_prefix="R"):
    rng = np.random.Generator(np.random.PCG64(seed))
    r1, r2 = [], []
    for i in range(n_pairs):
        for mate, out in ((1, r1), (2, r2)):
            s, q = _adv_read(rng, maxlen)
            out.append((("@%s%d/%d extra" % (id_prefix, i, mate)).encode(), s.tobytes(), q.tobytes()))
    return r1, r2


def long_reads(seed=51, n_pairs=36, maxlen=8000):
    """Ragged reads of 20..8 000 bases from the adversarial recipe, plus hand-made ones: the longest read the drop-in takes (32 767),
    1 025 bases (one past the chunked kernels), terminal N runs and low-quality tails longer than a 64-base piece, a low-complexity
    read, lower-case stretches, an adapter near the end of a long read, and short reads in the same file."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r1, r2 = [], []
    for i in range(n_pairs):
        for mate, out in ((1, r1), (2, r2)):
            ml = int(rng.integers(1100, maxlen + 1)) if rng.random() < 0.7 else int(rng.integers(30, 1100))
            s, q = _adv_read(rng, ml)
            out.append((("@K%d/%d extra" % (i, mate)).encode(), s.tobytes(), q.tobytes()))

    def put(lst, k, s, q):
        lst[k] = (lst[k][0], bytes(s.tobytes()), bytes((np.asarray(q) + 33).astype(np.uint8).tobytes()))

    L = 32767
    s = ACGT[rng.integers(0, 4, L)]; q = rng.integers(12, 41, L); q[L - 700:] = 2
    put(r1, 3, s, q)
    L = 1025
    put(r2, 3, ACGT[rng.integers(0, 4, L)], rng.integers(20, 41, L))
    L = 2000
    s = ACGT[rng.integers(0, 4, L)]; s[:200] = ord("N"); s[L - 300:] = ord("N"); q = rng.integers(25, 41, L)
    put(r1, 5, s, q)
    L = 5000
    s = ACGT[rng.integers(0, 4, L)]; q = rng.integers(15, 41, L); q[2000:] = rng.integers(0, 5, L - 2000); q[:90] = 3
    put(r2, 5, s, q)
    L = 3000
    put(r1, 7, np.resize(np.frombuffer(b"AT", np.uint8), L).copy(), rng.integers(20, 41, L))
    L = 4100
    s = ACGT[rng.integers(0, 4, L)]; s[1000:1400] |= 0x20; s[2000:2002] = ord("N"); s[3000:3003] = ord("N")
    put(r2, 7, s, rng.integers(20, 41, L))
    L = 6000
    s = ACGT[rng.integers(0, 4, L)]
    ad = np.frombuffer(ADV_ADAPTERS[1].encode(), np.uint8)
    s[5800:5800 + len(ad)] = ad
    put(r1, 9, s, rng.integers(25, 41, L))
    L = 1500
    put(r2, 9, np.full(L, ord("N"), np.uint8), rng.integers(25, 41, L))
    L = 2500
    s = ACGT[rng.integers(0, 4, L)]; s[s == ord("G")] = ord("G"); q = rng.integers(2, 41, L)
    put(r1, 11, s, q)
    return r1, r2


def headline_arrays(n_reads, L=150, seed=20260101, adapter_frac=0.0, mate=1):
    """Vectorised section-8(d) generator: returns (seq[n,L] u8, qual[n,L] u8)."""
    rng = np.random.Generator(np.random.PCG64([seed, mate]))
    seq = ACGT[rng.integers(0, 4, (n_reads, L))]
    seq[rng.random((n_reads, L)) < 0.002] = ord("N")
    pos = np.arange(L)[None, :]
    q = rng.integers(30, 41, (n_reads, L))
    head = rng.integers(2, 38, (n_reads, 3))
    q[:, :3] = head
    b = rng.integers(L // 2, L + 41, (n_reads, 1))
    tail_is_q2 = rng.random((n_reads, 1)) < 0.7
    tail = np.where(tail_is_q2, 2, rng.integers(3, 16, (n_reads, L)))
    q = np.where(pos >= b, tail, q)
    if adapter_frac > 0:
        from faqcs_amd.options import BUILTIN_ADAPTERS, POLYA  # data table (options.cpp:583-625)
        ads = [a[1] for a in BUILTIN_ADAPTERS] + [POLYA[1]]
        rows = np.nonzero(rng.random(n_reads) < adapter_frac)[0]
        for r in rows:
            ad = np.frombuffer(ads[int(rng.integers(0, len(ads)))].encode(), np.uint8)
            ad = _mutate(rng, ad, 0.05)
            p = int(rng.integers(40, L - 9))
            n = min(len(ad), L - p)
            seq[r, p:p + n] = ad[:n]
    return np.ascontiguousarray(seq, dtype=np.uint8), np.ascontiguousarray(q + 33, dtype=np.uint8)


def headline(n_pairs, L=150, seed=20260101, adapter_frac=0.0):
    out = []
    for mate in (1, 2):
        s, q = headline_arrays(n_pairs, L, seed, adapter_frac, mate)
        out.append([(("@SYN:%d/%d" % (i, mate)).encode(), s[i].tobytes(), q[i].tobytes()) for i in range(n_pairs)])
    return out[0], out[1]


def write_fastq(path, reads):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "wb") as f:
        for d, s, q in reads:
            f.write(d + b"\n" + s + b"\n+\n" + q + b"\n")


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    out = []
    with op(path, "rb") as f:
        while True:
            d = f.readline()
            if not d:
                break
            s = f.readline()
            f.readline()
            q = f.readline()
            out.append((d.rstrip(b"\r\n"), s.rstrip(b"\r\n"), q.rstrip(b"\r\n")))
    return out


def materialise(name, outdir):
    """Write the named fixture pair into outdir (idempotent); returns (r1_path, r2_path)."""
    os.makedirs(outdir, exist_ok=True)
    p1 = os.path.join(outdir, name + "_1.fastq")
    p2 = os.path.join(outdir, name + "_2.fastq")
    if os.path.exists(p1) and os.path.exists(p2):
        return p1, p2
    if name == "adv":
        r1, r2 = adversarial(1500, seed=7)
    elif name == "advbig":  # > 32768 reads per mate: exercises buffer boundaries / tail groups / epochs
        r1, r2 = adversarial(34000, seed=11)
    elif name == "nextseq":
        r1, r2 = adversarial(300, seed=3, id_prefix="NS500:")
    elif name == "head150":
        r1, r2 = headline(3000, 150, adapter_frac=0.05)
    elif name == "head250":
        r1, r2 = headline(1200, 250)
    elif name == "adv64":  # the adversarial set re-encoded as Phred+64 (pre-1.8 Illumina)
        r1, r2 = adversarial(600, seed=41)
        up = bytes((min(255, c + 31) if c >= 33 else c) for c in range(256))
        r1 = [(d, sq, q.translate(up)) for d, sq, q in r1]
        r2 = [(d, sq, q.translate(up)) for d, sq, q in r2]
    elif name == "errq":  # a quality byte above Q41: the reference throws from quality_score() (fastq.h:31-33)
        r1, r2 = adversarial(200, seed=31)
        d, sq, q = r1[57]
        r1[57] = (d, sq, q[:3] + b"~" + q[4:])
    elif name == "errbase":  # a base the adapter aligner's na_to_bits() rejects (seq_overlap.cpp:372-411)
        r1, r2 = adversarial(200, seed=37)
        d, sq, q = r2[91]
        r2[91] = (d, sq[:5] + b"*" + sq[6:], q)
    elif name == "errid":  # mate ids disagree at record 20 (FaQCs.cpp:370-389)
        r1, r2 = adversarial(200, seed=33)
        r2[20] = (b"@OTHER20/2 extra", r2[20][1], r2[20][2])
    elif name == "errcount":  # read 2 file is shorter
        r1, r2 = adversarial(200, seed=35)
        r2 = r2[:150]
    elif name == "long300":  # MiSeq 2x300: past the 256-base row kernels
        r1, r2 = adversarial(700, seed=21, maxlen=300, id_prefix="M")
    elif name == "long1000":  # ragged 20..1000-base reads (Ion Torrent / 454 lengths)
        r1, r2 = adversarial(260, seed=23, maxlen=1000, id_prefix="L")
    elif name == "long8k":  # reads past 1 024 bases (long-read platforms, contigs): the one-wave-per-read kernels
        r1, r2 = long_reads(seed=51)
    else:
        raise KeyError(name)
    write_fastq(p1, r1)
    write_fastq(p2, r2)
    return p1, p2


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
    for n in sys.argv[2:]:
        print(materialise(n, sys.argv[1]))
