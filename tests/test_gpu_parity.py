"""GPU parity tests proper (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
libfaqcs_mi.so, against (1) the committed golden vectors produced by the real reference and (2) the CPU
oracle on seeded random batches, bit-exact; plus size-independent properties at larger sizes."""
import os

import numpy as np
import pytest

from faqcs_amd import _capi as capi
from faqcs_amd.options import parse_args

pytestmark = pytest.mark.gpu

# FAQCS_TEST_SEED=<n> shifts every seeded random batch (fuzzing runs; the default 0 is what the suite is pinned to)
SEED = int(__import__("os").environ.get("FAQCS_TEST_SEED", "0"))


def hip_factory(opt, max_read_length, in_off):
    from faqcs_amd.engine import HipEngine

    return HipEngine(opt, max_read_length, in_off, kmer_table_slots=1 << 24)


def _native_loaded():
    with open("/proc/self/maps") as f:
        return "libfaqcs_mi.so" in f.read()


def random_batch(rng, n, maxlen, kind):
    """Ragged random reads exercising every branch: N runs at ends / inside, lower case, low-complexity,
    adapters, empty and tiny reads, low-quality heads / tails."""
    import make_fixtures

    reads = []
    for _ in range(n):
        if kind == "adv":
            s, q = make_fixtures._adv_read(rng, maxlen)
        elif kind == "clean":  # random bases, good qualities: every 31-mer of a read is counted, nearly all of them distinct
            s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, maxlen)].copy()
            q = (rng.integers(30, 41, maxlen) + 33).astype(np.uint8)
        else:
            L = int(rng.integers(0, maxlen + 1))
            s = np.frombuffer(b"ACGTNacgtnRY", np.uint8)[rng.choice(12, L, p=[.22, .22, .22, .22, .05, .02, .02, .01, .01, .005, .0025, .0025])].copy()
            q = (rng.integers(0, 42, L) + 33).astype(np.uint8)
            if L and rng.random() < 0.5:
                cut = int(rng.integers(0, L))
                q[cut:] = 33 + rng.integers(0, 6, L - cut)
        reads.append((b"@x", s.tobytes(), q.tobytes()))
    return reads


def compare_engines(opt, reads, R=256, in_off=33, seg_size=None):
    from oracle_engine import OracleEngine

    from faqcs_amd import driver

    seg_size = seg_size or len(reads)
    bufs = [reads[i:i + seg_size] for i in range(0, len(reads), seg_size)] or [[]]
    seq, qual, offset, seg = driver.pack_segments(bufs)
    hip = hip_factory(opt, R, in_off)
    ora = OracleEngine(opt, R, in_off)
    r1 = hip.process(seq, qual, offset, seg)
    r2 = ora.process(seq, qual, offset, seg)
    bad = np.nonzero(r1 != r2)[0]
    assert len(bad) == 0, "first differing read %d: hip=%s oracle=%s seq=%r qual=%r" % (
        bad[0], r1[bad[0]], r2[bad[0]], reads[bad[0]][1], reads[bad[0]][2])
    c1, c2 = hip.counters(), ora.counters()
    if not (c1 == c2).all():
        lay = capi.python_layout(R, hip.holder.n_adapters)
        ks = np.nonzero(c1 != c2)[0]
        what = []
        for k in ks[:8]:
            k = int(k)
            name = [nm for nm, v in lay.items() if nm != "total" and v[0] <= k < v[0] + v[1]][0]
            what.append("%s[%d]: hip=%d oracle=%d" % (name, k - lay[name][0], c1[k], c2[k]))
        import ctypes as C
        dbg = (C.c_uint64 * 4)()
        hip.lib.faqcs_debug_words(hip.ctx, dbg, 4)
        raise AssertionError("counter block differs in %d places: %s ; debug words %s" % (len(ks), "; ".join(what), [hex(int(x)) for x in dbg]))
    assert _native_loaded()
    return hip, ora


OPTION_SETS = [
    [],
    ["--mode", "BWA"],
    ["--mode", "HARD", "-q", "10"],
    ["--mode", "HARD", "-q", "10", "--5trim_off"],
    ["--5trim_off"],
    ["-q", "20", "--min_L", "30"],
    ["-q", "0", "--min_L", "1"],
    ["-q", "41"],
    ["--5end", "3", "--3end", "5"],
    ["--5end", "60", "--3end", "100", "--min_L", "10"],
    ["--avg_q", "25"],
    ["--avg_q", "12.5", "-n", "1"],
    ["-n", "0"],
    ["-n", "3"],
    ["-n", "7", "--min_L", "5"],
    ["--lc", "0.5"],
    ["--lc", "0.2", "--min_L", "5"],
    ["--lc", "1.0"],
    ["--replace_to_N_q", "15"],
    ["--replace_to_N_q", "30", "--lc", "0.4", "-n", "4"],
    ["--out_ascii", "64"],
    ["--qc_only"],
    ["--adapter"],
    ["--adapter", "--polyA"],
    ["--adapter", "--polyA", "--rate", "0.3", "--qc_only"],
    ["--adapter", "--5end", "4", "--3end", "2", "--min_L", "20"],
]


@pytest.mark.parametrize("args", OPTION_SETS, ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("kind,maxlen", [("adv", 150), ("ragged", 64), ("ragged", 250), ("adv", 100)])
def test_random_batches_match_oracle(args, kind, maxlen):
    rng = np.random.Generator(np.random.PCG64([len(kind), maxlen, OPTION_SETS.index(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    n = 700 if "--adapter" in args else 3000
    reads = random_batch(rng, n, maxlen, kind)
    compare_engines(opt, reads, seg_size=517)  # ragged segments: tail groups of the adapter pre-pass


@pytest.mark.parametrize("args", OPTION_SETS, ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("kind,maxlen", [("adv", 300), ("ragged", 500), ("ragged", 700), ("adv", 1024)])
def test_long_reads_match_oracle(args, kind, maxlen):
    """Reads past 256 bases: 32 lanes per read (widths 320 / 512) and the whole wave per read (768 / 1024)."""
    rng = np.random.Generator(np.random.PCG64([7, len(kind), maxlen, OPTION_SETS.index(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    n = 150 if "--adapter" in args else 500
    reads = random_batch(rng, n, maxlen, kind)
    compare_engines(opt, reads, R=1024, seg_size=117)


@pytest.mark.parametrize("maxlen", [150, 300])
@pytest.mark.parametrize("args", [["--phiX"], ["--adapter", "--polyA", "--phiX", "--min_L", "20"]], ids=["phix", "adapter+phix"])
def test_phix_reads_match_oracle(args, maxlen):
    """Long targets (2 x 5 386-base PhiX): the sliding-window prefilter and the exact stage on true PhiX reads (both strands,
    with substitutions, partial overlaps at the genome ends) mixed with random reads."""
    import make_fixtures

    from faqcs_amd import options

    rng = np.random.Generator(np.random.PCG64([11, len(args), SEED, maxlen]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    phix = np.frombuffer(options.phix_sequence().encode(), np.uint8) if hasattr(options, "phix_sequence") else None
    if phix is None:
        phix = np.frombuffer([a for a in parse_args(["-u", "x", "-d", "y", "--phiX"]).adapter if "phi" in a[0].lower()][0][1].encode(), np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    reads = []
    for i in range(600):
        if i % 3 == 0:
            s, q = make_fixtures._adv_read(rng, maxlen)
        else:
            L = int(rng.integers(30, maxlen + 1))
            p0 = int(rng.integers(-40, len(phix) - L + 40))
            lo = min(max(p0, 0), len(phix))
            hi = max(lo, min(p0 + L, len(phix)))
            s = np.concatenate([make_fixtures.ACGT[rng.integers(0, 4, max(0, min(L, lo - p0)))], phix[lo:hi]])
            s = np.concatenate([s, make_fixtures.ACGT[rng.integers(0, 4, L - len(s))]])
            if i % 2:
                s = np.frombuffer(s.tobytes().translate(comp)[::-1], np.uint8)
            s = make_fixtures._mutate(rng, s, float(rng.choice([0.0, 0.03, 0.15, 0.3])))
            q = (rng.integers(20, 41, L) + 33).astype(np.uint8)
        reads.append((b"@p", bytes(s.tobytes()), bytes(q.tobytes())))
    compare_engines(opt, reads, R=256 if maxlen <= 256 else 1024, seg_size=211)


def test_longest_read():
    """Exactly FAQCS_MAX_READ_LENGTH (32 767) bases is accepted, one more is refused loudly (no silent truncation)."""
    from faqcs_amd.engine import FaqcsError

    L = capi.MAX_READ_LENGTH
    rng = np.random.Generator(np.random.PCG64([L, SEED]))
    s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
    q = rng.integers(5, 42, L) + 33
    q[L - 900:] = 35
    reads = [(b"@e", s.tobytes(), q.astype(np.uint8).tobytes()), (b"@e", b"N" * L, bytes([33 + 20] * L)), (b"@e", b"ACGT" * 30, bytes([70] * 120))]
    for args in ([], ["--adapter", "--polyA"], ["--mode", "BWA", "--min_L", "1"], ["--kmer_rarefaction", "--split_size", "1"]):
        hip, ora = compare_engines(parse_args(["-u", "x", "-d", "y"] + args), reads, R=L, seg_size=1)
        if "--kmer_rarefaction" in args:
            hip.kmer_end_table()
            ora.kmer_end_table()
            assert len(hip.kmer_points()) == 3 and (hip.kmer_points() == ora.kmer_points()).all()
            h1, h2 = hip.kmer_histogram(), ora.kmer_histogram()
            assert (h1[0] == h2[0]).all() and (h1[1] == h2[1]).all()
    with pytest.raises(FaqcsError) as ei:
        compare_engines(parse_args(["-u", "x", "-d", "y"]), [(b"@e", b"A" * (L + 1), bytes([70] * (L + 1)))], R=L)
    assert ei.value.code == capi.E_INVAL


@pytest.mark.parametrize("args", OPTION_SETS, ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("kind,maxlen", [("adv", 1025), ("ragged", 3000), ("ragged", 6000), ("adv", 9000)])
def test_reads_past_1024_bases_match_oracle(args, kind, maxlen):
    """Batches that hold a read of more than 1 024 bases: trim_long and adapter_overlap<1, 32768> (one wave per read) against the
    oracle, every option set; short reads ride in the same batches."""
    rng = np.random.Generator(np.random.PCG64([13, len(kind), maxlen, OPTION_SETS.index(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    n = 40 if "--adapter" in args else 120
    reads = random_batch(rng, n, maxlen, kind)
    if kind == "adv":  # (the adversarial recipe makes most reads `maxlen` long: mix in every scale)
        reads += random_batch(rng, n // 2, 150, "adv") + random_batch(rng, n // 4, maxlen // 3, "adv")
    hip, _ = compare_engines(opt, reads, R=16384, seg_size=37)
    import ctypes as C

    kt = capi.KernelTimes()
    hip.lib.faqcs_kernel_report(hip.ctx, C.byref(kt))
    assert (kt.trim_kernel or b"").decode() == "trim_long"


@pytest.mark.parametrize("args", [[], ["-q", "2"], ["-q", "20"], ["--mode", "BWA", "-q", "12"], ["--5trim_off", "-q", "10"], ["--mode", "HARD", "-q", "7"],
                                  ["-q", "15", "--5end", "37", "--3end", "101"]], ids=lambda a: " ".join(a) or "default")
def test_long_reads_with_long_low_quality_ends_match_oracle(args):
    """trim_long takes a whole 64-position piece of a low-quality head / tail in ONE step of its BWA_plus / BWA walk when none of the
    piece's scores exceeds Q (the area is monotone there).  Reads of 1 100 ... 6 000 bases whose ends are low-quality stretches of
    0 ... 3 000 bases -- at, just under and just over the threshold, with single good bases sprinkled in, of every alignment
    against the 64-position pieces -- against the oracle's base-by-base walk."""
    rng = np.random.Generator(np.random.PCG64([23, len(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1"] + args)
    Q = opt.quality
    reads = []
    for k in range(160):
        L = int(rng.integers(1100, 6001))
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        q = rng.integers(Q + 1, 42, L) if Q < 41 else np.full(L, 41)
        for side in (0, 1):
            n = int(rng.integers(0, 3001)) if rng.random() < 0.8 else int(rng.integers(0, 130))
            n = min(n, L)
            kind = rng.random()
            if kind < 0.4:
                low = rng.integers(0, Q + 1, n)                 # every score <= Q
            elif kind < 0.7:
                low = np.full(n, Q)                                # exactly at the threshold: the area does not move
            else:
                low = rng.integers(0, Q + 1, n)
                hit = rng.random(n) < 0.01
                low[hit] = rng.integers(Q + 1, 42, int(hit.sum())) if Q < 41 else 41  # a good base every ~100: the fast path must give way there
            if side == 0:
                q[:n] = low
            else:
                q[L - n:] = low
        if rng.random() < 0.1:
            s[: int(rng.integers(1, 200))] = ord("N")
        reads.append((b"@t%d" % k, s.tobytes(), (q + 33).astype(np.uint8).tobytes()))
    compare_engines(opt, reads, R=8192, seg_size=41)


@pytest.mark.parametrize("args", OPTION_SETS, ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("kind,maxlen", [("adv", 150), ("ragged", 250), ("adv", 700)])
def test_trim_long_equals_the_chunked_kernels_on_short_reads(args, kind, maxlen, monkeypatch):
    """FAQCS_TRIM_LONG=1 sends every batch to trim_long: the same results as the oracle (and therefore as the chunked kernels) on the
    shapes the rest of the suite runs, where a wave's 64 lanes mostly idle past the read's end and several reads share a 64-base piece."""
    monkeypatch.setenv("FAQCS_TRIM_LONG", "1")
    rng = np.random.Generator(np.random.PCG64([17, len(kind), maxlen, OPTION_SETS.index(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    reads = random_batch(rng, 300 if "--adapter" in args else 1200, maxlen, kind)
    compare_engines(opt, reads, R=256 if maxlen <= 256 else 1024, seg_size=211)


def test_long_read_limits():
    """The chunked kernels' own limit: exactly 1 024 bases stays on them, constant qualities included."""
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1"])
    L = capi.FAST_READ_LENGTH
    ok = [(b"@e", (b"ACGT" * 300)[:L], bytes([33 + 38] * (L - 40) + [35] * 40)), (b"@e", b"N" * L, bytes([33 + 20] * L))]
    compare_engines(opt, ok, R=L)
    # constant quality (FASTA converted to FASTQ): every read hits the same position x quality cells, the worst case for
    # the 8-bit packed LDS counters of the 1024-wide kernel
    compare_engines(opt, [(b"@e", (b"ACGGT" * 205)[:L], bytes([33 + 40] * L))] * 1500 + [(b"@e", b"ACGT" * 150, bytes([33 + 40] * 600))] * 300, R=L)


@pytest.mark.parametrize("args", [OPTION_SETS[i] for i in (0, 1, 2, 8, 10, 13, 15, 18, 21, 23)], ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("kind,maxlen", [("adv", 157), ("adv", 120), ("ragged", 104), ("adv", 200), ("ragged", 112), ("adv", 75), ("ragged", 50),
                                         ("adv", 252), ("adv", 256), ("ragged", 224), ("adv", 300), ("ragged", 304), ("adv", 320), ("adv", 52), ("ragged", 30), ("adv", 53)])
def test_every_kernel_width_matches_oracle(args, kind, maxlen):
    """One batch per position-slot width the dispatcher can pick (4 lanes per read: C = 16/19; 8 lanes: C = 13/16/19/20; 16 lanes: C = 13/16;
    trim_lds with 16 lanes per read up to 252 bases, trim_filter_accumulate past that and for multiples of 32)."""
    rng = np.random.Generator(np.random.PCG64([3, len(kind), maxlen, OPTION_SETS.index(args), SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    reads = random_batch(rng, 500 if "--adapter" in args else 1500, maxlen, kind)
    compare_engines(opt, reads, R=256 if maxlen <= 256 else 1024, seg_size=389)


@pytest.mark.parametrize("in_off", [40, 64, 90, 100])
@pytest.mark.parametrize("kind,maxlen", [("adv", 150), ("ragged", 100), ("adv", 70), ("adv", 250), ("ragged", 251), ("adv", 300)])
def test_quality_offsets_that_leave_scores_outside_the_valid_range(in_off, kind, maxlen):
    """Raw quality bytes are generated for Phred+33; decoding them with a larger offset makes most scores negative (clamped to
    0 by the trimmers, not by the averages) -- the per-position exact pass of the two-phase kernel instead of its four-bytes-per-
    instruction sums.  Offsets above 86 switch the packed range check off altogether."""
    rng = np.random.Generator(np.random.PCG64([11, in_off, maxlen, SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", str(in_off), "--min_L", "20"])
    compare_engines(opt, random_batch(rng, 1200, maxlen, kind), R=256 if maxlen <= 256 else 1024, in_off=in_off, seg_size=500)


@pytest.mark.parametrize("switch", ["tpr_off", "lds_off"])
def test_single_pass_kernel_variants_still_match_oracle(switch):
    """Reads of up to 304 bases run trim_lds.  FAQCS_TRIM_LDS=0 (read once per process) sends the same batches through the single-pass kernel
    trim_filter_accumulate -- what --replace_to_N_q and reads of 305 ... 1 024 bases use.  (Round 1's two-phase trim_tpr is compiled in only with
    -DFAQCS_WITH_TRIM_TPR since round 6; in such a build the first of the two runs goes through it, FAQCS_TRIM_TPR=0 switches it off again.)"""
    import subprocess
    import sys
    env = dict(os.environ, FAQCS_TRIM_LDS="0", **({"FAQCS_TRIM_TPR": "0"} if switch == "tpr_off" else {}))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "test_every_kernel_width_matches_oracle or test_edge_reads or test_quality_error_is_reported"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]


def test_edge_reads():
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1", "--adapter", "--polyA"])
    Q = lambda s: bytes([33 + c for c in s])  # noqa: E731
    reads = [
        (b"@e", b"", b""),
        (b"@e", b"A", Q([40])),
        (b"@e", b"N", Q([40])),
        (b"@e", b"NN", Q([40, 40])),
        (b"@e", b"NNNNNNNNNN", Q([30] * 10)),
        (b"@e", b"ACGT", Q([2, 2, 2, 2])),
        (b"@e", b"ACGTA", Q([41, 0, 41, 0, 41])),
        (b"@e", b"ACGTAC", Q([0, 0, 0, 41, 41, 41])),
        (b"@e", b"A" * 150, Q([38] * 150)),
        (b"@e", b"AT" * 75, Q([38] * 150)),
        (b"@e", b"ACG" * 50, Q([38] * 150)),
        (b"@e", b"G" * 150, Q([38] * 150)),
        (b"@e", b"N" + b"ACGT" * 37 + b"N", Q([38] * 150)),
        (b"@e", b"acgtn" * 30, Q([38] * 150)),
        (b"@e", (b"ACGT" * 64)[:250], Q([38] * 250)),
        (b"@e", (b"ACGT" * 64), Q(([38] * 200) + [2] * 56)),
    ]
    for k in range(1, 12):
        reads.append((b"@e", (b"ACGGTCA" * 3)[:k], Q([2 + 3 * (i % 7) for i in range(k)])))
    compare_engines(opt, reads)


def test_quality_error_is_reported():
    from faqcs_amd.engine import FaqcsError

    opt = parse_args(["-u", "x", "-d", "y"])
    reads = [(b"@e", b"ACGT" * 10, bytes([33 + 30] * 39 + [33 + 42]))]
    with pytest.raises(FaqcsError) as ei:
        compare_engines(opt, reads)
    assert ei.value.code == capi.E_QUALITY


@pytest.mark.parametrize("name", __import__("golden_util").case_names())
def test_golden_cases_on_gpu(name, fixture_cache, tmp_path):
    """The reference's own outputs (QC.stats.txt, trimmed FASTQ, --debug tables) reproduced by the HIP path."""
    from golden_util import case_max_read_length, load_case, run_case

    case = load_case(name)
    bad = run_case(case, fixture_cache, tmp_path, hip_factory, max_read_length=case_max_read_length(case))
    assert not bad, "\n".join(bad)
    assert case["exit_code"] != 0 or _native_loaded()  # (an input error can stop the run before any read reaches the engine)


def test_kmer_matches_oracle():
    rng = np.random.Generator(np.random.PCG64(99 + SEED))
    import make_fixtures

    for args in (["--kmer_rarefaction", "--split_size", "300"], ["--kmer_rarefaction", "--split_size", "400", "--qc_only", "--subset", "2"],
                 ["--kmer_rarefaction", "--split_size", "250", "-m", "5", "--replace_to_N_q", "20"],
                 ["--kmer_rarefaction", "--split_size", "150", "-m", "17", "--subset", "30"], ["--kmer_rarefaction", "--split_size", "500", "-m", "2"]):
        opt = parse_args(["-u", "x", "-d", "y"] + args)
        reads = random_batch(rng, 2500, 150, "adv")
        hip, ora = compare_engines(opt, reads, seg_size=333)
        hip.kmer_end_table()
        ora.kmer_end_table()
        assert (hip.kmer_points() == ora.kmer_points()).all()
        h1, h2 = hip.kmer_histogram(), ora.kmer_histogram()
        assert (h1[0] == h2[0]).all() and (h1[1] == h2[1]).all()


def _genome_reads(rng, genome, n, lo, hi, n_rate=0.002, q_hi=41):
    """Windows of `genome` on either strand, 0.5 % substitutions, an N now and then, good qualities (most of a read is kept)."""
    comp = np.zeros(256, np.uint8)
    comp[np.frombuffer(b"ACGTN", np.uint8)] = np.frombuffer(b"TGCAN", np.uint8)
    reads = []
    for _ in range(n):
        L = int(rng.integers(lo, hi + 1))
        at = int(rng.integers(0, len(genome) - L))
        s = genome[at:at + L].copy()
        sub = rng.random(L) < 0.005
        s[sub] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(sub.sum()))]
        s[rng.random(L) < n_rate] = ord("N")
        if rng.random() < 0.5:
            s = comp[s[::-1]]
        q = (rng.integers(30, q_hi, L) + 33).astype(np.uint8)
        reads.append((b"@g", s.tobytes(), q.tobytes()))
    return reads


def _kmer_engines_agree(hip, ora):
    hip.kmer_end_table()
    ora.kmer_end_table()
    assert len(ora.kmer_points()) > 0 and (hip.kmer_points() == ora.kmer_points()).all()
    h1, h2 = hip.kmer_histogram(), ora.kmer_histogram()
    assert (h1[0] == h2[0]).all() and (h1[1] == h2[1]).all()


@pytest.mark.parametrize("seed", range(3))
def test_kmer_submissions_either_side_of_256_bases_share_their_keys(seed):
    """one engine, several submissions whose longest reads lie either side of 256 bases -- the 16-positions-per-lane
    extraction kernel, the general one and its multi-piece path all feed ONE table, so a k-mer seen by two of them must be one key
    (round 4's two extraction kernels wrote different integers for it).  Reads are windows of one small genome: nearly every k-mer
    of a submission occurs in the others too (trim.cpp:887-931; sampling points trim.cpp:157-185)."""
    from oracle_engine import OracleEngine

    from faqcs_amd import driver

    rng = np.random.Generator(np.random.PCG64([2560, seed, SEED]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 30000)]
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "700", "--subset", "12"])
    hip, ora = hip_factory(opt, 1024, 33), OracleEngine(opt, 1024, 33)
    for lo, hi in ((60, 150), (200, 700), (240, 256), (257, 300), (31, 40)):
        reads = _genome_reads(rng, genome, 1500, lo, hi)
        bufs = [reads[i:i + 409] for i in range(0, len(reads), 409)]
        seq, qual, offset, seg = driver.pack_segments(bufs)
        r1, r2 = hip.process(seq, qual, offset, seg), ora.process(seq, qual, offset, seg)
        assert (r1 == r2).all()
    _kmer_engines_agree(hip, ora)


def test_kmer_partition_counted_in_several_rounds(monkeypatch, capfd):
    """A partition with more distinct keys in one group than the combine kernel's LDS table takes (2 816 of 4 096 slots): the workgroup
    writes its table out, clears it and goes on, and the next write-out reads slots the first one stored.  4 000 reads share a 45-base
    core inside random flanks, the group is large enough (FAQCS_KMER_GROUP_ITEMS) for the partitions of the core's minimizers to keep
    their items, and FAQCS_KMER_DEBUG reports the fullest partition of every flush -- the test asserts that the case it is named
    after occurred.  Every sampling point and every histogram bin equals the oracle's (trim.cpp:157-185,887-931; FaQCs.cpp:518-521)."""
    import re

    rng = np.random.Generator(np.random.PCG64([9191, SEED]))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    core = acgt[rng.integers(0, 4, 45)]
    reads = []
    for i in range(4000):
        L = int(rng.integers(100, 251))
        s = acgt[rng.integers(0, 4, L)]
        at = int(rng.integers(0, L - 45))
        s[at:at + 45] = core
        reads.append((b"@r", s.tobytes(), (rng.integers(30, 41, L) + 33).astype(np.uint8).tobytes()))
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "1000", "--subset", "5", "--lc", "1.0", "--min_L", "31"])
    from faqcs_amd.engine import HipEngine

    monkeypatch.setenv("FAQCS_KMER_GROUP_ITEMS", str(1 << 29))  # (a partition's region: 2 600 items)
    monkeypatch.setenv("FAQCS_KMER_DEBUG", "1")
    monkeypatch.setenv("FAQCS_KMER_STATS", "1")
    monkeypatch.setattr("test_gpu_parity.hip_factory", lambda o, r, q: HipEngine(o, r, q, kmer_table_slots=1 << 24))
    hip, ora = compare_engines(opt, reads, seg_size=1000)
    _kmer_engines_agree(hip, ora)
    err = capfd.readouterr().err
    fullest = [int(m) for m in re.findall(r"fullest partition (\d+) keys", err)]
    assert fullest and max(fullest) > 2816, fullest
    # (round 6: the pass was counted in one piece, and the partitions that did not fit one LDS round went through their table slices)
    st = _kmer_stats(err)
    assert len(st) == 1 and st[0][2] and st[0][0] >= 1, st


@pytest.mark.parametrize("seed", range(3))
def test_kmer_repeats_and_a_partition_larger_than_its_slice(seed, monkeypatch):
    """The rare paths of the super-k-mer kernels: (1) tandem repeats and homopolymers -- equal minimizers over more than 17 k-mers: the
    16-positions-per-lane kernel leaves such a read to the general kernel, which cuts its runs on a grid; (2) thousands of reads that
    share a 45-base core with random flanks -- their k-mers share a handful of minimizers, so a few partitions hold far more
    distinct keys than the 64 slots of their slice of a 2^22-slot table: the keys move to the overflow area behind the table.
    Every sampling point and every histogram bin equals the oracle's (trim.cpp:157-185,887-931; FaQCs.cpp:518-521)."""
    rng = np.random.Generator(np.random.PCG64([9090, seed, SEED]))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    core = acgt[rng.integers(0, 4, 45)]
    reads = []
    for i in range(3600):
        L = int(rng.integers(80, 251))
        kind = i % 3
        if kind == 0:  # repeats
            s = np.zeros(0, np.uint8)
            while len(s) < L:
                unit = acgt[rng.integers(0, 4, int(rng.integers(1, 7)))]
                s = np.concatenate([s, np.tile(unit, int(rng.integers(8, 60)))])
            s = s[:L].copy()
        else:  # the shared core somewhere inside random flanks
            s = acgt[rng.integers(0, 4, L)]
            at = int(rng.integers(0, L - 45))
            s[at:at + 45] = core
        if rng.random() < 0.2:
            s[int(rng.integers(0, L))] = ord("N")
        reads.append((b"@r", s.tobytes(), (rng.integers(30, 41, L) + 33).astype(np.uint8).tobytes()))
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "500", "--subset", "5", "--lc", "1.0", "--min_L", "31"])
    from faqcs_amd.engine import HipEngine

    monkeypatch.setattr("test_gpu_parity.hip_factory", lambda o, r, q: HipEngine(o, r, q, kmer_table_slots=1 << 22))
    hip, ora = compare_engines(opt, reads, seg_size=431)
    _kmer_engines_agree(hip, ora)


def _kmer_stats(err):
    """(fine partitions that went through their table slices, inserts into the overflow area, 'in one piece'?) of every pass FAQCS_KMER_STATS reported"""
    import re

    return [(int(a), int(b), "in one piece" in line) for line in err.splitlines() if line.startswith("[kmer stats]")
            for a, b in re.findall(r"; (\d+) of \d+ fine partitions through their table slices; (\d+) inserts", line)]


@pytest.mark.parametrize("fine", [0, 1, 2, 3])
def test_kmer_pass_in_one_piece_for_every_partition_width(fine, monkeypatch, capfd):
    """Round 6: a pass that fits the group buffers is counted when it ENDS, fine partition by fine partition, without the table
    (skm_split_sort + skm_combine<KS_COUNT>).  The table's size sets the number of fine partitions, 2^(16 + F); here every F on one
    small table (FAQCS_KMER_FINE_BITS), two passes on one engine (the second must start from nothing), reads of a small genome with
    N and either strand.  Points and the whole histogram of counts equal the oracle's (trim.cpp:157-185,887-931; FaQCs.cpp:518-537)."""
    from oracle_engine import OracleEngine

    from faqcs_amd import driver
    from faqcs_amd.engine import HipEngine

    monkeypatch.setenv("FAQCS_KMER_FINE_BITS", str(fine))
    monkeypatch.setenv("FAQCS_KMER_STATS", "1")
    rng = np.random.Generator(np.random.PCG64([606, fine, SEED]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 60000)]
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "900", "--subset", "9"])
    hip, ora = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 25), OracleEngine(opt, 256, 33)
    for n_pass in range(2):
        for lo, hi in ((100, 250), (31, 60), (180, 256)):
            reads = _genome_reads(rng, genome, 2500, lo, hi)
            seq, qual, offset, seg = driver.pack_segments([reads[i:i + 611] for i in range(0, len(reads), 611)])
            assert (hip.process(seq, qual, offset, seg) == ora.process(seq, qual, offset, seg)).all()
        _kmer_engines_agree(hip, ora)
    st = _kmer_stats(capfd.readouterr().err)
    assert len(st) == 2 and all(one_piece for _, _, one_piece in st), st


def test_kmer_pass_of_many_submissions_stays_in_one_piece(monkeypatch, capfd):
    """faqcs_mi submits a launch per 32 768-read buffer: a pass of 50 M reads is 1 500 submissions.  Round 5's groups took 1 000 extraction
    launches (the item's 10-bit run field) and charged every launch 64 items of slack per sub-region, so such a pass was cut into groups --
    through the table -- whatever the buffers could hold.  An item now carries its epoch relative to the group's first, and a sub-region's
    slack is added once: 1 300 submissions of 40 reads each are ONE group, counted in one piece, equal to the oracle."""
    from oracle_engine import OracleEngine

    from faqcs_amd import driver

    monkeypatch.setenv("FAQCS_KMER_STATS", "1")
    rng = np.random.Generator(np.random.PCG64([909, SEED]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50000)]
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "5000", "--subset", "7"])
    hip, ora = hip_factory(opt, 256, 33), OracleEngine(opt, 256, 33)
    reads = _genome_reads(rng, genome, 1300 * 40, 60, 150)
    for i in range(0, len(reads), 40):
        seq, qual, offset, seg = driver.pack_segments([reads[i:i + 40]])
        assert (hip.process(seq, qual, offset, seg) == ora.process(seq, qual, offset, seg)).all()
    _kmer_engines_agree(hip, ora)
    st = _kmer_stats(capfd.readouterr().err)
    assert len(st) == 1 and st[0][2], st


def test_kmer_curve_asked_for_in_the_middle_of_a_pass(monkeypatch, capfd):
    """faqcs_kmer_points() / _totals() while the pass goes on: the open group has to go INTO THE TABLE (its keys must live somewhere), and
    the rest of the pass -- and its end -- then goes through the table too; the answers on the way and at the end equal the oracle's.
    Then a second pass on the same engine, not interrupted: counted in one piece again."""
    from oracle_engine import OracleEngine

    from faqcs_amd import driver

    monkeypatch.setenv("FAQCS_KMER_STATS", "1")
    rng = np.random.Generator(np.random.PCG64([707, SEED]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 40000)]
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "700", "--subset", "20"])
    hip, ora = hip_factory(opt, 256, 33), OracleEngine(opt, 256, 33)
    for n_pass in range(2):
        for part in range(4):
            reads = _genome_reads(rng, genome, 1800, 60, 250)
            seq, qual, offset, seg = driver.pack_segments([reads[i:i + 450] for i in range(0, len(reads), 450)])
            assert (hip.process(seq, qual, offset, seg) == ora.process(seq, qual, offset, seg)).all()
            if n_pass == 0 and part in (0, 2):
                assert (hip.kmer_points() == ora.kmer_points()).all() and hip.kmer_totals() == ora.kmer_totals()
        _kmer_engines_agree(hip, ora)
    st = _kmer_stats(capfd.readouterr().err)
    assert [one_piece for _, _, one_piece in st] == [False, True], st


def test_kmer_pass_that_has_been_counted_takes_no_more_reads():
    """faqcs_kmer_finish_pass(): the pass is complete and counted; a submission with k-mers is refused (FAQCS_E_INVAL) until
    faqcs_kmer_end_table() has started the next pass -- it cannot silently join keys that are no longer anywhere."""
    from faqcs_amd import driver
    from faqcs_amd.engine import FaqcsError

    rng = np.random.Generator(np.random.PCG64([808, SEED]))
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20000)]
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "500", "--subset", "4"])
    hip = hip_factory(opt, 256, 33)
    seq, qual, offset, seg = driver.pack_segments([_genome_reads(rng, genome, 800, 60, 200)])
    hip.process(seq, qual, offset, seg)
    hip.kmer_finish_pass()
    pts = hip.kmer_points().copy()
    assert len(pts) == 1 and pts[0]["distinct_kmer"] > 0
    with pytest.raises(FaqcsError) as e:
        hip.process(seq, qual, offset, seg)
    assert e.value.code == capi.E_INVAL
    hip.kmer_end_table()
    assert (hip.kmer_points() == pts).all() and hip.kmer_totals() == (int(pts[0]["distinct_kmer"]), int(pts[0]["total_kmer"]))
    hip.process(seq, qual, offset, seg)  # the next pass takes reads again (the curve is complete: nothing is counted, trim.cpp:180-184)


@pytest.mark.timeout(120)
def test_kmer_table_that_is_too_small_is_an_error_at_once(monkeypatch):
    """Three times more distinct k-mers than the table (2^22 slots + its overflow area) can hold.  Round 6: a pass that fits the group
    buffers never touches the table, so it is counted all the same -- every one of its 15.4 M random 31-mers a key.  When the pass does go
    through the table (here: FAQCS_KMER_FINAL=0, round 5's path; in production: a pass larger than the buffers, or a curve asked for on
    the way) the answer is FAQCS_E_KMER_FULL at once -- not a scan of the whole overflow area for every key behind the first one that
    did not fit (the probe sequence there is cut and nothing is tried once the table has been declared full)."""
    from faqcs_amd.engine import FaqcsError, HipEngine

    rng = np.random.Generator(np.random.PCG64(5 + SEED))
    L, n = 250, 70000
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * L)]
    qual = np.full(n * L, 33 + 38, np.uint8)
    pad = np.zeros(64, np.uint8)
    s, q = np.concatenate([pad, seq, pad])[64:], np.concatenate([pad, qual, pad])[64:]
    off = (np.arange(n + 1, dtype=np.uint64) * L).astype(np.uint32)
    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", "100000"])
    eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 22)
    eng.process(s, q, off, np.array([0, n], dtype=np.uint32))
    eng.kmer_end_table()
    d, t = eng.kmer_totals()
    assert t == n * (L - 30) and (1 << 22) + (1 << 18) < d <= t  # (more keys than the table and its overflow area have slots)
    monkeypatch.setenv("FAQCS_KMER_FINAL", "0")
    eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 22)
    eng.process(s, q, off, np.array([0, n], dtype=np.uint32))
    with pytest.raises(FaqcsError) as e:
        eng.kmer_end_table()
    assert e.value.code == capi.E_KMER_FULL


@pytest.mark.parametrize("group_items,maxlen", [(1 << 14, 150), (1 << 16, 250), (1 << 20, 250), (1 << 16, 600)], ids=["g14", "g16", "g20", "g16_long"])
def test_kmer_groups_of_every_size_match_oracle(group_items, maxlen, monkeypatch):
    """The combine-before-insert path with groups far smaller than a submission (FAQCS_KMER_GROUP_ITEMS): runs are cut into many
    launches, groups are flushed in the middle of a segment, sub-regions overflow into the per-occurrence path -- and every point and
    every histogram bin still equals the oracle's (trim.cpp:157-185,887-931; FaQCs.cpp:518-521).  maxlen 600: the chunked extraction
    kernel (reads past 256 bases); the others: four positions per lane."""
    monkeypatch.setenv("FAQCS_KMER_GROUP_ITEMS", str(group_items))
    rng = np.random.Generator(np.random.PCG64(4321 + SEED))
    for args in (["--kmer_rarefaction", "--split_size", "700", "--subset", "6"], ["--kmer_rarefaction", "--split_size", "450", "-m", "11", "--qc_only"]):
        opt = parse_args(["-u", "x", "-d", "y"] + args)
        reads = random_batch(rng, 6000, maxlen, "adv")
        if maxlen > 256:
            hip, ora = compare_engines(opt, reads, R=1024, seg_size=977)
        else:
            hip, ora = compare_engines(opt, reads, seg_size=977)
        hip.kmer_end_table()
        ora.kmer_end_table()
        assert len(ora.kmer_points()) > 0 and (hip.kmer_points() == ora.kmer_points()).all()
        h1, h2 = hip.kmer_histogram(), ora.kmer_histogram()
        assert (h1[0] == h2[0]).all() and (h1[1] == h2[1]).all()


@pytest.mark.parametrize("genome,slots_log2,group_items", [(12_000_000, 29, None), (200_000_000, 30, 1 << 30)], ids=["40x_of_12Mbp", "distinct_heavy_one_group"])
def test_kmer_group_path_equals_the_per_occurrence_path_at_scale(genome, slots_log2, group_items, monkeypatch):
    """3 M genome-sampled reads of 250 bases (0.5 G occurrences): the combine-before-insert path and round 3's one-atomic-per-occurrence
    path (FAQCS_KMER_DIRECT=1, kmer_count) are two implementations of update_kmer() (trim.cpp:887-931) that share no insert code -- the
    sampling points and the whole count histogram must be identical.  First case: 40x coverage of a 12 Mbp genome, several groups, a
    partition's keys fit one LDS table.  Second case: a 200 Mbp genome (most k-mers distinct) in ONE group of 2^30 occurrences -- every
    partition is counted in several LDS rounds, each write-out reading slots an earlier one of the same launch stored."""
    import ctypes as C

    import torch

    from faqcs_amd.engine import HipEngine, _check

    n, L = 3_000_000, 250
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "--kmer_rarefaction", "--split_size", "200000", "--subset", "50"])
    dev = torch.device("cuda:0")
    seq = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    qual = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    res = torch.empty((n, 4), dtype=torch.int16, device=dev)
    results = []
    if group_items:
        monkeypatch.setenv("FAQCS_KMER_GROUP_ITEMS", str(group_items))
    for direct in ("0", "1"):
        monkeypatch.setenv("FAQCS_KMER_DIRECT", direct)
        eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << slots_log2)
        _check(eng.lib, eng.lib.faqcs_synth_fill_genome(0, seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, L, 20260101 + SEED, 0, genome))
        seg = np.arange(0, n + 32768, 32768, dtype=np.uint32)
        seg[-1] = n
        b = capi.Batch(seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
        _check(eng.lib, eng.lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
        eng.sync()
        tot = eng.kmer_totals()  # (the curve so far: the open group goes INTO THE TABLE -- round 5's path, several LDS rounds per partition in the second case)
        eng.kmer_end_table()
        results.append((tot, eng.kmer_points().copy(), [a.copy() for a in eng.kmer_histogram()]))
        eng.close()
    # ... and a third time as a pass that is counted in one piece when it ends (round 6), on a fresh engine
    monkeypatch.setenv("FAQCS_KMER_DIRECT", "0")
    eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << slots_log2)
    _check(eng.lib, eng.lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    eng.kmer_end_table()
    results.append((eng.kmer_totals(), eng.kmer_points().copy(), [a.copy() for a in eng.kmer_histogram()]))
    eng.close()
    (t0, p0, h0), (t1, p1, h1), (t2, p2, h2) = results
    assert t0 == t1 == t2 and t0[1] > 400_000_000 and len(p0) >= 10
    assert (p0 == p1).all() and (p0 == p2).all()
    assert (h0[0] == h1[0]).all() and (h0[1] == h1[1]).all() and (h0[0] == h2[0]).all() and (h0[1] == h2[1]).all()


def _kmer_rank(rank, world, port, args, n_reads, seg_size, out, maxlen=150, backend="gloo", kind="adv"):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for d in (os.path.dirname(here), here, os.path.join(here, "golden")):
        sys.path.insert(0, d)
    import torch

    torch.cuda.init()  # before libfaqcs_mi.so (torch ships its own HIP runtime)
    import torch.distributed as dist
    from oracle_engine import OracleEngine

    from faqcs_amd import driver, parallel
    from faqcs_amd.engine import HipEngine

    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    rng = np.random.Generator(np.random.PCG64(4242))
    reads = random_batch(rng, n_reads, maxlen, kind)
    segs = [reads[i:i + seg_size] for i in range(0, n_reads, seg_size)]
    epochs, points = parallel.rarefaction_schedule([len(s) for s in segs], opt.split_size, opt.num_subsample)
    lo, hi = parallel.shard_bounds(len(segs), rank, world)
    eng = HipEngine(opt, 256, 33, device=0, kmer_table_slots=1 << 22)
    ex = parallel.KmerExchange(eng, rank, world, opt.num_subsample)
    # every rank makes the same number of (collective) exchanges; with three or more parts the exchange is PIPELINED as bench.py drives it:
    # the items of part i are received and inserted behind the submission of part i + 1, finish() completes the last one
    n_parts = 2 if (maxlen <= 150 or os.environ.get("FAQCS_TEST_SERIAL_EXCHANGE")) else 4
    cuts = [lo + (hi - lo) * i // n_parts for i in range(n_parts + 1)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            seq, qual, offset, seg = driver.pack_segments(segs[a:b])
            eng.kmer_set_epochs(epochs[a:b])
            eng.process(seq, qual, offset, seg)
        if n_parts == 2:
            ex.exchange()
        else:
            ex.exchange_end()
            ex.exchange_begin()
    pts, hist = ex.finish(points, n_reads)
    if rank == 0:
        ora = OracleEngine(opt, 256, 33)
        s2, q2, o2, g2 = driver.pack_segments(segs)
        ora.process(s2, q2, o2, g2)
        ora.kmer_end_table()
        want = [(int(p["num_seq"]), int(p["distinct_kmer"]), int(p["total_kmer"])) for p in ora.kmer_points()]
        hc, hk = ora.kmer_histogram()
        want_hist = {int(a): int(b) for a, b in zip(hc, hk)}
        with open(out, "w") as f:
            diff = sorted((k, hist.get(k, 0), want_hist.get(k, 0)) for k in set(hist) | set(want_hist) if hist.get(k, 0) != want_hist.get(k, 0))
            f.write("ok" if (pts == want and hist == want_hist and len(want) > 0) else "mismatch:\n%r\n%r\nhistogram (count, keys here, keys in the oracle): %r" % (pts, want, diff[:40]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("args,n_reads,maxlen", [
    (["--kmer_rarefaction", "--split_size", "300", "--subset", "4"], 3000, 150),   # curve completes mid-run
    (["--kmer_rarefaction", "--split_size", "700"], 2600, 150),                       # curve still open at the end
    (["--kmer_rarefaction", "--split_size", "5000", "--qc_only"], 1500, 150),         # no scheduled point: the fallback one
    (["--kmer_rarefaction", "--split_size", "400", "--subset", "200"], 2400, 250),    # BASELINE configs[4]'s shape: 2x250, --subset 200
], ids=["complete", "open", "fallback", "len250_subset200"])
def test_two_rank_kmer_exchange(args, n_reads, maxlen, tmp_path, world=2):
    """owner-partitioned k-mer tables, (key, epoch) all-to-all (gloo here, two ranks sharing the
    one GPU of the box), additive epoch histograms -> the same rarefaction points and count histogram as one process."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_kmer_rank, args=(world, port, args, n_reads, 333, out, maxlen), nprocs=world, join=True)
    assert open(out).read() == "ok", open(out).read()


def test_three_rank_kmer_exchange(tmp_path):
    """Three owners: a rank count that does not divide the key space (or the 9 segments of the input) evenly."""
    test_two_rank_kmer_exchange(["--kmer_rarefaction", "--split_size", "400", "--subset", "200"], 2900, 250, tmp_path, world=3)


def _counter_rank(rank, world, port, args, n_reads, out, backend="gloo", native=False):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for d in (os.path.dirname(here), here, os.path.join(here, "golden")):
        sys.path.insert(0, d)
    import torch

    torch.cuda.init()  # before libfaqcs_mi.so: the library then binds to torch's HIP runtime (one runtime per process)
    import torch.distributed as dist
    from oracle_engine import OracleEngine

    from faqcs_amd import driver, parallel
    from faqcs_amd.engine import HipEngine

    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    rng = np.random.Generator(np.random.PCG64(777))
    reads = random_batch(rng, n_reads, 150, "adv")
    # shards are cut on reference trim() call boundaries (segments): the adapter groups of 8 restart there (trim.cpp:977-1071)
    seg_size = 517
    segs = [reads[i:i + seg_size] for i in range(0, n_reads, seg_size)]
    lo, hi = parallel.shard_bounds(len(segs), rank, world)
    eng = HipEngine(opt, 256, 33, device=0)
    seq, qual, offset, seg = driver.pack_segments(segs[lo:hi])
    if native:  # the library's own RCCL communicator: the block is all-reduced in place on the engine's compute stream
        parallel.native_comm_init(eng)
    res = eng.process(seq, qual, offset, seg)
    parallel.allreduce_counters_device(eng)  # export -> all-reduce -> import (or in place): the block on the device is the job's total
    total = eng.counters()
    n_before = sum(len(x) for x in segs[:lo])
    ora = OracleEngine(opt, 256, 33)
    s2, q2, o2, g2 = driver.pack_segments(segs)
    res_all = ora.process(s2, q2, o2, g2)
    ok = bool((total == ora.counters()).all()) and bool((res == res_all[n_before:n_before + len(res)]).all())
    flags = [None] * world
    dist.all_gather_object(flags, ok)
    if rank == 0:
        with open(out, "w") as f:
            f.write("ok" if all(flags) else "mismatch %r" % (flags,))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("args", [[], ["--adapter", "--polyA"], ["-q", "20", "--min_L", "30", "--5end", "3"]], ids=["plain", "adapter", "windowed"])
def test_two_rank_hip_counters_allreduce(args, tmp_path):
    """BASELINE configs[3] in miniature: two ranks (sharing the box's one GPU, gloo) each run the HIP engine on their shard of
    reference trim() calls; after parallel.allreduce_counters_device() EVERY rank's device block equals the single-process
    oracle's, and the per-read results equal the oracle's rows of that shard (merge semantics: trim.cpp:120-154)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_counter_rank, args=(2, port, args, 4000, out), nprocs=2, join=True)
    assert open(out).read() == "ok", open(out).read()


def test_eight_rank_hip_counters_allreduce(tmp_path):
    """BASELINE configs[3]'s rank count on the one GPU of the box: EIGHT ranks (gloo), each with its own HIP context on device 0 and its
    shard of the reference's trim() calls -- a shard count that the input's segments do not divide evenly --, --adapter --polyA so that the
    adapter groups of 8 meet the shard cuts; after the all-reduce every rank's block equals the single-process oracle's (trim.cpp:120-154)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_counter_rank, args=(8, port, ["--adapter", "--polyA"], 9000, out), nprocs=8, join=True)
    assert open(out).read() == "ok", open(out).read()


def test_eight_rank_kmer_exchange(tmp_path):
    """BASELINE configs[4]'s rank count and shape (2x250, --subset 200) on one GPU: eight owners, super-k-mer items all-to-all over gloo."""
    test_two_rank_kmer_exchange(["--kmer_rarefaction", "--split_size", "400", "--subset", "200"], 5200, 250, tmp_path, world=8)


def test_eight_rank_owner_tables_use_all_their_slots(tmp_path, monkeypatch):
    """an owner indexed its table, its level-1 buckets and its level-2 regions by the GLOBAL partition, so each of 8
    ranks used one eighth of each and a weak-scaled job hit FAQCS_E_KMER_FULL at about the single-GPU key count.  The owner now maps the
    partitions it owns onto its whole local partition space (KmerGroupDev::part_mul).  Here 8 owners with 2^22-slot tables take about
    0.4 M distinct keys each -- three quarters of the 0.52 M slots an eighth of a table has, far past what its slices' probe windows and
    the overflow area took --, THROUGH THEIR TABLES (FAQCS_KMER_FINAL=0: round 5's path; a pass counted in one piece would not touch the
    table at all); points and histogram equal the oracle's."""
    import socket

    import torch.multiprocessing as mp

    monkeypatch.setenv("FAQCS_KMER_FINAL", "0")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "result.txt")
    mp.spawn(_kmer_rank, args=(8, port, ["--kmer_rarefaction", "--split_size", "2500", "--subset", "5"], 15000, 577, out, 250, "gloo", "clean"), nprocs=8, join=True)
    assert open(out).read() == "ok", open(out).read()[:2000]


def test_bench_kmer_memory_budget_dry_run():
    """`bench.py --config kmer --dry-run-memory` prints a rank's HBM budget from faqcs_kmer_memory_plan (reads, table, group
    buffers) before anything is allocated.  One rank at BASELINE configs[4]'s per-GPU share (25 M pairs, 2^31 slots) fits the device and
    its pass is counted in one piece; eight such ranks SHARING the one GPU of this box do not fit, and the launcher says so and fails fast
    instead of running out of memory half way."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "kmer", "--dry-run-memory"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
    plan = json.loads(r.stdout.decode().strip().splitlines()[-1])["dry_run_memory"]
    assert plan["fits"] and plan["pass_counted_in_one_piece"] and plan["kmer_table_GB"] > 30 and plan["sum_GB"] < plan["device_free_GB"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "kmer", "--gpus", "8", "--dry-run-memory"], env=dict(env, FAQCS_BENCH_SHARE_GPU="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    errs = r.stderr.decode(errors="replace")
    for k in range(8):
        f = os.path.join(root, "gpurun_out", "rank%d.err" % k)
        if os.path.exists(f):
            errs += open(f, errors="replace").read()
    assert "do not fit" in errs, errs[-2000:]


@pytest.mark.parametrize("config", ["plain", "adapter", "kmer"])
def test_bench_gpus_8_on_one_gpu(config):
    """`bench.py --gpus 8` on every configuration, eight rank processes sharing the box's GPU (FAQCS_BENCH_SHARE_GPU=1, gloo): the launcher
    starts eight ranks, the line says so, and the all-reduced counter block accounts for the reads of all eight (bench.py ends the run
    otherwise); --config kmer: the owner-partitioned exchange among eight owners, its wire cost per occurrence in the line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FAQCS_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    pairs = {"plain": "4e5", "adapter": "2e5", "kmer": "5e4"}[config]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--config", config, "--pairs", pairs, "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-other-configs", "--e2e-pairs", "0", "--kmer-table-log2", "27"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and sorted(x[0] for x in line["ranks_seen"]) == list(range(8))
    assert line["reduced_block"]["reads_counted"] == line["reduced_block"]["reads_expected"] == 8 * 2 * int(float(pairs))
    if config == "kmer":
        assert line["kmer"]["points"] > 0 and 0 < line["kmer"]["wire_bytes_per_occurrence"] <= 4.0


def test_rccl_code_paths_on_one_rank(tmp_path):
    """The `nccl` (= RCCL) branches of parallel.py -- the counter all-reduce on a torch-owned CUDA tensor, the device all-to-all of
    the k-mer items, the epoch-histogram all-reduce -- need one GPU per rank, and a test box has one GPU: they run here with a
    process group of ONE rank, which still goes through RCCL's communicator setup, its registration of the staging tensors and its
    collective kernels on the real device (the two-rank tests above use gloo with both ranks on this GPU)."""
    import socket

    import torch.multiprocessing as mp

    for fn, args in ((_counter_rank, (["--adapter", "--polyA"], 2000)), (_kmer_rank, (["--kmer_rarefaction", "--split_size", "300", "--subset", "4"], 2000, 333))):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = str(tmp_path / ("result_%s.txt" % fn.__name__))
        if fn is _counter_rank:
            mp.spawn(fn, args=(1, port, args[0], args[1], out, "nccl"), nprocs=1, join=True)
            assert open(out).read() == "ok", open(out).read()
            os.remove(out)
            # ... and through the library's own communicator (faqcs_comm_init / faqcs_comm_allreduce_counters: librccl loaded by the library,
            # the block reduced in place behind the kernels and the composition fold, no staging tensor)
            mp.spawn(fn, args=(1, port, args[0], args[1], out, "nccl", True), nprocs=1, join=True)
        else:
            mp.spawn(fn, args=(1, port, args[0], args[1], args[2], out, 150, "nccl"), nprocs=1, join=True)
        assert open(out).read() == "ok", open(out).read()


def test_bench_gpus_2_runs_two_ranks(tmp_path):
    """`bench.py --gpus 2` with no launcher must START two ranks (here sharing the one GPU over gloo, FAQCS_BENCH_SHARE_GPU=1)
    and say so in its line; a disagreement between --gpus and WORLD_SIZE is an error, not a silent 1-GPU run."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FAQCS_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--pairs", "2e6", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and sorted(x[0] for x in line["ranks_seen"]) == [0, 1]
    env2 = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--pairs", "1e6"], env=env2,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r2.returncode != 0 and b"disagrees" in r2.stderr


@pytest.mark.parametrize("native", [False, True], ids=["torch_allreduce", "native_rccl"])
def test_bench_under_the_drivers_launcher_with_one_rank(native):
    """The command line the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`)
    with N = 1: bench.py then forms the `nccl` process group, all-reduces the counter block through RCCL every step and takes the MAX of
    the step times, exactly as on an 8-GPU node.  native_rccl: the default -- the first step runs the torch-staged AND the library's own
    in-place all-reduce on the same block, compares them bit for bit and uses the native one from then on."""
    import json
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FAQCS_BENCH_SHARE_GPU", "FAQCS_BENCH_BACKEND", "FAQCS_BENCH_NATIVE_RCCL"):
        env.pop(k, None)
    if not native:  # (round 5: the native collective is the default under nccl once the first step has validated it; 0 keeps the torch-staged one)
        env["FAQCS_BENCH_NATIVE_RCCL"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--pairs", "4e6", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--e2e-pairs", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = json.loads([l for l in r.stdout.decode().strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["kernel"] == "trim_lds"
    assert ("IN PLACE" in line["collective"]["what"]) == native and line["collective"]["calls"] >= 3
    assert ("bit-identical blocks on all 1 ranks" in line["collective"]["what"]) == native  # both forms ran on the first step's block and agreed
    assert line["reduced_block"]["reads_counted"] == line["reduced_block"]["reads_expected"]


def test_native_collective_in_one_process():
    """faqcs_comm_init_all / faqcs_comm_allreduce_counters_all (what `faqcs_mi --gpus N` sums its devices' blocks with): a communicator over
    ONE context leaves its block as it is -- through librccl's ncclCommInitAll and a grouped ncclAllReduce on the real device; two
    contexts on one device are refused (the command line then adds the blocks on the host), and nothing is touched by the refusal."""
    import ctypes as C

    from faqcs_amd.engine import HipEngine

    opt = parse_args(["-u", "x", "-d", "y", "--adapter"])
    rng = np.random.Generator(np.random.PCG64([17, SEED]))
    reads = random_batch(rng, 1500, 150, "adv")
    hip, ora = compare_engines(opt, reads, seg_size=400)
    want = ora.counters()
    one = (C.c_void_p * 1)(hip.ctx)
    assert hip.lib.faqcs_comm_init_all(one, 1) == 0, hip.lib.faqcs_last_error()
    assert hip.lib.faqcs_comm_allreduce_counters_all(one, 1) == 0, hip.lib.faqcs_last_error()
    assert (hip.counters() == want).all()
    assert hip.lib.faqcs_comm_init_all(one, 1) != 0  # (it has a communicator already)
    other = HipEngine(opt, 256, 33, device=0)
    two = (C.c_void_p * 2)(other.ctx, C.c_void_p(0))  # (filled below: two DIFFERENT contexts, both on device 0)
    third = HipEngine(opt, 256, 33, device=0)
    two = (C.c_void_p * 2)(other.ctx, third.ctx)
    assert hip.lib.faqcs_comm_init_all(two, 2) != 0 and b"one device" in (hip.lib.faqcs_last_error() or b"")
    assert (hip.counters() == want).all()


def test_full_size_properties():
    """BASELINE configs[1] shape at a size the oracle cannot follow: size-independent invariants of the
    counter block, device-resident submission == host submission, and idempotence of trimming."""
    import ctypes as C

    import torch

    from faqcs_amd.engine import HipEngine, _check

    n, L = 4_000_000, 150
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"])
    eng = HipEngine(opt, 256, 33, device=0)
    lib = eng.lib
    dev = torch.device("cuda:0")
    seq = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
    qual = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
    off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    res = torch.empty((n, 4), dtype=torch.int16, device=dev)
    _check(lib, lib.faqcs_synth_fill(0, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, L, 20260101, 0, 0.0))
    seg = np.array([0, n], dtype=np.uint32)
    b = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), n, 1, seg.ctypes.data, L)
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    blk = eng.counters()
    lay = capi.python_layout(256, 0)
    V = lambda name: blk[lay[name][0]:lay[name][0] + lay[name][1]]  # noqa: E731
    fs = V("filter_stats")
    assert fs[capi.TOTAL_NUMBER] == n and fs[capi.TOTAL_LENGTH] == n * L
    assert V("pre_len_hist")[L] == n and V("pre_len_hist").sum() == n
    pq = V("pre_qual").reshape(256, 42)
    assert (pq[:L].sum(axis=1) == n).all() and pq[L:].sum() == 0
    assert V("pre_read_qhist").sum() == n and V("pre_base_qhist").sum() == n * L
    r = res.cpu().numpy().view(np.uint16)
    valid = (r[:, 2] & 1) != 0
    assert fs[capi.TOTAL_TRIMMED_NUMBER] == valid.sum()
    assert fs[capi.TOTAL_TRIMMED_LENGTH] == r[valid, 1].astype(np.int64).sum()
    assert V("post_len_hist").sum() == valid.sum()
    assert V("post_qual").sum() == fs[capi.TOTAL_TRIMMED_LENGTH]
    discarded = n - int(valid.sum())
    assert discarded == fs[capi.READ_LENGTH] + fs[capi.READ_NN] + fs[capi.READ_LOW_COMPLEXITY] + fs[capi.READ_AVG_Q]
    comp = V("pre_comp").reshape(-1, 6)
    assert (comp.sum(axis=0) == n).all()
    # host submission of a slice must equal the oracle AND the device-resident result for the same reads
    m = 20000
    hs = np.concatenate([seq[: m * L + 64].cpu().numpy()])
    hq = np.concatenate([qual[: m * L + 64].cpu().numpy()])
    ho = off[: m + 1].cpu().numpy().view(np.uint32)
    from oracle_engine import OracleEngine

    ora = OracleEngine(opt, 256, 33)
    r2 = ora.process(hs, hq, ho, np.array([0, m], dtype=np.uint32))
    assert (r[:m].view(capi.RESULT_DTYPE).ravel() == r2).all()
    # linearity: the counter block is additive over any split of the reads (what the multi-GPU all-reduce
    # relies on) and per-read results do not depend on how a batch is cut
    eng2 = HipEngine(opt, 256, 33, device=0)
    res2 = torch.empty((n, 4), dtype=torch.int16, device=dev)
    cut = 1_234_567
    off_b = off[cut:]
    for lo_, hi_, o_ptr, r_ptr in ((0, cut, off.data_ptr(), res2.data_ptr()),
                                   (cut, n, off_b.data_ptr(), res2[cut:].data_ptr())):
        sg = np.array([0, hi_ - lo_], dtype=np.uint32)
        bb = capi.Batch(seq.data_ptr(), qual.data_ptr(), o_ptr, hi_ - lo_, 1, sg.ctypes.data, L)
        _check(lib, lib.faqcs_submit_device(eng2.ctx, C.byref(bb), r_ptr))
    assert (eng2.counters() == blk).all()
    assert bool((res2 == res).all())


def _fill_arenas(lib, dev, total_bytes, L):
    """Two synthetic arenas of total_bytes + padding (may exceed 4 GiB: filled in pieces of whole L-base reads)."""
    import torch

    from faqcs_amd.engine import _check

    n_fill = (total_bytes + 64 + L - 1) // L
    seq = torch.empty(n_fill * L + 128, dtype=torch.uint8, device=dev)
    qual = torch.empty(n_fill * L + 128, dtype=torch.uint8, device=dev)
    piece = (1 << 31) // L
    done = 0
    while done < n_fill:
        m = min(piece, n_fill - done)
        scratch = torch.empty(m + 1, dtype=torch.int32, device=dev)
        _check(lib, lib.faqcs_synth_fill(0, seq.data_ptr() + 64 + done * L, qual.data_ptr() + 64 + done * L, scratch.data_ptr(), m, L, 20260101, done, 0.0))
        done += m
    torch.cuda.synchronize()
    return seq, qual  # (the arenas proper start 64 bytes in: FAQCS_ARENA_PAD_BEFORE)


def _check_slices(opt, R, seq, qual, off_host, res_dev, slices, L):
    """Per-read results of the given read ranges against the oracle (arenas copied back slice by slice)."""
    from oracle_engine import OracleEngine

    for lo, hi in slices:
        b0, b1 = int(off_host[lo]), int(off_host[hi])
        hs = seq[64 + b0: 64 + b1 + 64].cpu().numpy()
        hq = qual[64 + b0: 64 + b1 + 64].cpu().numpy()
        ho = (off_host[lo:hi + 1].astype(np.int64) - b0).astype(np.uint32)
        ora = OracleEngine(opt, R, 33)
        want = ora.process(hs, hq, ho, np.array([0, hi - lo], dtype=np.uint32))
        got = res_dev[lo:hi].cpu().numpy().view(np.uint16).view(capi.RESULT_DTYPE).ravel()
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, "reads %d..%d: first differing read %d: hip=%s oracle=%s" % (lo, hi, lo + bad[0], got[bad[0]], want[bad[0]])


@pytest.mark.parametrize("L,kernel,ragged", [(150, "trim_lds", False), (150, "trim_lds", True), (128, "trim_lds", False), (75, "trim_lds", False),
                                             (250, "trim_lds", False), (250, "trim_lds", True), (300, "trim_lds", False), (320, "trim_filter_accumulate", False)])
def test_launch_at_the_4gib_arena_limit_matches_oracle(L, kernel, ragged):
    """The launch size bench.py uses: an arena of (2^32 - 4096) // L reads (u32 offsets up to 4 GiB).  The kernels do 32-bit
    arithmetic on offsets, so the END of such an arena -- the last chunk is partial -- and the reads either side of 2^31 are
    compared with the oracle; the ragged variant (read lengths L - 40 .. L) ends 33 bytes short of 2^32."""
    import ctypes as C

    import torch

    from faqcs_amd.engine import HipEngine, _check

    R = 256 if L <= 256 else 320
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"])
    eng = HipEngine(opt, R, 33, device=0)
    lib = eng.lib
    dev = torch.device("cuda:0")
    if ragged:
        rng = np.random.Generator(np.random.PCG64([L, SEED, 4]))
        lens = rng.integers(L - 40, L + 1, size=(1 << 32) // (L - 40) + 8, dtype=np.int64)
        cum = np.cumsum(lens)
        limit = (1 << 32) - 33
        n = int(np.searchsorted(cum, limit - (L - 40), side="right"))  # reads that end at least one short read before the limit
        last = limit - int(cum[n - 1])                                 # one more read that ends exactly at the limit
        assert L - 40 <= last <= 2 * L
        if last > L:                                                   # (split it so that no read is longer than L)
            ends = np.concatenate([cum[:n], [cum[n - 1] + last - (L - 20), limit]])
        else:
            ends = np.concatenate([cum[:n], [limit]])
        off_host = np.concatenate([[0], ends]).astype(np.uint32)
        total = limit
    else:
        n_reads = (0xFFFFFFFF - 4096) // L
        off_host = (np.arange(n_reads + 1, dtype=np.uint64) * L).astype(np.uint32)
        total = int(off_host[-1])
    n = len(off_host) - 1
    assert int(np.diff(off_host.astype(np.int64)).max()) <= L and (1 << 32) - total < (64 if ragged else 4096 + 2 * L)
    seq, qual = _fill_arenas(lib, dev, total, L)
    off = torch.from_numpy(off_host.view(np.int32)).to(dev)
    res = torch.empty((n, 4), dtype=torch.int16, device=dev)
    seg = np.array([0, n], dtype=np.uint32)
    b = capi.Batch(seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, 1, seg.ctypes.data, L)
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    eng.sync()
    kt = capi.KernelTimes()
    lib.faqcs_kernel_report(eng.ctx, C.byref(kt))
    assert (kt.trim_kernel or b"").decode() == kernel
    blk = eng.counters()
    lay = capi.python_layout(R, 0)
    fs = blk[lay["filter_stats"][0]:lay["filter_stats"][0] + 32]
    assert int(fs[capi.TOTAL_NUMBER]) == n and int(fs[capi.TOTAL_LENGTH]) == total
    mid = int(np.searchsorted(off_host, 1 << 31))
    _check_slices(opt, R, seq, qual, off_host, res, [(n - 20000, n), (mid - 5000, mid + 5000), (0, 5000)], L)
    # the sum of the per-read results is what the counter block says
    r = res.cpu().numpy().view(np.uint16)
    valid = (r[:, 2] & 1) != 0
    assert int(fs[capi.TOTAL_TRIMMED_NUMBER]) == int(valid.sum())
    assert int(fs[capi.TOTAL_TRIMMED_LENGTH]) == int(r[valid, 1].astype(np.int64).sum())


@pytest.mark.parametrize("args", [[], ["--5end", "3", "--3end", "5"], ["-n", "0"], ["-n", "3"], ["--avg_q", "25"], ["--adapter", "--polyA"],
                                  ["--5trim_off"], ["--mode", "HARD", "-q", "10"]], ids=lambda a: " ".join(a) or "default")
@pytest.mark.parametrize("maxlen", [150, 250, 700])
def test_terminal_n_flags_in_the_batch_equal_the_kernels_own_lookup(args, maxlen):
    """faqcs_batch.terminal_n (ABI 2): per-read flags -- bit 0 the read's first base is N, bit 1 its last -- computed by
    faqcs_terminal_n_flags() or the caller's parser.  Results with the flags must equal results without them (the kernels then
    read the two bases themselves) and the oracle's, on reads whose ends are N / n / empty / one base long."""
    import ctypes as C

    import torch

    from faqcs_amd import driver
    from faqcs_amd.engine import HipEngine, _check
    from oracle_engine import OracleEngine

    rng = np.random.Generator(np.random.PCG64([11, maxlen, len(args), SEED]))
    reads = random_batch(rng, 4000, maxlen, "adv")
    for k in range(0, len(reads), 7):  # force terminal N / n on a share of the reads
        d, s, q = reads[k]
        if len(s):
            s = bytearray(s)
            if k % 3 != 1:
                s[0] = ord("N") if k % 2 else ord("n")
            if k % 3 != 2:
                s[-1] = ord("N")
            reads[k] = (d, bytes(s), q)
    reads[5] = (b"@x", b"", b"")
    reads[6] = (b"@x", b"N", b"I")
    reads[8] = (b"@x", b"NN", b"II")
    R = 256 if maxlen <= 256 else 1024
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"] + args)
    seq, qual, offset, seg = driver.pack_segments([reads[i:i + 517] for i in range(0, len(reads), 517)])
    n = len(reads)
    want = OracleEngine(opt, R, 33)
    want_res = want.process(seq, qual, offset, seg)
    dev = torch.device("cuda:0")
    pad = np.zeros(64, np.uint8)
    d_seq = torch.from_numpy(np.concatenate([pad, seq, pad])).to(dev)
    d_qual = torch.from_numpy(np.concatenate([pad, qual, pad])).to(dev)
    d_off = torch.from_numpy(offset.view(np.int32)).to(dev)
    tn = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    outs = []
    for use_flags in (False, True):
        eng = HipEngine(opt, R, 33, device=0)
        lib = eng.lib
        if use_flags:
            _check(lib, lib.faqcs_terminal_n_flags(0, d_seq.data_ptr() + 64, d_off.data_ptr(), n, tn.data_ptr()))
            torch.cuda.synchronize()
            f = tn[:n].cpu().numpy()
            lens = np.diff(offset.astype(np.int64))
            first = np.array([lens[i] > 0 and seq[offset[i]] == ord("N") for i in range(n)])
            last = np.array([lens[i] > 0 and seq[offset[i + 1] - 1] == ord("N") for i in range(n)])
            assert ((f & 1) == first).all() and (((f >> 1) & 1) == last).all() and (f < 4).all()
        res = torch.empty((n, 4), dtype=torch.int16, device=dev)
        b = capi.Batch(d_seq.data_ptr() + 64, d_qual.data_ptr() + 64, d_off.data_ptr(), n, len(seg) - 1, seg.ctypes.data,
                       int(np.diff(offset.astype(np.int64)).max()), tn.data_ptr() if use_flags else None)
        _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
        eng.sync()
        got = res.cpu().numpy().view(np.uint16).view(capi.RESULT_DTYPE).ravel()
        bad = np.nonzero(got != want_res)[0]
        assert len(bad) == 0, "flags=%s: first differing read %d: hip=%s oracle=%s" % (use_flags, bad[0], got[bad[0]], want_res[bad[0]])
        assert (eng.counters() == want.counters()).all(), "flags=%s" % use_flags
        outs.append(got)
    assert (outs[0] == outs[1]).all()
    # the host-buffer entry points take host flags (the native CLI's parser and the Python driver fill them while packing)
    eng = HipEngine(opt, R, 33, device=0)
    f = driver.terminal_n_flags(seq, offset)
    assert (f == tn[:n].cpu().numpy()).all()
    got = eng.process(seq, qual, offset, seg, f)
    assert (got == want_res).all() and (eng.counters() == want.counters()).all()


def test_full_size_adapter_polya():
    """BASELINE configs[2]'s option set (--adapter --polyA, 5 % read-through) on 4 M reads: a size at which the adapter pre-pass runs
    thousands of 32 768-read segments and the trim kernel's flush / register-spill paths with an adapter window trigger.  Invariants
    of the adapter statistics against the per-read results, and the oracle on the first two and the last (partial) segment."""
    import ctypes as C

    import torch

    from faqcs_amd.engine import HipEngine, _check

    n, L, R = 4_000_000 + 12_345, 150, 256
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33", "--adapter", "--polyA"])
    eng = HipEngine(opt, R, 33, device=0)
    lib = eng.lib
    dev = torch.device("cuda:0")
    seq = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    qual = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    res = torch.empty((n, 4), dtype=torch.int16, device=dev)
    _check(lib, lib.faqcs_synth_fill(0, seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, L, 20260101, 0, 0.05))
    seg = np.arange(0, n + capi.SEGMENT_READS, capi.SEGMENT_READS, dtype=np.uint32)
    seg[-1] = n
    b = capi.Batch(seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, len(seg) - 1, seg.ctypes.data, L)
    _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(b), res.data_ptr()))
    blk = eng.counters()
    na = eng.holder.n_adapters
    lay = capi.python_layout(R, na)
    fs = blk[lay["filter_stats"][0]:lay["filter_stats"][0] + 32]
    assert int(fs[capi.TOTAL_NUMBER]) == n and int(fs[capi.TOTAL_LENGTH]) == n * L
    r = res.cpu().numpy().view(np.uint16)
    ast = blk[lay["adapter_stats"][0]:lay["adapter_stats"][0] + 2 * na].reshape(na, 2)
    credited = np.bincount(r[:, 3], minlength=na + 1)
    assert (credited[1:] == ast[:, 0]).all(), "reads credited per adapter"
    assert 0.03 * n < credited[1:].sum() < 0.12 * n  # (5 % read-through plus chance hits)
    valid = (r[:, 2] & 1) != 0
    assert int(fs[capi.TOTAL_TRIMMED_NUMBER]) == int(valid.sum())
    assert int(fs[capi.TOTAL_TRIMMED_LENGTH]) == int(r[valid, 1].astype(np.int64).sum())
    post_len = blk[lay["post_len_hist"][0]:lay["post_len_hist"][0] + lay["post_len_hist"][1]]
    assert (np.bincount(r[valid, 1], minlength=R + 1)[:R + 1] == post_len).all()
    off_host = off.cpu().numpy().view(np.uint32)
    from oracle_engine import OracleEngine

    for lo, hi in ((0, 2 * capi.SEGMENT_READS), (int(seg[-2]), n)):
        b0, b1 = int(off_host[lo]), int(off_host[hi])
        hs, hq = seq[64 + b0: 64 + b1 + 64].cpu().numpy(), qual[64 + b0: 64 + b1 + 64].cpu().numpy()
        ho = (off_host[lo:hi + 1].astype(np.int64) - b0).astype(np.uint32)
        sg = np.arange(0, hi - lo + capi.SEGMENT_READS, capi.SEGMENT_READS, dtype=np.uint32)
        sg[-1] = hi - lo
        want = OracleEngine(opt, R, 33).process(hs, hq, ho, sg)
        got = r[lo:hi].view(capi.RESULT_DTYPE).ravel()
        assert (got == want).all(), "reads %d..%d" % (lo, hi)


def test_full_size_kmer_250():
    """BASELINE configs[4]'s shape (2x250, --kmer_rarefaction, genome-sampled reads): the first 50 000 reads against the oracle
    (sampling points, distinct / total k-mers, the count histogram), then 1 M reads for the invariants of the curve."""
    import ctypes as C

    import torch

    from faqcs_amd.engine import HipEngine, _check
    from oracle_engine import OracleEngine

    n, m, L, R = 1_000_000, 50_000, 250, 256
    args = ["-u", "x", "-d", "y", "--ascii", "33", "--kmer_rarefaction", "--split_size", "10000", "--subset", "40"]
    opt = parse_args(args)
    dev = torch.device("cuda:0")
    eng = HipEngine(opt, R, 33, device=0, kmer_table_slots=1 << 28)
    lib = eng.lib
    seq = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    qual = torch.empty(n * L + 128, dtype=torch.uint8, device=dev)
    off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    res = torch.empty((n, 4), dtype=torch.int16, device=dev)
    _check(lib, lib.faqcs_synth_fill_genome(0, seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), n, L, 20260101, 0, 50_000_000))

    def run(engine, count):
        seg = np.arange(0, count + 4000, 4000, dtype=np.uint32)  # (segments = the reference's trim() calls: sampling points fall on their ends)
        seg[-1] = count
        b = capi.Batch(seq.data_ptr() + 64, qual.data_ptr() + 64, off.data_ptr(), count, len(seg) - 1, seg.ctypes.data, L)
        _check(engine.lib, engine.lib.faqcs_submit_device(engine.ctx, C.byref(b), res.data_ptr()))
        engine.sync()
        return seg

    seg = run(eng, m)
    hs, hq = seq[64: 64 + m * L + 64].cpu().numpy(), qual[64: 64 + m * L + 64].cpu().numpy()
    ho = off[: m + 1].cpu().numpy().view(np.uint32)
    ora = OracleEngine(opt, R, 33)
    want = ora.process(hs, hq, ho, seg)
    assert (res[:m].cpu().numpy().view(np.uint16).view(capi.RESULT_DTYPE).ravel() == want).all()
    assert (eng.counters() == ora.counters()).all()
    eng.kmer_end_table()
    ora.kmer_end_table()
    p1, p2 = eng.kmer_points(), ora.kmer_points()
    assert len(p1) >= 4 and (p1 == p2).all()
    h1, h2 = eng.kmer_histogram(), ora.kmer_histogram()
    assert (h1[0] == h2[0]).all() and (h1[1] == h2[1]).all()
    # the whole set on a fresh engine: the curve is monotone, ends at the table's totals, and its first points are the prefix's
    eng2 = HipEngine(opt, R, 33, device=0, kmer_table_slots=1 << 28)
    run(eng2, n)
    eng2.kmer_end_table()
    pts = eng2.kmer_points()
    assert len(pts) > len(p1)
    k = min(len(p1), len(pts)) - 1
    assert (pts[:k] == p1[:k]).all()
    assert (np.diff(pts["distinct_kmer"].astype(np.int64)) >= 0).all() and (np.diff(pts["total_kmer"].astype(np.int64)) > 0).all()
    assert (pts["distinct_kmer"] <= pts["total_kmer"]).all()
    cnt, nk = eng2.kmer_histogram()
    assert int((cnt * nk).sum()) >= int(pts["total_kmer"][-1])


_SHIM_BIN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))),
                                       "oracle", "_ref", "FaQCs_hip")


@pytest.mark.parametrize("name", __import__("golden_util").case_names())
def test_reference_driver_with_hip_trim(name, fixture_cache, tmp_path):
    """the reference's OWN driver (FaQCs.cpp, options.cpp, fastq.cpp, plot.cpp compiled from where
    they lie) linked against integration/trim_shim.cpp -> libfaqcs_mi.so must reproduce the reference's outputs."""
    import os

    from golden_util import load_case, run_case_binary

    if not os.path.exists(_SHIM_BIN):
        pytest.skip("oracle/_ref/FaQCs_hip not built (needs /root/reference at build time: make -C oracle ref_hip)")
    bad = run_case_binary(load_case(name), fixture_cache, tmp_path, _SHIM_BIN)
    assert not bad, "\n".join(bad)


_CLI_BIN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))),
                                      "faqcs_amd", "faqcs_mi")


@pytest.mark.parametrize("name", __import__("golden_util").case_names())
def test_native_cli_reproduces_reference(name, fixture_cache, tmp_path):
    """faqcs_amd/faqcs_mi (C++ driver: threaded FASTQ readers -> pinned SoA buffers -> pipelined C ABI -> ordered
    writer) against the reference's own outputs, including the k-mer rarefaction cases."""
    from golden_util import load_case, run_case_binary

    bad = run_case_binary(load_case(name), fixture_cache, tmp_path, _CLI_BIN)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["advbig_default", "advbig_adapter_polyA", "adv_default", "adv_discard", "adv_unpaired_only",
                                  "adv_kmer", "adv_kmer_qc_only_subset1", "advbig_kmer", "head250_kmer", "long300_kmer_q20", "long8k_kmer_replaceN"])
def test_native_cli_on_two_devices(name, fixture_cache, tmp_path):
    """faqcs_mi --gpu_ids 0,0: the 32 768-read buffers are dealt round robin to TWO device contexts (both on this box's one GPU),
    the ordered writers re-serialise by buffer number and the two counter blocks are summed -- one QC.stats.txt, the same
    bytes as the reference (the advbig cases cross the buffer boundary, so both contexts see work; FaQCs.cpp:240-501)."""
    from golden_util import case_names, load_case, run_case_binary

    if name not in case_names():
        pytest.skip("no such golden case")
    bad = run_case_binary(load_case(name), fixture_cache, tmp_path, _CLI_BIN, extra_args=["--gpu_ids", "0,0"])
    assert not bad, "\n".join(bad)
    if "kmer" in name:  # --kmer_rarefaction on several contexts: every context trims its buffers, every k-mer has one owner context
        (tmp_path / "three").mkdir()  # (faqcs_kmer_forward; three contexts: an owner count that does not divide the key space evenly)
        bad = run_case_binary(load_case(name), fixture_cache, tmp_path / "three", _CLI_BIN, extra_args=["--gpu_ids", "0,0,0"])
        assert not bad, "\n".join(bad)


@pytest.mark.parametrize("name", ["adv_kmer", "adv_kmer_qc_only_subset1", "advbig_kmer", "head250_kmer", "long300_kmer_q20", "long8k_kmer_replaceN"])
def test_native_cli_forwards_k_mers_by_peer_copy(name, fixture_cache, tmp_path):
    """The branch of faqcs_kmer_forward a multi-GPU node takes -- the owner sits on ANOTHER device: hipMemcpyPeerAsync into one of the
    owner's two staging buffers on its copy stream, the insert behind an event on its compute stream -- has never run for want of a
    second GPU.  FAQCS_KMER_FORCE_PEER_COPY=1 sends same-device owners through it: the six k-mer goldens on two and on
    three contexts, byte-identical tables (trim.cpp:133-135 is what the exchange replaces)."""
    from golden_util import case_names, load_case, run_case_binary

    if name not in case_names():
        pytest.skip("no such golden case")
    for ids in ("0,0", "0,0,0"):
        d = tmp_path / ids.replace(",", "_")
        d.mkdir()
        bad = run_case_binary(load_case(name), fixture_cache, d, _CLI_BIN, extra_args=["--gpu_ids", ids], env={"FAQCS_KMER_FORCE_PEER_COPY": "1"})
        assert not bad, "\n".join(bad)


def test_native_cli_reports_a_device_error_in_a_large_input(tmp_path):
    """A read whose quality exceeds Q41 in the FIRST buffer of an input of 20 buffers: the gate thread meets the error while
    the producer still has many buffers to submit.  The run must end with the reference's message and a failure status, not
    hang with every buffer in flight (fastq.h:31-33; FaQCs.cpp:136-147)."""
    import subprocess

    n = 20 * 32768 + 7
    rec = b"@r%d\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 40 + b"\n"
    path = str(tmp_path / "big.fastq")
    with open(path, "wb") as f:
        chunks = []
        for i in range(n):
            if i == 5:
                chunks.append(b"@r5\n" + b"ACGT" * 10 + b"\n+\n" + b"I" * 39 + bytes([33 + 42]) + b"\n")
            else:
                chunks.append(rec % i)
            if len(chunks) == 65536:
                f.write(b"".join(chunks))
                chunks = []
        f.write(b"".join(chunks))
    for extra in ([], ["--gpu_ids", "0,0"]):
        r = subprocess.run([_CLI_BIN, "-u", path, "-d", str(tmp_path / "out"), "--ascii", "33"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1, (r.returncode, r.stderr.decode()[-500:])
        assert b"Caught the error fastq.h:quality_score" in r.stderr
    p2 = str(tmp_path / "big2.fastq")
    os.link(path, p2)
    r = subprocess.run([_CLI_BIN, "-1", path, "-2", p2, "-d", str(tmp_path / "out2"), "--ascii", "33"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1 and b"Caught the error fastq.h:quality_score" in r.stderr


@pytest.mark.gpu
def test_native_cli_second_input_that_cannot_be_opened(tmp_path):
    """-2 names a file that is not there while the first file's readers already run (FaQCs.cpp:167-176): the reference's two lines on
    stderr and exit code 1 -- not an abort (until round 6 the exception unwound past the first file's running threads: std::terminate).
    Compared with the reference binary where it is built."""
    import gzip
    import subprocess

    p1 = str(tmp_path / "a_1.fastq.gz")
    open(p1, "wb").write(gzip.compress(b"".join(b"@r%d/1\n" % i + b"ACGTTGCAAC" * 6 + b"\n+\n" + b"I" * 60 + b"\n" for i in range(3000))))
    missing = str(tmp_path / "not_there_2.fastq.gz")
    r = subprocess.run([_CLI_BIN, "-1", p1, "-2", missing, "-d", str(tmp_path / "out")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1, (r.returncode, r.stderr.decode()[-400:])
    want = ("Unable to open %s for loading read two sequences\nCaught the error I/O error\n" % missing).encode()
    assert r.stderr.endswith(want), r.stderr.decode()[-400:]
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "FaQCs_ref")
    if os.path.exists(ref_bin):
        q = subprocess.run([ref_bin, "-1", p1, "-2", missing, "-d", str(tmp_path / "out_ref")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert q.returncode == r.returncode and q.stderr.endswith(want), (q.returncode, q.stderr.decode()[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [True, False], ids=["gz", "plain"])
@pytest.mark.parametrize("paired", [True, False], ids=["paired", "unpaired"])
def test_native_cli_output_that_cannot_be_opened(paired, packed, tmp_path):
    """An output path that is a directory (FaQCs.cpp:188-223, :560-579), with compressed input (the streaming path, whose readers run by
    then) and with plain input (the mapped path): the reference's lines and exit code 1, compared with the reference binary where it is
    built (no overwrite notice for what is not a regular file: file_util.cpp:11-20)."""
    import gzip
    import subprocess

    text = lambda m: b"".join(b"@r%d/%d\n" % (i, m) + b"ACGTTGCAAC" * 6 + b"\n+\n" + b"I" * 60 + b"\n" for i in range(3000))
    blob = (lambda m: gzip.compress(text(m))) if packed else text
    p1, p2 = str(tmp_path / ("a_1.fastq" + (".gz" if packed else ""))), str(tmp_path / ("a_2.fastq" + (".gz" if packed else "")))
    open(p1, "wb").write(blob(1)); open(p2, "wb").write(blob(2))
    ref_bin = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "FaQCs_ref")
    tails = []
    for who, exe in (("mine", _CLI_BIN), ("reference", ref_bin)):
        if not os.path.exists(exe):
            continue
        d = tmp_path / ("out_" + who)
        (d / ("QC.1.trimmed.fastq" if paired else "QC.unpaired.trimmed.fastq")).mkdir(parents=True)
        cmd = [exe, "-1", p1, "-2", p2, "-d", str(d)] if paired else [exe, "-u", p1, "-d", str(d)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 1, (who, r.returncode, r.stderr.decode()[-400:])
        tails.append(r.stderr.decode().replace("out_" + who, "out").splitlines()[-2:])
    what = "read one sequences" if paired else "unpaired read sequences"
    assert tails[0][0].startswith("Unable to open ") and tails[0][0].endswith(" for writing " + what) and tails[0][1] == "Caught the error I/O error", tails[0]
    if len(tails) == 2:
        assert tails[0] == tails[1], tails


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["streaming", "gzip", "bgzf", "unpaired"])
def test_native_cli_streaming_path_writes_what_came_before_a_device_error(mode, tmp_path):
    """The streaming path's output side (a gate, a pool of formatters, two committers that write in input order): a quality above Q41 in
    the THIRD pair of buffers ends the run with the reference's message, and the mate files hold exactly the reads of the first two
    pairs of buffers -- trim() throws before anything of its buffer is written (FaQCs.cpp:296-361), what was rendered before is written
    in order, nothing behind it is.  Through gzread of plain files, the parallel gzip reader and the BGZF reader."""
    import gzip
    import subprocess

    n = 5 * 32768 + 11
    bad = 2 * 32768 + 5

    def text(mate):
        out = []
        for i in range(n):
            q = b"I" * 59 + (bytes([33 + 42]) if (i == bad and mate == 1) else b"I")
            out.append(b"@r%d/%d\n" % (i, mate) + b"ACGTTGCAAC" * 6 + b"\n+\n" + q + b"\n")
        return out

    paths, want = [], []
    for mate in (1, 2):
        recs = text(mate)
        want.append(b"".join(recs[: 2 * 32768]))
        blob = b"".join(recs)
        if mode == "streaming":
            p = str(tmp_path / ("m%d.fastq" % mate))
            open(p, "wb").write(blob)
        elif mode in ("gzip", "unpaired"):
            p = str(tmp_path / ("m%d.fastq.gz" % mate))
            open(p, "wb").write(gzip.compress(blob, 1))
        else:
            import struct
            import zlib
            p = str(tmp_path / ("m%d.bgzf.gz" % mate))
            with open(p, "wb") as f:
                for o in list(range(0, len(blob), 65280)) + [None]:
                    raw = b"" if o is None else blob[o:o + 65280]
                    c = zlib.compressobj(1, zlib.DEFLATED, -15)
                    body = c.compress(raw) + c.flush()
                    f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))
        paths.append(p)
    d = str(tmp_path / "out")
    env = dict(os.environ, FAQCS_MI_STREAMING="1", FAQCS_MI_PARGZ_MIN="1")
    inputs = ["-u", paths[0]] if mode == "unpaired" else ["-1", paths[0], "-2", paths[1]]
    r = subprocess.run([_CLI_BIN] + inputs + ["-d", d, "--ascii", "33", "--min_L", "30", "--trim_only"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr.decode()[-500:])
    assert b"Caught the error fastq.h:quality_score" in r.stderr
    if mode == "unpaired":  # (process_unpaired's own gate / formatters / committer)
        assert open(os.path.join(d, "QC.unpaired.trimmed.fastq"), "rb").read() == want[0]
        return
    got1, got2 = open(os.path.join(d, "QC.1.trimmed.fastq"), "rb").read(), open(os.path.join(d, "QC.2.trimmed.fastq"), "rb").read()
    assert got1 == want[0] and got2 == want[1], (len(got1), len(want[0]), len(got2), len(want[1]))
    assert open(os.path.join(d, "QC.unpaired.trimmed.fastq"), "rb").read() == b""


_REF_BIN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))),
                                      "oracle", "_ref", "FaQCs_ref")


@pytest.mark.parametrize("args,n_reads", [([], 34000), (["--adapter", "--polyA", "--discard", "-t", "1"], 2600), (["--kmer_rarefaction", "--split_size", "9000", "-q", "15"], 34000)],
                         ids=["default", "adapter", "kmer"])
def test_native_cli_on_a_long_read_file_equals_the_reference_run_here(args, n_reads, tmp_path):
    """A file of ragged reads of 60 ... 5 000 bases -- more than one 32 768-read buffer of them -- through faqcs_mi (trim_long, adapter_overlap<1, 8192>,
    the k-mer kernels) and through the REAL reference binary built by `make -C oracle ref` (it travels to the GPU box under oracle/_ref), run
    here on the same file: QC.stats.txt, every --debug table and the trimmed FASTQ must be byte-identical."""
    import hashlib
    import subprocess

    if not os.path.exists(_REF_BIN):
        pytest.skip("oracle/_ref/FaQCs_ref not built (needs /root/reference at build time: make -C oracle ref)")
    import make_fixtures

    rng = np.random.Generator(np.random.PCG64([29, n_reads, SEED]))
    path = str(tmp_path / "long.fastq")
    with open(path, "wb") as f:
        for i in range(n_reads):
            u = rng.random()
            L = int(rng.integers(1100, 5001)) if u < 0.35 else (int(rng.integers(60, 1100)) if u < 0.9 else int(rng.integers(30, 60)))
            sq, q = make_fixtures._adv_read(rng, L)
            f.write(b"@L%d extra\n" % i + sq.tobytes() + b"\n+\n" + q.tobytes() + b"\n")
    outs = {}
    for name, binary in (("ref", _REF_BIN), ("mi", _CLI_BIN)):
        out = str(tmp_path / name)
        for _ in range(6):  # (the reference can die of SIGPIPE feeding the absent R: see make_golden.py)
            subprocess.run(["rm", "-rf", out])
            r = subprocess.run([binary, "-u", path, "-d", out, "--debug", "--ascii", "33"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            if r.returncode != -13:
                break
        assert r.returncode == 0, (name, r.returncode, r.stderr.decode(errors="replace")[-800:])
        outs[name] = {fn: hashlib.md5(open(os.path.join(out, fn), "rb").read()).hexdigest() for fn in sorted(os.listdir(out)) if not fn.endswith(".pdf")}
    assert outs["ref"].keys() == outs["mi"].keys(), (sorted(outs["ref"]), sorted(outs["mi"]))
    bad = [fn for fn in outs["ref"] if outs["ref"][fn] != outs["mi"][fn]]
    assert not bad, bad


@pytest.mark.parametrize("args,maxlen", [
    ([], 150), (["--adapter", "--polyA", "--discard", "-t", "1"], 150), (["--kmer_rarefaction", "--split_size", "20000"], 250),
    (["--mode", "HARD", "-q", "11", "--5trim_off"], 100), (["--mode", "BWA", "--avg_q", "22", "-n", "1"], 151),
    (["--5end", "5", "--3end", "11", "--min_L", "30", "--lc", "0.6"], 126), (["--replace_to_N_q", "14", "-q", "12", "--out_ascii", "64"], 150),
    (["--qc_only", "--adapter", "-t", "1"], 150), (["--phiX", "-t", "1"], 300)],
    ids=["default", "adapter", "kmer250", "hard100", "bwa_avgq151", "ends126", "replaceN", "qc_only_adapter", "phix300"])
def test_native_cli_equals_the_reference_run_here_on_fresh_pairs(args, maxlen, tmp_path):
    """Beyond the committed goldens: 40 000 fresh adversarial pairs (FAQCS_TEST_SEED re-seeds them) through faqcs_mi and through the REAL
    reference binary (oracle/_ref/FaQCs_ref travels to the GPU box), both run here: every output file byte-identical."""
    import hashlib
    import subprocess

    if not os.path.exists(_REF_BIN):
        pytest.skip("oracle/_ref/FaQCs_ref not built (needs /root/reference at build time: make -C oracle ref)")
    import make_fixtures

    n = 6000 if "--phiX" in args else 40000
    r1, r2 = make_fixtures.adversarial(n, seed=1000 + SEED * 31 + maxlen, maxlen=maxlen, id_prefix="F")
    p1, p2 = str(tmp_path / "f_1.fastq"), str(tmp_path / "f_2.fastq")
    make_fixtures.write_fastq(p1, r1)
    make_fixtures.write_fastq(p2, r2)
    outs = {}
    for name, binary in (("ref", _REF_BIN), ("mi", _CLI_BIN)):
        out = str(tmp_path / name)
        for _ in range(6):  # (the reference can die of SIGPIPE feeding the absent R: see make_golden.py)
            subprocess.run(["rm", "-rf", out])
            r = subprocess.run([binary, "-1", p1, "-2", p2, "-d", out, "--debug"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            if r.returncode != -13:
                break
        assert r.returncode == 0, (name, r.returncode, r.stderr.decode(errors="replace")[-800:])
        outs[name] = {fn: hashlib.md5(open(os.path.join(out, fn), "rb").read()).hexdigest() for fn in sorted(os.listdir(out)) if not fn.endswith(".pdf")}
    assert outs["ref"].keys() == outs["mi"].keys(), (sorted(outs["ref"]), sorted(outs["mi"]))
    bad = [fn for fn in outs["ref"] if outs["ref"][fn] != outs["mi"][fn]]
    assert not bad, bad


@pytest.mark.parametrize("args", [[], ["--discard", "--adapter"], ["-u"]], ids=["paired", "paired_discard_adapter", "unpaired"])
def test_native_cli_mapped_path_equals_streaming_path(args, tmp_path):
    """faqcs_mi reads uncompressed regular files through the memory-mapped path (parallel index / parse / format, pwrite at
    offsets handed out in input order) and everything else through the streaming path (FAQCS_MI_STREAMING=1 forces it):
    same bytes in every output file on an input of several 32 768-record buffers with ragged lengths, a last buffer that is
    short, and records without a final newline."""
    import subprocess

    import make_fixtures

    rng = np.random.Generator(np.random.PCG64(2024))
    n = 3 * 32768 + 1234
    r1, r2 = [], []
    for i in range(n):
        s1, q1 = make_fixtures._adv_read(rng, 150)
        s2, q2 = make_fixtures._adv_read(rng, 150)
        r1.append(b"@P%d/1 x\n%s\n+\n%s\n" % (i, s1.tobytes(), q1.tobytes()))
        r2.append(b"@P%d/2 y\n%s\n+\n%s\n" % (i, s2.tobytes(), q2.tobytes()))
    p1, p2 = str(tmp_path / "a_1.fastq"), str(tmp_path / "a_2.fastq")
    with open(p1, "wb") as f:
        f.write(b"".join(r1)[:-1])  # no final newline
    with open(p2, "wb") as f:
        f.write(b"".join(r2))
    outs = []
    for mode, env in (("mapped", {}), ("streaming", {"FAQCS_MI_STREAMING": "1"})):
        d = str(tmp_path / mode)
        if args == ["-u"]:
            cmd = [_CLI_BIN, "-u", p1, "-d", d, "--debug", "--discard"]
        else:
            cmd = [_CLI_BIN, "-1", p1, "-2", p2, "-d", d, "--debug"] + args
        r = subprocess.run(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-500:]
        outs.append({fn: open(os.path.join(d, fn), "rb").read() for fn in sorted(os.listdir(d))})
    assert sorted(outs[0]) == sorted(outs[1])
    for fn in outs[0]:
        assert outs[0][fn] == outs[1][fn], "%s differs between the mapped and the streaming path" % fn


@pytest.mark.parametrize("args", [[], ["--lc", "0.5"], ["--lc", "0.3", "--min_L", "20"], ["--adapter"]], ids=["default", "lc05", "lc03", "adapter"])
def test_at_rich_reads_match_oracle(args):
    """AT- and GC-rich reads: two bases each fill more than dthr of the kept window, so the dinucleotide part of the
    low-complexity filter (trim.cpp:405-513) has to count transitions exactly -- the two-class counter of trim_lds at the
    default --lc, the general pass at small --lc -- including reads that do trip it (alternating ATAT.. stretches)."""
    rng = np.random.Generator(np.random.PCG64([77, SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--ascii", "33"] + args)
    reads = []
    for i in range(2500):
        L = int(rng.integers(60, 151))
        kind = i % 5
        if kind == 0:
            p = [0.46, 0.46, 0.04, 0.04]      # AT-rich, random order
        elif kind == 1:
            p = [0.05, 0.05, 0.45, 0.45]      # GC-rich
        elif kind == 2:
            p = [0.5, 0.5, 0.0, 0.0]          # only A and T
        else:
            p = [0.3, 0.3, 0.2, 0.2]
        s = np.frombuffer(b"ATCG", np.uint8)[rng.choice(4, L, p=p)].copy()
        if kind == 3:                          # mostly alternating: trips
            s[: L - 10] = np.frombuffer(b"AT" * 80, np.uint8)[: L - 10]
            flips = rng.integers(0, L - 10, 6)
            s[flips] = ord("A")
        if kind == 4 and L > 40:               # a lower-case stretch and an N inside the run
            s[10:30] = np.frombuffer(bytes(s[10:30]).lower(), np.uint8)
            s[35] = ord("N")
        q = (rng.integers(25, 41, L) + 33).astype(np.uint8)
        if rng.random() < 0.4:
            q[int(rng.integers(L // 2, L)):] = 35
        reads.append((b"@x", s.tobytes(), q.tobytes()))
    compare_engines(opt, reads, seg_size=700)


@pytest.mark.parametrize("args", [[], ["--lc", "0.5"], ["--mode", "BWA"], ["--avg_q", "20", "-n", "1"]], ids=["default", "lc05", "bwa", "avgq"])
@pytest.mark.parametrize("L", [4, 8, 32, 36, 50, 51, 52, 53, 64, 75, 76, 80, 96, 100, 104, 108, 120, 128, 144, 148, 150, 152, 156, 160, 161, 164, 192, 200, 224, 240, 248, 250, 251, 252, 253, 256, 288, 300, 301, 304])
def test_equal_length_batches_with_a_kept_window_that_ends_on_the_last_base(L, args):
    """Every read of the batch has the same length (mostly a multiple of 4: the wave's dword loops stop exactly at the read's
    end), a low-quality head that the 5' walk cuts and a high-quality tail, so the kept window is [a, L) with a > 0: its base
    counts are (total) - (prefix in front of a).  Found by fuzzing: the prefix at a window end on the wave's last dword boundary
    was never captured for widths where that boundary is not the slot's last dword."""
    rng = np.random.Generator(np.random.PCG64([91, L, SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1"] + args)
    reads = []
    for i in range(900):
        s = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, L, p=[.235, .235, .235, .235, .02, .01, .01, .01, .01])].copy()
        q = (rng.integers(28, 41, L) + 33).astype(np.uint8)
        head = int(rng.integers(0, min(L, 12))) if i % 3 else 0
        q[:head] = 33 + rng.integers(0, 4, head)
        if i % 7 == 0 and L > 20:
            q[L - int(rng.integers(1, 9)):] = 34      # some reads do lose their tail
        reads.append((b"@x", s.tobytes(), q.tobytes()))
    compare_engines(opt, reads, R=256 if L <= 256 else 1024, seg_size=300)


@pytest.mark.parametrize("args", [[], ["--adapter", "--polyA"], ["--lc", "0.5", "-n", "1"], ["--mode", "HARD", "-q", "10", "--avg_q", "30"], ["--5end", "7", "--3end", "3", "--mode", "BWA"]],
                         ids=["default", "adapter", "lc05_n1", "hard_avgq", "ends_bwa"])
@pytest.mark.parametrize("L", [32, 64, 96, 128, 192, 224, 256, 288])
def test_chunks_of_equal_length_reads_in_padded_rows(L, args):
    """trim_lds stages a chunk whose reads all have the same length, a multiple of 32 bases, as padded rows (dma_rows) and any other chunk as
    one contiguous span: a batch that holds both kinds of chunk -- runs of equal-length reads broken by a shorter read here and there --
    with terminal-N runs, low-quality heads and tails, reads the filters reject after their cells were counted (the take-back passes
    stage the qualities a second time) and out-of-range quality bytes in some reads."""
    rng = np.random.Generator(np.random.PCG64([92, L, SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "20"] + args)
    reads = []
    for i in range(2600):
        l = L if (i % 211) or i < 300 else int(rng.integers(1, L))   # the first 300 reads: whole chunks of equal length for sure
        if 900 <= i < 1100:
            l = int(rng.integers(max(1, L - 40), L + 1))               # a ragged stretch
        s = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, l, p=[.235, .235, .235, .235, .02, .01, .01, .01, .01])].copy()
        q = (rng.integers(25, 41, l) + 33).astype(np.uint8)
        k = i % 13
        if k == 0 and l > 8: s[:int(rng.integers(1, 6))] = ord("N")              # terminal N runs
        if k == 1 and l > 8: s[l - int(rng.integers(1, 6)):] = ord("N")
        if k == 2 and l > 30: q[l - int(rng.integers(1, 30)):] = 35               # '#' tail
        if k == 3 and l > 12: q[:int(rng.integers(1, 12))] = 33 + rng.integers(0, 4)
        if k == 4 and l > 40: s[10:10 + int(rng.integers(2, 30))] = ord("N")      # poly-N: vetoed after counting
        if k == 5: s[:] = np.frombuffer(b"AT", np.uint8)[np.arange(l) % 2]         # dinucleotide repeat: low complexity
        if k == 6: s[:] = ord("A")                                                 # mononucleotide
        if i % 401 == 7 and l > 3: q[int(rng.integers(0, l))] = 20                 # a byte below the offset (clamped)
        reads.append((b"@x", s.tobytes(), q.tobytes()))
    compare_engines(opt, reads, R=256 if L <= 256 else 1024, seg_size=700)


@pytest.mark.parametrize("L", [75, 150, 250])
def test_take_back_pass_under_load(L):
    """Most reads of this batch are rejected AFTER their post-trim cells were counted (-n 1 on reads with 2 % N), so nearly every chunk
    stages its qualities a second time and takes cells back -- with hundreds of chunks in flight on every CU, lower-case bases in the
    reads and a length that leaves the lanes of a chunk's last read reading behind the span.  The zero-increment adds of that pass once
    went wherever the byte behind the span pointed (profiles/r4c/restage_race.txt: a race that needed another wave's LDS-DMA to hit the same
    dword at the same moment -- this batch did NOT reproduce it on the build that had it, the re-seeded runs of profiles/fuzz.sh did);
    the whole counter block must match the oracle."""
    rng = np.random.Generator(np.random.PCG64([93, L, SEED]))
    opt = parse_args(["-u", "x", "-d", "y", "--min_L", "1", "--avg_q", "20", "-n", "1"])
    n = 48000
    s = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.choice(9, (n, L), p=[.2, .2, .2, .2, .02, .045, .045, .045, .045])]
    q = (rng.integers(28, 41, (n, L)) + 33).astype(np.uint8)
    q[::7, L - 5:] = 34
    reads = [(b"@x", s[i].tobytes(), q[i].tobytes()) for i in range(n)]
    compare_engines(opt, reads, seg_size=4000)


@pytest.mark.parametrize("L,args,kernel", [
    (150, [], "trim_lds"), (151, ["--adapter"], "trim_lds"), (100, ["--mode", "HARD", "-q", "10"], "trim_lds"), (125, ["--qc_only"], "trim_lds"),
    (128, [], "trim_lds"), (96, [], "trim_lds"), (75, [], "trim_lds"), (160, [], "trim_lds"), (157, ["--5trim_off"], "trim_lds"), (64, [], "trim_lds"), (50, ["--adapter"], "trim_lds"), (36, ["--mode", "BWA"], "trim_lds"),
    (75, ["--replace_to_N_q", "15"], "trim_filter_accumulate"),
    (150, ["--replace_to_N_q", "15"], "trim_filter_accumulate"), (128, ["--qc_only"], "trim_lds"), (192, ["--adapter"], "trim_lds"),
    (250, [], "trim_lds"), (251, ["--adapter", "--polyA"], "trim_lds"), (200, ["--mode", "BWA", "--avg_q", "20"], "trim_lds"), (161, [], "trim_lds"), (252, [], "trim_lds"),
    (253, [], "trim_lds"), (224, [], "trim_lds"), (256, [], "trim_lds"), (250, ["--replace_to_N_q", "15"], "trim_filter_accumulate"),
    (300, [], "trim_lds"), (301, ["--adapter", "--polyA"], "trim_lds"), (304, ["--mode", "HARD", "-q", "10"], "trim_lds"), (305, [], "trim_filter_accumulate"),
    (300, ["--replace_to_N_q", "15"], "trim_filter_accumulate"),
])
def test_dispatcher_picks_the_documented_trim_kernel(L, args, kernel):
    """DESIGN.md section 4 names the trim kernel of every (read length, option set) class; faqcs_kernel_report() says which one ran."""
    import ctypes as C

    rng = np.random.Generator(np.random.PCG64([5, L, SEED]))
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    reads = [(b"@x", make_uniform(rng, L), bytes((rng.integers(20, 41, L) + 33).astype(np.uint8))) for _ in range(300)]
    hip, _ = compare_engines(opt, reads, R=256 if L <= 256 else 1024)
    kt = capi.KernelTimes()
    assert hip.lib.faqcs_kernel_report(hip.ctx, C.byref(kt)) == 0
    assert (kt.trim_kernel or b"").decode() == kernel


def make_uniform(rng, L):
    return bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)])


def test_native_cli_pipes_a_report_script_to_R(tmp_path):
    """plot.cpp:93-515 process contract: without --trim_only the tables are written, an R script is piped into
    `R --vanilla --silent --slave`, and the tables are deleted afterwards unless --debug.  A stub `R` on PATH records its
    command line, its stdin and the files it could see."""
    import stat
    import subprocess

    rng = np.random.Generator(np.random.PCG64([404, SEED]))
    reads = random_batch(rng, 4000, 150, "adv")
    fq = tmp_path / "in.fastq"
    with open(fq, "wb") as f:
        for i, (_, s, q) in enumerate(reads):
            if s:
                f.write(b"@r%d\n%s\n+\n%s\n" % (i, s, q))
    stub_dir = tmp_path / "bin"
    stub_dir.mkdir()
    stub = stub_dir / "R"
    stub.write_text('#!/bin/sh\necho "$@" > "$FAQCS_TEST_R_OUT/args"\ncat > "$FAQCS_TEST_R_OUT/script"\nls "$FAQCS_TEST_R_DIR" > "$FAQCS_TEST_R_OUT/ls"\n')
    stub.chmod(stub.stat().st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    tables = ["qa.QC.quality.matrix", "QC.quality.matrix", "qa.QC.base.matrix", "QC.base.matrix", "qa.QC.for_qual_histogram.txt",
              "QC.for_qual_histogram.txt", "qa.QC.base_content.txt", "QC.base_content.txt", "qa.QC.length_count.txt", "QC.length_count.txt"]
    for mode, extra in (("plain", []), ("debug", ["--debug"]), ("trim_only", ["--trim_only"]), ("kmer", ["--kmer_rarefaction", "--split_size", "1000"])):
        out, cap = tmp_path / ("out_" + mode), tmp_path / ("cap_" + mode)
        cap.mkdir()
        env = dict(os.environ, PATH=str(stub_dir) + os.pathsep + os.environ["PATH"], FAQCS_TEST_R_OUT=str(cap), FAQCS_TEST_R_DIR=str(out))
        r = subprocess.run([_CLI_BIN, "-u", str(fq), "-d", str(out), "--ascii", "33"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        left = sorted(os.listdir(out))
        if mode == "trim_only":
            assert not (cap / "script").exists() and not any(t in left for t in tables)
            continue
        assert (cap / "args").read_text().split() == ["--vanilla", "--silent", "--slave"]
        script = (cap / "script").read_text()
        assert str(out / "QC_qc_report.pdf") in script and str(out / "QC.stats.txt") in script and "dev.off()" in script
        seen = (cap / "ls").read_text().split()
        want = tables + (["QC.kmerH.txt", "QC.Kmercount.txt"] if mode == "kmer" else [])
        assert all(t in seen for t in want) and "QC.stats.txt" in seen, seen          # R ran while the tables were there
        assert all((t in left) == (mode == "debug") for t in want), left                # and they are gone afterwards unless --debug


def test_native_cli_inflates_ordinary_gzip_in_parallel(tmp_path):
    """Ordinary single-member .fastq.gz inputs (what FaQCs users feed it; fastq.cpp:8-125 reads them through gzread) are inflated by a pool
    of threads (faqcs_pargz.h: guessed block starts, offset-encoding dictionaries, CRC-checked): the same output bytes as from the plain
    files and as through gzread (FAQCS_MI_NO_PARGZ=1); a cut file and a file with a wrong byte end the run like a failing gzread."""
    import gzip
    import hashlib
    import subprocess

    import make_fixtures

    rng = np.random.Generator(np.random.PCG64([606, SEED]))
    n = 2 * 32768 + 777
    texts = [[], []]
    for i in range(n):
        for m in (0, 1):
            s, q = make_fixtures._adv_read(rng, 150)
            if len(s) == 0:
                s, q = np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"IIII", np.uint8)
            texts[m].append(b"@p%d/%d\n%s\n+\n%s\n" % (i, m + 1, s.upper().tobytes() if hasattr(s, "upper") else s.tobytes(), q.tobytes()))
    plain, gz = [], []
    for m in (0, 1):
        data = b"".join(texts[m])
        plain.append(str(tmp_path / ("r%d.fastq" % (m + 1))))
        gz.append(str(tmp_path / ("r%d.fastq.gz" % (m + 1))))
        with open(plain[-1], "wb") as f:
            f.write(data)
        with open(gz[-1], "wb") as f:
            f.write(gzip.compress(data, 6))

    def run(tag, inputs, env_extra):
        out = tmp_path / tag
        env = dict(os.environ, FAQCS_MI_PARGZ_MIN="1", FAQCS_MI_PARGZ_THREADS="6", **env_extra)
        r = subprocess.run([_CLI_BIN, "-1", inputs[0], "-2", inputs[1], "-d", str(out), "--ascii", "33", "--trim_only"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        return {f: hashlib.md5(open(out / f, "rb").read()).hexdigest() for f in sorted(os.listdir(out))}

    want = run("plain", plain, {})
    assert run("pargz", gz, {}) == want
    assert run("gzread", gz, {"FAQCS_MI_NO_PARGZ": "1"}) == want
    data = open(gz[0], "rb").read()
    for tag, blob in (("cut", data[: len(data) // 2]), ("flip", data[: len(data) // 2] + bytes([data[len(data) // 2] ^ 0x21]) + data[len(data) // 2 + 1:])):
        bad = str(tmp_path / (tag + ".fastq.gz"))
        with open(bad, "wb") as f:
            f.write(blob)
        for env_extra in ({}, {"FAQCS_MI_NO_PARGZ": "1"}):
            r = subprocess.run([_CLI_BIN, "-u", bad, "-d", str(tmp_path / (tag + "_out" + str(len(env_extra)))), "--ascii", "33", "--trim_only"],
                               env=dict(os.environ, FAQCS_MI_PARGZ_MIN="1", **env_extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert r.returncode == 1 and b"Caught the error" in r.stderr, (tag, env_extra, r.stderr.decode()[-400:])


def _write_bgzf(path, data, block=60000, level=4):
    """BGZF (bgzip / htslib): gzip members of <= 64 KiB with a 'BC' extra subfield that holds the member's size - 1."""
    import struct
    import zlib

    with open(path, "wb") as f:
        for o in list(range(0, len(data), block)) + [None]:
            raw = b"" if o is None else data[o:o + block]   # (the last, empty member is BGZF's end-of-file marker)
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(raw) + c.flush()
            bsize = 12 + 6 + len(body) + 8
            f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize - 1))
            f.write(body)
            f.write(struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw) & 0xffffffff))


def test_native_cli_reads_bgzf_in_parallel(tmp_path):
    """bgzip-compressed inputs are inflated member by member by a thread pool (BgzfReader in faqcs_cli.cpp): same output bytes
    as from the plain files, and as from the same .gz read through gzread (FAQCS_MI_NO_BGZF=1)."""
    import hashlib
    import struct
    import subprocess

    import make_fixtures

    rng = np.random.Generator(np.random.PCG64([505, SEED]))
    n = 3 * 32768 + 1234
    texts = [[], []]
    for i in range(n):
        for m in (0, 1):
            s, q = make_fixtures._adv_read(rng, 120)
            if len(s) == 0:
                s, q = np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"IIII", np.uint8)
            texts[m].append(b"@p%d/%d\n%s\n+\n%s\n" % (i, m + 1, s.tobytes(), q.tobytes()))
    plain, gz = [], []
    for m in (0, 1):
        data = b"".join(texts[m])
        plain.append(str(tmp_path / ("r%d.fastq" % (m + 1))))
        gz.append(str(tmp_path / ("r%d.fastq.gz" % (m + 1))))
        with open(plain[-1], "wb") as f:
            f.write(data)
        _write_bgzf(gz[-1], data)

    def run(tag, inputs, env_extra):
        out = tmp_path / tag
        env = dict(os.environ, **env_extra)
        r = subprocess.run([_CLI_BIN, "-1", inputs[0], "-2", inputs[1], "-d", str(out), "--ascii", "33", "--adapter", "--trim_only"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-800:]
        return {f: hashlib.md5(open(out / f, "rb").read()).hexdigest() for f in sorted(os.listdir(out))}

    want = run("plain", plain, {})
    assert run("bgzf", gz, {}) == want
    assert run("gzread", gz, {"FAQCS_MI_NO_BGZF": "1"}) == want
    # unpaired, and a file whose last member is cut off: the run ends like a failing gzread (no hang, no crash)
    data = open(gz[0], "rb").read()
    cut = str(tmp_path / "cut.fastq.gz")
    with open(cut, "wb") as f:
        f.write(data[: len(data) // 2])
    r = subprocess.run([_CLI_BIN, "-u", cut, "-d", str(tmp_path / "cut_out"), "--ascii", "33", "--trim_only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Caught the error fastq.cpp:next_read:" in r.stderr, r.stderr.decode()[-400:]  # (never a silent short run)
    # ... and so do the same bytes read through gzread, and the reference itself when it is here
    r2 = subprocess.run([_CLI_BIN, "-u", cut, "-d", str(tmp_path / "cut_out2"), "--ascii", "33", "--trim_only"], env=dict(os.environ, FAQCS_MI_NO_BGZF="1"),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r2.returncode == 1 and b"Caught the error fastq.cpp:next_read:" in r2.stderr
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "FaQCs_ref")
    if os.path.exists(ref):
        r3 = subprocess.run([ref, "-u", cut, "-d", str(tmp_path / "cut_ref"), "--ascii", "33", "--trim_only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        msg = lambda e: e.decode().split("next_read:")[-1].strip()  # noqa: E731  (the reference's text carries its source path)
        assert r3.returncode == 1 and b"fastq.cpp:next_read:" in r3.stderr and msg(r3.stderr) == msg(r.stderr) == msg(r2.stderr)
    # a wrong CRC32 in a member (the deflate stream itself intact): gzread checks the trailer, so does the BGZF reader
    raw = bytearray(data)
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1
    raw[bsize - 8] ^= 0x01
    crc = str(tmp_path / "crc.fastq.gz")
    with open(crc, "wb") as f:
        f.write(raw)
    r = subprocess.run([_CLI_BIN, "-u", crc, "-d", str(tmp_path / "crc_out"), "--ascii", "33", "--trim_only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Caught the error fastq.cpp:next_read:" in r.stderr
    # bgzip members followed by ordinary gzip members (cat a.bgz b.gz): every read arrives
    import gzip

    half = len(texts[0]) // 2
    mixed = str(tmp_path / "mixed.fastq.gz")
    _write_bgzf(mixed, b"".join(texts[0][:half]))
    blob = open(mixed, "rb").read()[:-28]  # (without the end-of-file marker: the gzip members follow directly)
    with open(mixed, "wb") as f:
        f.write(blob)
        f.write(gzip.compress(b"".join(texts[0][half:half + 1000])))
        f.write(gzip.compress(b"".join(texts[0][half + 1000:])))
    def run_u(tag, path):
        out = tmp_path / tag
        r_ = subprocess.run([_CLI_BIN, "-u", path, "-d", str(out), "--ascii", "33", "--trim_only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r_.returncode == 0, r_.stderr.decode()[-800:]
        return {f: hashlib.md5(open(out / f, "rb").read()).hexdigest() for f in sorted(os.listdir(out))}
    assert run_u("mixed_out", mixed) == run_u("plain_u_out", plain[0])


# ---- the rare paths again, under seeds nobody has seen ------------------------------------------------------------------------------
# The suite is pinned to FAQCS_TEST_SEED=0; a race that shows once in a thousand chunks passed it twice (round 4).  Every run of the suite
# therefore repeats the tests that take the rare paths -- the take-back pass, the two-class transition counter, every lane geometry, the
# k-mer kernels' deferred reads / overflow area / mixed read lengths -- under 24 seeds derived from a session seed, which is printed:
# FAQCS_SESSION_SEED=<n> replays a failing run, FAQCS_TEST_SEED=<the printed seed> replays one turn through the plain tests.
SESSION_SEED = int(os.environ.get("FAQCS_SESSION_SEED", "0")) or (int(__import__("time").time()) ^ (os.getpid() << 8)) & 0x3fffffff


@pytest.mark.parametrize("turn", range(24))
def test_rare_paths_under_fresh_seeds(turn, monkeypatch, capsys):
    import sys

    seed = (SESSION_SEED * 1000003 + 7919 * turn) % (1 << 30)
    with capsys.disabled():
        print("\n[fresh seeds] FAQCS_SESSION_SEED=%d turn %d -> FAQCS_TEST_SEED=%d" % (SESSION_SEED, turn, seed), flush=True)
    me = sys.modules[__name__]
    monkeypatch.setattr(me, "SEED", seed)
    for L in (75, 150, 250):
        test_take_back_pass_under_load(L)
    for args in ([], ["--lc", "0.3", "--min_L", "20"]):
        test_at_rich_reads_match_oracle(args)
    widths = [("adv", 157), ("ragged", 104), ("adv", 75), ("adv", 252), ("adv", 300), ("ragged", 30), ("adv", 320), ("ragged", 224)]
    kind, maxlen = widths[turn % len(widths)]
    for args in (OPTION_SETS[0], OPTION_SETS[1], OPTION_SETS[10], OPTION_SETS[13], OPTION_SETS[15], OPTION_SETS[23]):
        test_every_kernel_width_matches_oracle(args, kind, maxlen)
    test_chunks_of_equal_length_reads_in_padded_rows((64, 128, 224, 96)[turn % 4], ["--lc", "0.5", "-n", "1"])
    test_kmer_submissions_either_side_of_256_bases_share_their_keys(turn)
    test_kmer_repeats_and_a_partition_larger_than_its_slice(turn, monkeypatch)
