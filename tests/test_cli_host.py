"""CPU-side checks of host code in the native command line that needs no device: the BGZF reader and the report script
(`faqcs_mi --bgzf_cat`, `faqcs_mi --report_script`; faqcs_cli.cpp:host_self_check)."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "faqcs_amd", "faqcs_mi")

pytestmark = pytest.mark.skipif(not os.path.exists(CLI), reason="faqcs_mi is not built (python -c 'import __graft_entry__ as g; g.build()')")


def write_bgzf(path, data, block, level=4, eof_marker=True):
    with open(path, "wb") as f:
        for o in list(range(0, len(data), block)) + ([None] if eof_marker else []):
            raw = b"" if o is None else data[o:o + block]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(raw) + c.flush()
            f.write(struct.pack("<4BI2BH2BHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, 12 + 6 + len(body) + 8 - 1))
            f.write(body)
            f.write(struct.pack("<II", zlib.crc32(raw) & 0xffffffff, len(raw)))


@pytest.mark.parametrize("size,block", [(0, 1000), (1, 1000), (70_000, 65280), (5_000_000, 65280), (3_000_000, 777), (2_000_000, 65536)])
def test_bgzf_reader_returns_the_bytes_in_order(tmp_path, size, block):
    """Members of every size up to the 64 KiB limit, more members than one task holds (64), an empty file body, with and without
    the end-of-file marker: the thread pool hands the inflated bytes back in file order."""
    rng = np.random.Generator(np.random.PCG64([9, size, block]))
    data = bytes(rng.integers(33, 75, size, dtype=np.uint8))  # quality-like bytes: compressible but not trivially
    for marker in (True, False):
        if size == 0 and not marker:
            continue
        p = str(tmp_path / ("x%d.gz" % marker))
        write_bgzf(p, data, block, eof_marker=marker)
        r = subprocess.run([CLI, "--bgzf_cat", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert r.stdout == data
        if size:  # a BGZF file is also plain multi-member gzip: zlib decodes its first member
            assert zlib.decompressobj(31).decompress(open(p, "rb").read()) == data[:block]


def test_bgzf_reader_stops_at_a_corrupt_member(tmp_path):
    data = bytes(np.random.Generator(np.random.PCG64(3)).integers(33, 75, 400_000, dtype=np.uint8))
    p = str(tmp_path / "bad.gz")
    write_bgzf(p, data, 60000)
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0xFF
    open(p, "wb").write(raw)
    r = subprocess.run([CLI, "--bgzf_cat", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode in (0, 3) and data.startswith(r.stdout) and len(r.stdout) < len(data)
    plain = str(tmp_path / "plain.gz")
    import gzip
    with gzip.open(plain, "wb") as f:
        f.write(data)
    r = subprocess.run([CLI, "--bgzf_cat", plain], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 2  # a single-member gzip file is not BGZF: the command line reads it through gzread


def test_report_script_names_the_files_of_the_run(tmp_path):
    out = str(tmp_path / "o u\"t")
    r = subprocess.run([CLI, "--report_script", "-u", "x.fastq", "-d", out, "--prefix", "S1", "--qc_only"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    s = r.stdout.decode()
    assert 'pdf_file <- "%s/S1_qc_report.pdf"' % out.replace('"', '\\"') in s
    assert 'stats_file <- "%s/S1.stats.txt"' % out.replace('"', '\\"') in s
    assert "qc_only <- TRUE" in s and s.rstrip().endswith('quit(save = "no")')
    assert s.count("page({") == 12 and s.count("{") == s.count("}") and s.count("(") == s.count(")")


def test_bgzf_followed_by_ordinary_gzip_members_is_read_to_the_end(tmp_path):
    """`cat a.bgz b.gz`: gzread (the reference's reader) decodes every member whatever its framing; the BGZF reader hands what
    follows its last whole member to zlib instead of ending the input (100 001 of 150 002 bytes, exit code 0 downstream)."""
    import gzip

    rng = np.random.Generator(np.random.PCG64(17))
    a = bytes(rng.integers(33, 75, 100_001, dtype=np.uint8))
    b = bytes(rng.integers(33, 75, 50_001, dtype=np.uint8))
    c = bytes(rng.integers(33, 75, 777, dtype=np.uint8))
    p = str(tmp_path / "mixed.gz")
    write_bgzf(p, a, 40000, eof_marker=False)
    with open(p, "ab") as f:
        f.write(gzip.compress(b))
        f.write(gzip.compress(c))  # (two ordinary members)
    assert gzip.open(p).read() == a + b + c
    r = subprocess.run([CLI, "--bgzf_cat", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == a + b + c


def _zlib_gzread(path):
    """What zlib's own gzread() -- the call the reference reads FASTQ through (fastq.cpp:34-52) -- delivers for a file:
    (bytes, failed).  Through ctypes, so the comparison is with zlib itself and not with Python's gzip module (which is stricter:
    it raises on trailing garbage that zlib ignores)."""
    import ctypes as C

    z = C.CDLL("libz.so.1")
    z.gzopen.restype = C.c_void_p
    z.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    z.gzread.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    z.gzerror.restype = C.c_char_p
    z.gzerror.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    z.gzclose.argtypes = [C.c_void_p]
    f = z.gzopen(path.encode(), b"rb")
    assert f
    buf = C.create_string_buffer(1 << 20)
    out, failed = bytearray(), False
    while True:
        n = z.gzread(f, buf, len(buf))
        if n > 0:
            out += buf.raw[:n]
        if n < (1 << 20):
            err = C.c_int(0)
            z.gzerror(f, C.byref(err))
            failed = n < 0 or err.value not in (0, 1)  # Z_OK / Z_STREAM_END
            break
    z.gzclose(f)
    return bytes(out), failed


def test_bgzf_reader_reports_what_gzread_reports(tmp_path):
    """A wrong CRC32 trailer, a truncated last member, a broken header behind the last member: gzread fails on each (the reference
    then throws `Unable to read header`, fastq.cpp:34-41) and --bgzf_cat returns 3.  Bytes behind the last member that do not start
    with the gzip magic are trailing garbage to zlib: gzread delivers everything in front of them and reports NO error (gz_look in
    zlib's gzread.c), so the reference exits 0 there and so does --bgzf_cat.  Every case is compared with zlib's gzread itself."""
    data = bytes(np.random.Generator(np.random.PCG64(5)).integers(33, 75, 300_000, dtype=np.uint8))
    good = str(tmp_path / "good.gz")
    write_bgzf(good, data, 50000)
    raw = open(good, "rb").read()
    # the first member's CRC32 (8 bytes before the second member's header)
    first = 12 + 6 + struct.unpack_from("<H", raw, 16)[0] + 1 - (12 + 6)  # BSIZE + 1 = the member's size
    bad_crc = bytearray(raw)
    bad_crc[first - 8] ^= 0x01
    cases = {"crc": bytes(bad_crc), "truncated": raw[:len(raw) - 28 - 40], "garbage": raw + b"this is not gzip at all" * 3,
             "one_byte": raw + b"\x1f", "magic_then_junk": raw + b"\x1f\x8b" + b"junk that is no deflate stream" * 4}
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        want, want_failed = _zlib_gzread(p)
        r = subprocess.run([CLI, "--bgzf_cat", p], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == (3 if want_failed else 0), (name, r.returncode, want_failed)
        if want_failed:
            assert data.startswith(r.stdout), name  # (whole members in front of the failure may or may not have been delivered)
        else:
            assert r.stdout == want == data, name
    assert not _zlib_gzread(str(tmp_path / "garbage.gz"))[1] and not _zlib_gzread(str(tmp_path / "one_byte.gz"))[1]
    assert _zlib_gzread(str(tmp_path / "crc.gz"))[1] and _zlib_gzread(str(tmp_path / "truncated.gz"))[1] and _zlib_gzread(str(tmp_path / "magic_then_junk.gz"))[1]


def test_report_script_is_structurally_sound():
    """No R in the build image: the script the report step pipes into `R --vanilla --silent --slave` has never been rendered (README,
    INTEGRATION.md say so).  What CAN be checked without R: every bracket / quote is balanced outside comments and strings, and every
    table the script reads is one that write_tables() writes (faqcs_cli.cpp:table_files), under the name the script builds."""
    import re

    r = subprocess.run([CLI, "--report_script", "-1", "a.fastq", "-2", "b.fastq", "-d", "/tmp/out dir", "--prefix", "S1", "--kmer_rarefaction"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    script = r.stdout.decode()
    # a tiny R lexer: strings ("...", '...' with backslash escapes), comments (# to the end of the line), brackets
    stack, i, n = [], 0, len(script)
    pairs = {")": "(", "]": "[", "}": "{"}
    while i < n:
        c = script[i]
        if c == "#":
            while i < n and script[i] != "\n":
                i += 1
            continue
        if c in "\"'":
            q, i = c, i + 1
            while i < n and script[i] != q:
                i += 2 if script[i] == "\\" else 1
            assert i < n, "unterminated string"
        elif c in "([{":
            stack.append((c, script.count("\n", 0, i) + 1))
        elif c in ")]}":
            assert stack and stack[-1][0] == pairs[c], "unbalanced %r on line %d" % (c, script.count("\n", 0, i) + 1)
            stack.pop()
        i += 1
    assert not stack, "unclosed %r opened on line %d" % stack[-1]
    # the tables: names handed to pre() / post() / both() vs the files write_tables() produces
    used = set(re.findall(r'\b(?:pre|post|both)\("([^"]+)"\)', script))
    written = {"quality.matrix", "base.matrix", "for_qual_histogram.txt", "base_content.txt", "length_count.txt", "kmerH.txt", "Kmercount.txt"}
    assert used and used <= written, used - written
    assert {"quality.matrix", "base.matrix", "length_count.txt", "base_content.txt", "for_qual_histogram.txt"} <= used
    assert 'paste0("qa.", "S1", ".", name)' in script and 'file.path("/tmp/out dir"' in script
    # every function the script defines is used, every plotting page is closed
    for fn in re.findall(r"^(\w+) <- function", script, flags=re.M):
        assert len(re.findall(r"\b%s\b" % re.escape(fn), script)) >= 2, fn  # (its definition and at least one use, called or passed on)
    assert "dev.off()" in script


def test_terminal_n_flags_of_a_packed_arena():
    """faqcs_batch.terminal_n as the Python driver fills it (bit 0: first base is an upper-case N, bit 1: last base is): against a
    plain loop, on reads that are empty, one base long, all N, lower-case n at the ends."""
    import numpy as np

    from faqcs_amd import driver

    rng = np.random.Generator(np.random.PCG64(5))
    reads = []
    for k in range(500):
        L = int(rng.integers(0, 40))
        s = np.frombuffer(b"ACGTNn", np.uint8)[rng.integers(0, 6, L)].tobytes()
        reads.append((b"@r", s, b"I" * L))
    reads += [(b"@r", b"", b""), (b"@r", b"N", b"I"), (b"@r", b"n", b"I"), (b"@r", b"NN", b"II"), (b"@r", b"AN", b"II"), (b"@r", b"NA", b"II")]
    seq, qual, offset, seg = driver.pack_segments([reads[:300], reads[300:]])
    got = driver.terminal_n_flags(seq, offset)
    want = np.array([(1 if s[:1] == b"N" else 0) | (2 if s[-1:] == b"N" else 0) for _, s, _ in reads], dtype=np.uint8)
    assert got.dtype == np.uint8 and (got == want).all()
    assert len(driver.terminal_n_flags(seq[:0], offset[:1])) == 0


def test_worker_process_only_outside_profilers():
    """faqcs_mi hands the command to a forked worker so that its caller gets control back when the outputs are complete
    (faqcs_cli.cpp:main); never under a profiler (its library has initialised the GPU before main()), and a preloaded library
    that is not a profiler -- a launcher's guard, a malloc replacement -- does not switch the worker off."""
    def plan(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "FAQCS_MI_NO_FORK") and not k.startswith(("ROCP", "HSA_TOOLS"))}
        # (LD_PRELOAD of a file that is not there only makes ld.so print a warning: the decision reads the variable, not the file)
        r = subprocess.run([CLI, "--process_plan"], env=dict(e, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        assert r.returncode == 0
        return r.stdout.decode().strip()

    assert plan() == "worker"
    assert plan(LD_PRELOAD="/nonexistent/libexecguard.so") == "worker"
    assert plan(LD_PRELOAD="/opt/rocm/lib/librocprofiler-sdk-tool.so") == "one process"
    assert plan(ROCPROF_OUTPUT_PATH="/tmp/x") == "one process"
    assert plan(HSA_TOOLS_LIB="libx.so") == "one process"
    assert plan(FAQCS_MI_NO_FORK="1") == "one process"
    assert plan(FAQCS_MI_NO_FORK="0") == "worker"


# ---- parallel inflate of ordinary gzip files (faqcs_pargz.h; `faqcs_mi --pargz_cat`: host only) --------------------------------------
def _fastq_text(n, L=150, seed=5):
    import numpy as np

    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGTN", np.uint8)
    recs = []
    for i in range(n):
        l = int(rng.integers(L // 2, L + 1))
        recs.append(b"@SYN:%09d/1 some text\n" % i + bases[rng.choice(5, l, p=[.249, .249, .249, .249, .004])].tobytes() + b"\n+\n"
                    + (rng.integers(2, 41, l) + 33).astype(np.uint8).tobytes() + b"\n")
    return b"".join(recs)


def _pargz(path, threads, piece):
    r = subprocess.run([CLI, "--pargz_cat", str(path), str(threads), str(piece)], capture_output=True, timeout=300)
    return r.returncode, r.stdout


@pytest.mark.parametrize("level", [1, 6, 9])
def test_parallel_inflate_equals_zlib(level, tmp_path):
    """An ordinary single-member .fastq.gz through the speculative parallel inflate (block starts guessed, two inflates with offset-encoding
    dictionaries, markers patched in order, CRC and length checked): byte-identical to zlib's own output, for every compression level,
    piece size and thread count -- pieces far smaller than a deflate block included (most of them find no block start and are run through)."""
    import gzip

    text = _fastq_text(40000, seed=level)
    p = tmp_path / "r.fq.gz"
    p.write_bytes(gzip.compress(text, level))
    for threads, piece in ((4, 300000), (3, 70001), (8, 1 << 20), (2, 20000)):
        rc, out = _pargz(p, threads, piece)
        assert rc == 0 and out == text, "level %d threads %d piece %d: rc %d, %d bytes" % (level, threads, piece, rc, len(out))
    # round 5's scheme (two zlib passes per piece with offset-encoding dictionaries) is still there for A/B runs
    r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "300000"], capture_output=True, timeout=300, env=dict(os.environ, FAQCS_MI_PARGZ_TWO_PASS="1"))
    assert r.returncode == 0 and r.stdout == text


def test_parallel_inflate_on_odd_files(tmp_path):
    """Stored blocks only (level 0: no dynamic block to find), concatenated members (the first in parallel, the rest through zlib's gzip decoder, as
    gzread reads on), trailing garbage (ends the data, as in zlib), a header with a file name, an empty member, a file that is not ASCII."""
    import gzip
    import io

    text = _fastq_text(12000, seed=77)
    cases = {}
    cases["stored"] = (gzip.compress(text, 0), text)
    cases["two_members"] = (gzip.compress(text, 6) + gzip.compress(text[:100000], 9), text + text[:100000])
    cases["garbage_behind"] = (gzip.compress(text, 6) + b"\\0\\0not gzip", text)
    b = io.BytesIO()
    with gzip.GzipFile(filename="reads_R1.fastq", mode="wb", fileobj=b, compresslevel=5) as f:
        f.write(text)
    cases["named"] = (b.getvalue(), text)
    cases["empty_then_data"] = (gzip.compress(b"", 6) + gzip.compress(text, 6), text)
    for name, (blob, want) in cases.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(blob)
        rc, out = _pargz(p, 4, 150000)
        assert rc == 0 and out == want, "%s: rc %d, %d bytes (want %d)" % (name, rc, len(out), len(want))
    # text that is not ASCII: the 16-bit symbols of round 6 keep markers out of band, so it inflates like any other file (round 5's
    # two-dictionary scheme checked the first piece only and took a later byte >= 128 for a marker); that scheme itself still refuses it
    p = tmp_path / "binary.gz"
    data = (bytes(range(256)) * 997 + text[:70000]) * 4
    p.write_bytes(gzip.compress(data, 6))
    rc, out = _pargz(p, 4, 150000)
    assert rc == 0 and out == data
    r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "150000"], capture_output=True, timeout=300, env=dict(os.environ, FAQCS_MI_PARGZ_TWO_PASS="1"))
    assert r.returncode == 4 and r.stdout == b""


def test_parallel_inflate_of_concatenated_members(tmp_path):
    """`cat lane1.fastq.gz lane2.fastq.gz ...` (gzread reads it as one stream, fastq.cpp:34): every member that is long enough starts the
    parallel reader again at its header, short ones go through zlib in the consumer, in any order -- and the speculation of one member
    does not run on through the next: ADVICE r5 measured twice the decompressed text resident for a two-member file; the peak RSS of a
    two-member file now stays near that of the same text as one member."""
    import gzip
    import resource

    a, b, c = _fastq_text(30000, seed=11), _fastq_text(200, seed=12), _fastq_text(25000, seed=13)
    blob = gzip.compress(a, 6) + gzip.compress(b, 9) + gzip.compress(b"", 6) + gzip.compress(c, 1) + gzip.compress(b[:5000], 6)
    want = a + b + c + b[:5000]
    p = tmp_path / "lanes.gz"
    p.write_bytes(blob)
    for rearm in ("100000", "1", str(1 << 40)):  # long members in parallel / every member / only the first
        r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "150000"], capture_output=True, timeout=300, env=dict(os.environ, FAQCS_MI_PARGZ_REARM=rearm))
        assert r.returncode == 0 and r.stdout == want, "rearm %s: rc %d, %d bytes (want %d)" % (rearm, r.returncode, len(r.stdout), len(want))
    # damage in the SECOND long member is still an error
    bad = bytearray(blob)
    bad[len(gzip.compress(a, 6)) + len(gzip.compress(b, 9)) + 20 + 40000] ^= 0x33
    p2 = tmp_path / "lanes_bad.gz"
    p2.write_bytes(bytes(bad))
    r = subprocess.run([CLI, "--pargz_cat", str(p2), "4", "150000"], capture_output=True, timeout=300, env=dict(os.environ, FAQCS_MI_PARGZ_REARM="100000"))
    assert r.returncode == 3  # (caught by the member's CRC at the latest, like a flipped byte under gzread)

    # peak RSS: one member of 2 x text against two members of text each (the default re-arm threshold: the second member is read by zlib)
    big = _fastq_text(150000, seed=21)
    one, two = tmp_path / "one.gz", tmp_path / "two.gz"
    one.write_bytes(gzip.compress(big + big, 1))
    two.write_bytes(gzip.compress(big, 1) + gzip.compress(big, 1))

    def peak(path, **env):
        script = "import resource, subprocess, sys; r = subprocess.run(sys.argv[1:], stdout=subprocess.DEVNULL); print(r.returncode, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss)"
        out = subprocess.run([sys.executable, "-c", script, CLI, "--pargz_cat", str(path), "8", "0"], capture_output=True, timeout=300, env=dict(os.environ, **env)).stdout.split()
        assert out[0] == b"0"
        return int(out[1])

    rss_one, rss_two, rss_two_par = peak(one), peak(two), peak(two, FAQCS_MI_PARGZ_REARM="1000000")
    assert rss_two < 1.3 * rss_one + 16384 and rss_two_par < 1.3 * rss_one + 16384, (rss_one, rss_two, rss_two_par, len(big))


def test_parallel_inflate_reports_damage(tmp_path):
    """A truncated file and a file with a flipped byte end with an error (exit 3) after a prefix of the true text, never with different bytes
    and success: a piece ON the chain that fails ends the input there, and a wrong byte that still inflates is caught by the member's CRC."""
    import gzip

    text = _fastq_text(30000, seed=3)
    blob = gzip.compress(text, 6)
    p = tmp_path / "cut.gz"
    p.write_bytes(blob[: len(blob) * 2 // 3])
    rc, out = _pargz(p, 4, 200000)
    assert rc == 3 and text.startswith(out) and len(out) < len(text)
    for at in (len(blob) // 2, len(blob) // 3 + 17, len(blob) - 6):
        bad = bytearray(blob)
        bad[at] ^= 0x5A
        p = tmp_path / ("flip%d.gz" % at)
        p.write_bytes(bytes(bad))
        rc, out = _pargz(p, 4, 200000)
        assert rc == 3, "flip at %d: rc %d" % (at, rc)


def test_own_inflate_equals_zlib_under_the_sanitizers(tmp_path):
    """The repository's own inflate (faqcs_pargz.h: byte output for BGZF members, 16-bit symbols for pieces of ordinary gzip files) against
    zlib on generated streams -- every level and strategy, stored / fixed / dynamic blocks, flushes, starts at block boundaries inside a
    stream, damaged copies -- in a build with AddressSanitizer and UBSan (tools/inflate_fuzz.cpp says what is compared)."""
    exe = str(tmp_path / "inflate_fuzz")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tools", "inflate_fuzz.cpp"), "-lz"], capture_output=True, timeout=600)
    if r.returncode != 0 and b"asan" in r.stderr.lower():
        pytest.skip("no sanitizer runtime in this image")
    assert r.returncode == 0, r.stderr.decode()
    seed = os.environ.get("FAQCS_TEST_SEED", "20261004")
    r = subprocess.run([exe, seed, "250"], capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr).decode()[-2000:]
    assert b"250 streams equal to zlib's" in r.stdout


@pytest.mark.parametrize("level", [0, 1, 6, 9])
def test_bgzf_members_by_the_own_decoder_and_by_zlib(tmp_path, level):
    """BGZF members are inflated by the repository's decoder (FAQCS_MI_BGZF_ZLIB=1: by zlib): the same bytes either way, for stored blocks
    (level 0, where a member's deflate data ends with its last stored byte), and the same prefix and exit code for a damaged file."""
    rng = np.random.Generator(np.random.PCG64([11, level]))
    data = _fastq_text(9000, seed=level) + bytes(rng.integers(0, 256, 100_000, dtype=np.uint8)) + b"I" * 70000
    p = str(tmp_path / "x.gz")
    write_bgzf(p, data, 65280, level=level)
    outs = []
    for env in ({}, {"FAQCS_MI_BGZF_ZLIB": "1"}):
        r = subprocess.run([CLI, "--bgzf_cat", p], capture_output=True, timeout=120, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout == data, (env, r.returncode, len(r.stdout))
    raw = bytearray(open(p, "rb").read())
    for at in (len(raw) // 3, len(raw) // 2 + 5, len(raw) - 40):
        bad = bytearray(raw)
        bad[at] ^= 0x21
        q = str(tmp_path / "bad.gz")
        open(q, "wb").write(bad)
        outs = [subprocess.run([CLI, "--bgzf_cat", q], capture_output=True, timeout=120, env=dict(os.environ, **env)) for env in ({}, {"FAQCS_MI_BGZF_ZLIB": "1"})]
        assert outs[0].returncode == outs[1].returncode and outs[0].stdout == outs[1].stdout, at
        assert data.startswith(outs[0].stdout)


def test_crc_by_carry_less_multiplication_is_used_and_can_be_switched_off(tmp_path):
    """A member's CRC is computed with PCLMULQDQ where the CPU has it (tools/inflate_fuzz.cpp's build checks it against zlib on random
    lengths through the whole-file tests; here: the same verdicts with FAQCS_MI_NO_PCLMUL=1, i.e. through zlib's crc32)."""
    import gzip

    text = _fastq_text(40000, seed=8)
    p = tmp_path / "t.gz"
    p.write_bytes(gzip.compress(text, 6))
    for env in ({}, {"FAQCS_MI_NO_PCLMUL": "1"}):
        r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "150000"], capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 0 and r.stdout == text
    blob = bytearray(p.read_bytes())
    blob[-6] ^= 1  # the CRC in the trailer
    p.write_bytes(bytes(blob))
    for env in ({}, {"FAQCS_MI_NO_PCLMUL": "1"}):
        r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "150000"], capture_output=True, timeout=300, env=dict(os.environ, **env))
        assert r.returncode == 3


def test_parallel_inflate_behind_a_slow_consumer_stays_parallel(tmp_path):
    """When the consumer is the slower side every piece that may be in flight is, and a worker that finishes the last of them finds the
    next piece unclaimed.  It must END there -- the block boundary it stands at becomes that piece's start -- and not inflate on through
    it: round 6's first version did, and behind a slow consumer one worker after the other ran on to the end of the file (2.5 instead of
    11 M reads/s on the GPU box whenever fewer pieces were allowed in flight, profiles/r6n/e2e_gz_knobs_before_the_fix.txt).  Counted here: the pieces on the chain."""
    import gzip
    import re

    text = _fastq_text(60000, seed=12)
    p = tmp_path / "slow.gz"
    p.write_bytes(gzip.compress(text, 6))
    for window in ("6", "9", "14"):
        r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "100000"], capture_output=True, timeout=300,
                           env=dict(os.environ, FAQCS_MI_PARGZ_WINDOW=window, FAQCS_PARGZ_CAT_SLEEP_US="3000", FAQCS_PARGZ_STATS="1"))
        assert r.returncode == 0 and r.stdout == text
        m = re.search(rb"pieces (\d+) \(of \d+ bytes\), on the chain (\d+)", r.stderr)
        assert m, r.stderr
        n_pieces, on_chain = int(m.group(1)), int(m.group(2))
        assert n_pieces > 40 and on_chain >= n_pieces - 2, (window, n_pieces, on_chain)


def test_parallel_inflate_leaves_very_compressible_text_to_the_serial_reader(tmp_path):
    """deflate reaches 1 032 : 1 and every worker's piece grows at once: a file whose first piece already inflates more than 24 : 1 is not
    taken by the parallel reader (exit code 4 of --pargz_cat: "not eligible"; faqcs_mi then reads it through gzread, in constant memory),
    and a stretch of more than 128 : 1 behind an ordinary beginning ends the input with an error instead of taking gigabytes."""
    import gzip

    p = tmp_path / "zeros.gz"
    p.write_bytes(gzip.compress(b"N" * 40_000_000, 6))
    r = subprocess.run([CLI, "--pargz_cat", str(p), "4", "0"], capture_output=True, timeout=300)
    assert r.returncode == 4, (r.returncode, r.stderr[-300:])
    text = _fastq_text(40000, seed=4)
    q = tmp_path / "mixed.gz"
    q.write_bytes(gzip.compress(text + b"N" * 300_000_000 + text, 6))
    r = subprocess.run([CLI, "--pargz_cat", str(q), "4", "65536"], capture_output=True, timeout=600)
    assert r.returncode == 0 and len(r.stdout) == 2 * len(text) + 300_000_000  # (60 MB per 64 KB piece: below the bound)
    r = subprocess.run([CLI, "--pargz_cat", str(q), "4", "65536"], capture_output=True, timeout=600, env=dict(os.environ, FAQCS_MI_PARGZ_SYM_LIMIT="20000000"))
    assert r.returncode == 3 and len(r.stdout) < 2 * len(text) + 300_000_000 and (text + b"N" * 300_000_000).startswith(r.stdout)
