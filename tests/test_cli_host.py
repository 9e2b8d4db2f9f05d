"""CPU-side checks of host code in the native command line that needs no device: the BGZF reader and the report script
(`faqcs_mi --bgzf_cat`, `faqcs_mi --report_script`; faqcs_cli.cpp:host_self_check)."""
import os
import struct
import subprocess
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
