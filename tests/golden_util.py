"""Shared helpers for the golden-vector tests (CPU: OracleEngine, GPU: HipEngine)."""
import hashlib
import io
import json
import os

import make_fixtures
from make_golden import ARTIFACT_FASTA, CASES_DIR, HERE as GOLDEN_DIR, md5_file

from faqcs_amd import driver


def case_names(slow=True):
    names = sorted(f[:-5] for f in os.listdir(CASES_DIR) if f.endswith(".json"))
    return names


def load_case(name):
    with open(os.path.join(CASES_DIR, name + ".json")) as f:
        return json.load(f)


def exit_code_matches(rc, want):
    """The reference throws `const char*` for a quality above Q41 from inside its OpenMP region, where nothing catches it:
    the process dies of SIGABRT (exit -6).  A drop-in reports the same message and fails cleanly; any failure status is
    accepted where the reference was killed by a signal."""
    return rc == want or (want < 0 and rc != 0)


def fixture_paths(fixture, cache):
    if fixture == "example":
        return os.path.join(GOLDEN_DIR, "example_1.fastq.gz"), os.path.join(GOLDEN_DIR, "example_2.fastq.gz")
    return make_fixtures.materialise(fixture, cache)


def _compare_big_texts(case, outdir, have):
    """Text outputs the case holds as md5 + size + line count (make_golden.BIG_TEXT: the position tables of very long reads)."""
    bad = []
    for fn, meta in case.get("text_md5", {}).items():
        if fn not in have:
            bad.append("missing " + fn)
            continue
        with open(os.path.join(outdir, fn), "rb") as f:
            data = f.read()
        if hashlib.md5(data).hexdigest() != meta["md5"]:
            bad.append("%s: md5 mismatch (bytes %d vs %d, lines %d vs %d)" % (fn, len(data), meta["bytes"], data.count(b"\n"), meta["lines"]))
    return bad


def case_max_read_length(case):
    """Row capacity the Python driver gives its engine for a case (the native CLI always uses FAQCS_MAX_READ_LENGTH)."""
    return 32767 if case["fixture"] == "long8k" else 1024


def run_case(case, cache, tmp_path, engine_factory, **kw):
    """Runs our host driver with the given engine on the case's command line; returns a list of
    human-readable mismatches against the reference outputs stored in the case (empty == parity)."""
    p1, p2 = fixture_paths(case["fixture"], cache)
    assert [md5_file(p1), md5_file(p2)] == case["fixture_md5"], "fixture generator drifted: " + case["fixture"]
    outdir = os.path.join(str(tmp_path), "out")
    m = {"{1}": p1, "{2}": p2, "{U}": p1, "{D}": outdir, "{ART}": ARTIFACT_FASTA}
    argv = [m.get(a, a.replace("{D}", outdir)) for a in case["args"]]
    err = io.StringIO()
    rc = driver.run(argv, engine_factory=engine_factory, err=err, **kw)
    bad = []
    if not exit_code_matches(rc, case["exit_code"]):
        bad.append("exit code %d != %d (%s)" % (rc, case["exit_code"], err.getvalue()[-300:]))
    have = set(os.listdir(outdir)) if os.path.isdir(outdir) else set()
    for fn, text in case["text"].items():
        if fn not in have:
            bad.append("missing " + fn)
            continue
        with open(os.path.join(outdir, fn), errors="replace") as f:
            got = f.read()
        if got != text:
            gl, tl = got.splitlines(), text.splitlines()
            k = next((i for i in range(min(len(gl), len(tl))) if gl[i] != tl[i]), min(len(gl), len(tl)))
            bad.append("%s differs at line %d: got %r want %r" % (fn, k + 1, gl[k:k + 1], tl[k:k + 1]))
    bad += _compare_big_texts(case, outdir, have)
    for fn, meta in case["fastq"].items():
        if fn not in have:
            bad.append("missing " + fn)
            continue
        with open(os.path.join(outdir, fn), "rb") as f:
            data = f.read()
        if hashlib.md5(data).hexdigest() != meta["md5"]:
            bad.append("%s: md5 mismatch (records %d vs %d, bytes %d vs %d)" % (
                fn, data.count(b"\n") // 4, meta["records"], len(data), meta["bytes"]))
    extra = {f for f in have if not f.endswith(".pdf")} - set(case["text"]) - set(case["fastq"]) - set(case.get("text_md5", {}))
    if extra:
        bad.append("unexpected files: %s" % sorted(extra))
    return bad


def compare_outputs(case, outdir, rc, err_text):
    bad = []
    if not exit_code_matches(rc, case["exit_code"]):
        bad.append("exit code %d != %d (%s)" % (rc, case["exit_code"], err_text[-300:]))
    have = set(os.listdir(outdir)) if os.path.isdir(outdir) else set()
    for fn, text in case["text"].items():
        if fn not in have:
            bad.append("missing " + fn)
            continue
        with open(os.path.join(outdir, fn), errors="replace") as f:
            got = f.read()
        if got != text:
            gl, tl = got.splitlines(), text.splitlines()
            k = next((i for i in range(min(len(gl), len(tl))) if gl[i] != tl[i]), min(len(gl), len(tl)))
            bad.append("%s differs at line %d: got %r want %r" % (fn, k + 1, gl[k:k + 1], tl[k:k + 1]))
    bad += _compare_big_texts(case, outdir, have)
    for fn, meta in case["fastq"].items():
        if fn not in have:
            bad.append("missing " + fn)
            continue
        with open(os.path.join(outdir, fn), "rb") as f:
            data = f.read()
        if hashlib.md5(data).hexdigest() != meta["md5"]:
            bad.append("%s: md5 mismatch (records %d vs %d)" % (fn, data.count(b"\n") // 4, meta["records"]))
    return bad


def run_case_binary(case, cache, tmp_path, binary, extra_args=(), env=None):
    """Runs a FaQCs-compatible executable (the reference driver linked against integration/trim_shim.cpp) on the
    case's command line and compares every output file with the reference's own outputs."""
    import subprocess

    p1, p2 = fixture_paths(case["fixture"], cache)
    outdir = os.path.join(str(tmp_path), "out")
    m = {"{1}": p1, "{2}": p2, "{U}": p1, "{D}": outdir, "{ART}": ARTIFACT_FASTA}
    argv = [m.get(a, a.replace("{D}", outdir)) for a in case["args"]] + list(extra_args)
    for _ in range(6):  # the reference driver can die of SIGPIPE feeding the absent R (see make_golden.py)
        import shutil

        shutil.rmtree(outdir, ignore_errors=True)
        proc = subprocess.run([binary] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, **env) if env else None)
        if proc.returncode != -13:
            break
    return compare_outputs(case, outdir, proc.returncode, proc.stderr.decode(errors="replace"))
