#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/cases/ by running the REAL reference binary
(oracle/_ref/FaQCs_ref, compiled from /root/reference by `make -C oracle ref`) over the fixtures.

TEST INFRASTRUCTURE; runs only in the build container (the GPU box has no /root/reference and only
consumes the committed JSON).  Each case file holds the command line, the md5 of every input fixture,
and the reference's observable outputs: exit code, <prefix>.stats.txt, every --debug table, and
(md5, record count, byte count) of each trimmed FASTQ.

    python tests/golden/make_golden.py [case-name-substring ...]
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_fixtures  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "FaQCs_ref")
CACHE = os.path.join(HERE, "_cache")
CASES_DIR = os.path.join(HERE, "cases")
BIG_TEXT = 128 * 1024  # a text output above this size is stored as md5 + size + line count

# (case name, fixture, args).  {1} {2} {U} {D} {ART} are substituted.  -t 1 everywhere: SURVEY.md H1.
P = ["-1", "{1}", "-2", "{2}", "-d", "{D}", "-t", "1", "--debug"]
CASES = [
    ("adv_default", "adv", P),
    ("adv_ascii33_t4", "adv", ["-1", "{1}", "-2", "{2}", "-d", "{D}", "-t", "4", "--debug", "--ascii", "33"]),
    ("adv_bwa", "adv", P + ["--mode", "BWA"]),
    ("adv_hard_q10", "adv", P + ["--mode", "HARD", "-q", "10"]),
    ("adv_hard_q10_5off", "adv", P + ["--mode", "HARD", "-q", "10", "--5trim_off"]),
    ("adv_5trim_off", "adv", P + ["--5trim_off"]),
    ("adv_q20_minL30", "adv", P + ["-q", "20", "--min_L", "30"]),
    ("adv_5end3_3end5", "adv", P + ["--5end", "3", "--3end", "5"]),
    ("adv_5end60_3end100", "adv", P + ["--5end", "60", "--3end", "100", "--min_L", "10"]),
    ("adv_avgq25", "adv", P + ["--avg_q", "25"]),
    ("adv_n1", "adv", P + ["-n", "1"]),
    ("adv_lc05", "adv", P + ["--lc", "0.5"]),
    ("adv_replaceN15", "adv", P + ["--replace_to_N_q", "15"]),
    ("adv_out64", "adv", P + ["--out_ascii", "64"]),
    ("adv_qc_only", "adv", P + ["--qc_only"]),
    ("adv_trim_only", "adv", ["-1", "{1}", "-2", "{2}", "-d", "{D}", "-t", "1", "--trim_only"]),
    ("adv_discard", "adv", P + ["--discard"]),
    ("adv_adapter", "adv", P + ["--adapter"]),
    ("adv_adapter_polyA", "adv", P + ["--adapter", "--polyA"]),
    ("adv_adapter_polyA_rate03", "adv", P + ["--adapter", "--polyA", "--rate", "0.3"]),
    ("adv_adapter_qc_only", "adv", P + ["--adapter", "--polyA", "--qc_only"]),
    ("adv_adapter_5end", "adv", P + ["--adapter", "--5end", "4", "--3end", "2", "--min_L", "20"]),
    ("adv_artifact", "adv", P + ["--artifactFile", "{ART}"]),
    ("adv_unpaired_only", "adv", ["-u", "{1}", "-d", "{D}", "-t", "1", "--debug", "--discard"]),
    ("adv_paired_plus_unpaired", "adv", P + ["-u", "{2}"]),
    ("adv_kmer", "adv", P + ["--kmer_rarefaction", "--split_size", "500"]),
    ("adv_kmer_qc_only_subset1", "adv", P + ["--kmer_rarefaction", "--split_size", "700", "--qc_only", "--subset", "1"]),
    ("nextseq_default", "nextseq", P),
    ("nextseq_phix", "nextseq", P + ["--phiX"]),
    ("head150_default", "head150", P + ["--min_L", "50", "-q", "5"]),
    ("head150_adapter_polyA", "head150", P + ["--adapter", "--polyA"]),
    ("head250_default", "head250", P),
    # BASELINE configs[4]'s shape: 2x250, k-mer rarefaction with --subset 200 (no -m: the reference's getopt string lacks it)
    ("head250_kmer", "head250", P + ["--kmer_rarefaction", "--subset", "200", "--split_size", "300"]),
    ("long300_default", "long300", P),
    ("long300_adapter_polyA", "long300", P + ["--adapter", "--polyA"]),
    ("long300_kmer_q20", "long300", P + ["--kmer_rarefaction", "--split_size", "300", "-q", "20", "--replace_to_N_q", "12"]),
    ("long1000_default", "long1000", P),
    ("long1000_bwa_avgq", "long1000", P + ["--mode", "BWA", "--avg_q", "20", "-n", "3"]),
    ("long1000_hard_lc", "long1000", P + ["--mode", "HARD", "-q", "12", "--lc", "0.6", "--5end", "7", "--3end", "9"]),
    ("long1000_adapter_polyA", "long1000", P + ["--adapter", "--polyA", "--discard"]),
    # reads of up to 32 767 bases (trim_long, adapter_overlap<1, 32768>); their position tables are stored as md5 (see collect_outputs)
    ("long8k_default", "long8k", P),
    ("long8k_adapter_polyA", "long8k", P + ["--adapter", "--polyA", "--discard"]),
    ("long8k_hard_lc", "long8k", P + ["--mode", "HARD", "-q", "12", "--lc", "0.6", "--5end", "7", "--3end", "9"]),
    ("long8k_bwa_avgq_n3", "long8k", P + ["--mode", "BWA", "--avg_q", "20", "-n", "3", "--out_ascii", "64"]),
    ("long8k_kmer_replaceN", "long8k", P + ["--kmer_rarefaction", "--split_size", "10", "-q", "20", "--replace_to_N_q", "12", "--min_L", "30"]),
    ("long8k_qc_only_5off", "long8k", P + ["--qc_only", "--5trim_off"]),
    # --stats is honoured (options.cpp:291-293,739-740): the statistics go to the named file instead of <prefix>.stats.txt
    ("adv_prefix_stats", "adv", P + ["--prefix", "SAMPLE7", "--stats", "{D}/custom.stats.txt"]),
    ("adv_single_dash_long_options", "adv", ["-1", "{1}", "-2", "{2}", "-d", "{D}", "-t", "1", "-debug", "-min_L", "40", "-q", "12", "-lc", "0.7", "-discard", "-substitute"]),
    ("adv64_autodetect", "adv64", P),
    ("adv64_ascii64_out33", "adv64", P + ["--ascii", "64", "--out_ascii", "33"]),
    ("adv_t8_no_adapter", "adv", ["-1", "{1}", "-2", "{2}", "-d", "{D}", "-t", "8", "--debug", "-n", "3"]),
    ("err_quality_above_41", "errq", P),
    ("err_unknown_base_adapter", "errbase", P + ["--adapter"]),
    ("err_unknown_base_no_adapter", "errbase", P),
    ("err_mate_id_mismatch", "errid", P),
    ("err_unequal_record_counts", "errcount", P),
    ("example_fixed_point", "example", P),
    ("advbig_default", "advbig", P),
    ("advbig_adapter_polyA", "advbig", P + ["--adapter", "--polyA"]),
    ("advbig_kmer", "advbig", P + ["--kmer_rarefaction", "--split_size", "7000"]),
]

ARTIFACT_FASTA = os.path.join(HERE, "artifact.fa")


def md5_file(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def fixture_paths(name):
    if name == "example":
        return os.path.join(HERE, "example_1.fastq.gz"), os.path.join(HERE, "example_2.fastq.gz")
    return make_fixtures.materialise(name, CACHE)


def substitute(args, p1, p2, outdir):
    m = {"{1}": p1, "{2}": p2, "{U}": p1, "{D}": outdir, "{ART}": ARTIFACT_FASTA}
    return [m.get(a, a.replace("{D}", outdir)) for a in args]


def collect_outputs(outdir):
    out = {"text": {}, "fastq": {}}
    for fn in sorted(os.listdir(outdir)):
        path = os.path.join(outdir, fn)
        if fn.endswith(".fastq"):
            with open(path, "rb") as f:
                data = f.read()
            out["fastq"][fn] = {"md5": hashlib.md5(data).hexdigest(), "bytes": len(data), "records": data.count(b"\n") // 4}
        elif fn.endswith(".pdf"):
            continue
        else:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) > BIG_TEXT: # (the per-position tables of reads with tens of thousands of bases: megabytes of text)
                out.setdefault("text_md5", {})[fn] = {"md5": hashlib.md5(data).hexdigest(), "bytes": len(data), "lines": data.count(b"\n")}
            else:
                out["text"][fn] = data.decode(errors="replace")
    return out


def main():
    if not os.path.exists(REF_BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(CASES_DIR, exist_ok=True)
    want = sys.argv[1:]
    for name, fixture, args in CASES:
        if want and not any(w in name for w in want):
            continue
        p1, p2 = fixture_paths(fixture)
        tmp = tempfile.mkdtemp(prefix="faqcs_golden_")
        outdir = os.path.join(tmp, "out")
        argv = substitute(args, p1, p2, outdir)
        for _ in range(8):
            # R is absent in this image: the reference popen()s it (plot.cpp:507) and, depending on timing,
            # dies of SIGPIPE while feeding it the script -- after every output file is complete.  Retry
            # until the race lets it exit normally so the recorded exit code is the real one.
            shutil.rmtree(outdir, ignore_errors=True)
            proc = subprocess.run([REF_BIN] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if proc.returncode != -13:
                break
        case = {
            "name": name, "fixture": fixture, "args": args, "exit_code": proc.returncode,
            "fixture_md5": [md5_file(p1), md5_file(p2)],
            "stderr": [l for l in proc.stderr.decode(errors="replace").splitlines() if "R: not found" not in l],
        }
        case.update(collect_outputs(outdir) if os.path.isdir(outdir) else {"text": {}, "fastq": {}})
        with open(os.path.join(CASES_DIR, name + ".json"), "w") as f:
            json.dump(case, f, indent=1, sort_keys=True)
        print("%-28s exit=%d files=%d" % (name, proc.returncode, len(case["text"]) + len(case["fastq"])))
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
