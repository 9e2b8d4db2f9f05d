"""The arithmetic of the super-k-mer kernels (faqcs_amd/csrc/faqcs_skm.h, shared by host and device) against update_kmer()'s k-mers
(trim.cpp:887-931), on the CPU: tests/skm_model.cpp emulates a wave's extraction rounds lane by lane."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("skm") / "skm_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unknown-pragmas", "-o", exe, os.path.join(ROOT, "tests", "skm_model.cpp")])
    return exe


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_items_expand_to_the_reference_kmers_and_partitions_are_consistent(model, seed):
    out = subprocess.run([model, "1500", str(seed)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1].startswith("ok:")
