"""Pins the CPU oracle (oracle/faqcs_oracle.c) and the host driver against outputs of the REAL reference
(tests/golden/cases/*.json, produced by tests/golden/make_golden.py from oracle/_ref/FaQCs_ref)."""
import pytest
from golden_util import case_max_read_length, case_names, load_case, run_case
from oracle_engine import oracle_factory


@pytest.mark.parametrize("name", case_names())
def test_oracle_matches_reference(name, fixture_cache, tmp_path):
    case = load_case(name)
    bad = run_case(case, fixture_cache, tmp_path, oracle_factory, max_read_length=case_max_read_length(case))
    assert not bad, "\n".join(bad)
