"""The slot's validity rule of trim_lds, checked on the GPU: the address of a Q-B add is (quality byte) x (row
stride) + lane base even when its increment is zero, and an LDS add of zero is still a read-modify-write -- an address outside the
position x quality matrix can undo another wave's LDS-DMA write (the race of round 4, profiles/r4c/restage_race.txt).  build() compiles
a second library with -DFAQCS_LDS_DIAG_CHECK_QB_ADDR (tests/_diag/libfaqcs_mi_qbchk.so) that counts such adds; tools/qb_rule_probe.py
drives batches of every lane geometry through it (adversarial, equal-length with most reads taken back, ragged).  Not one may occur."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "tests", "_diag", "libfaqcs_mi_qbchk.so")


def test_no_quality_add_leaves_the_matrix():
    assert os.path.exists(DIAG), "tests/_diag/libfaqcs_mi_qbchk.so is missing: run `python __graft_entry__.py` (build())"
    env = dict(os.environ, FAQCS_MI_LIB=DIAG)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "qb_rule_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("adds outside the quality matrix: 0 ;"), last
