"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol include/faqcs_mi.h
declares, agrees with the python / oracle statement of the counter layout, fails loudly without a device,
and its host helpers (apply_edits, auto-detect, counter_rows) match the driver's numpy implementations."""
import ctypes as C

import numpy as np
import pytest

from faqcs_amd import _capi as capi
from faqcs_amd import driver
from faqcs_amd.options import parse_args


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    return capi.load_library()


def test_exports_every_declared_symbol(lib):
    declared = capi.declared_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.faqcs_abi_version() == capi.ABI_VERSION


@pytest.mark.parametrize("R,na", [(150, 0), (256, 10), (1024, 12)])
def test_layout_agrees(lib, R, na):
    lay = capi.Layout()
    assert lib.faqcs_counters_layout(R, na, C.byref(lay)) == 0
    py = capi.python_layout(R, na)
    for name, (off, _) in py.items() if False else [(k, v) for k, v in py.items() if k != "total"]:
        assert getattr(lay, name) == off, name
    assert lay.total == py["total"]
    from oracle_engine import OracleEngine

    opt = parse_args(["-u", "x", "-d", "y"] + (["--adapter", "--polyA"] if na == 10 else []))
    if na in (0, 10):
        eng = OracleEngine(opt, R, 33)
        assert eng.n_counters == lay.total


def test_create_fails_loudly_without_gpu(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    opt = parse_args(["-u", "x", "-d", "y"])
    h = capi.ParamsHolder(opt, 256, 33)
    ctx = C.c_void_p()
    rc = lib.faqcs_create(C.byref(h.p), -1, C.byref(ctx))
    assert rc == capi.E_NODEVICE
    assert b"no CPU fallback" in lib.faqcs_last_error()
    from faqcs_amd.engine import FaqcsError, HipEngine

    with pytest.raises(FaqcsError):
        HipEngine(opt, 256, 33)


def test_counter_rows(lib):
    m = np.zeros((10, 42), dtype=np.uint64)
    assert lib.faqcs_counter_rows(m.ctypes.data, 10, 42) == 0
    m[6, 3] = 1
    assert lib.faqcs_counter_rows(m.ctypes.data, 10, 42) == 7


def test_auto_detect(lib):
    q = np.frombuffer(b"IIIIJJJJ" + b"IIh", dtype=np.uint8)
    off = np.array([0, 8, 11], dtype=np.uint32)
    assert lib.faqcs_auto_detect_quality_offset(q.ctypes.data, off.ctypes.data, 2) == 64
    q = np.frombuffer(b"IIIIJJJ#", dtype=np.uint8)
    assert lib.faqcs_auto_detect_quality_offset(q.ctypes.data, off.ctypes.data, 1) == 33
    q = np.frombuffer(b"IIIIJJJJ", dtype=np.uint8)
    assert lib.faqcs_auto_detect_quality_offset(q.ctypes.data, off.ctypes.data, 1) == 0
    assert driver.auto_detect_quality_offset([b"IIIIJJJJ", b"IIh"]) == 64


@pytest.mark.parametrize("args", [[], ["--replace_to_N_q", "15"], ["--out_ascii", "64"], ["--ascii", "64", "--out_ascii", "33"]])
def test_apply_edits_matches_driver(lib, args):
    rng = np.random.default_rng(5)
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    in_off = 64 if "--ascii" in args else 33
    h = capi.ParamsHolder(opt, 256, in_off)
    reads = []
    for _ in range(200):
        L = int(rng.integers(1, 60))
        s = np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, L)].copy()
        if rng.random() < 0.3:
            s[: rng.integers(0, 4)] = ord("N")
        if rng.random() < 0.3:
            s[L - int(rng.integers(0, 4)):] = ord("N")
        q = (rng.integers(0, 42, L) + in_off).astype(np.uint8)
        reads.append((b"@r", s.tobytes(), q.tobytes()))
    seq, qual, offset, _ = driver.pack_segments([reads])
    es, eq = driver.edited_arenas(opt, in_off, seq, qual, offset)
    res = np.zeros(1, dtype=capi.RESULT_DTYPE)
    for i in range(len(reads)):
        a, b = int(offset[i]), int(offset[i + 1])
        L = b - a
        st = int(rng.integers(0, L))
        ln = int(rng.integers(0, L - st + 1))
        res["start"], res["len"] = st, ln
        os_, oq = np.zeros(ln + 1, np.uint8), np.zeros(ln + 1, np.uint8)
        rc = lib.faqcs_apply_edits(C.byref(h.p), seq[a:].ctypes.data, qual[a:].ctypes.data, L, res.ctypes.data,
                                   os_.ctypes.data, oq.ctypes.data)
        assert rc == 0
        assert (os_[:ln] == es[a + st:a + st + ln]).all()
        assert (oq[:ln] == eq[a + st:a + st + ln]).all()
