"""CPU-side checks of bench.py's multi-GPU launch contract and of the HIP-runtime binding of libfaqcs_mi.so (no GPU needed:
every path tested here stops before the first GPU call)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = dict(os.environ)
    for k in drop:
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def test_gpus_flag_is_authoritative():
    """--gpus N > visible GPUs is an error (the round-1 bench silently measured one GPU), and so is a WORLD_SIZE that
    disagrees with --gpus."""
    import torch

    if torch.cuda.device_count() < 2:
        r = _run(["--gpus", "2", "--pairs", "1e5"])
        assert r.returncode != 0 and b"GPU(s) are visible" in r.stderr
    r = _run(["--gpus", "1", "--pairs", "1e5"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"disagrees with WORLD_SIZE" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0


def test_library_binds_to_torch_hip_runtime():
    """Loaded after torch, libfaqcs_mi.so resolves libamdhip64.so.7 to the runtime torch already mapped (same SONAME):
    ONE HIP runtime in the process, so device pointers, streams and RCCL buffers belong to the same runtime."""
    code = (
        "import torch, ctypes, sys\n"
        "sys.path.insert(0, %r)\n"
        "from faqcs_amd import _capi\n"
        "_capi.load_library()\n"
        "libs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
        "print(len(libs), libs)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode().startswith("1 "), r.stdout.decode()


def test_a_rank_that_dies_before_the_rendezvous_fails_the_job_fast():
    """bench.py --gpus 2 starts the ranks itself: rank 1 exits before init_process_group, rank 0 is left waiting in the rendezvous.
    The launcher polls every rank, terminates the survivors and exits with the dead rank's code within seconds (it used to block
    on rank 0 until the store timeout), keeping every rank's stderr under gpurun_out/."""
    import time

    t0 = time.time()
    r = _run(["--gpus", "2", "--pairs", "1e5", "--steps", "1", "--warmup", "0"],
             {"FAQCS_BENCH_SHARE_GPU": "1", "FAQCS_BENCH_DIE_RANK": "1", "FAQCS_BENCH_RDZV_TIMEOUT": "600"})
    dt = time.time() - t0
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-1500:])
    assert dt < 30, dt
    assert os.path.exists(os.path.join(ROOT, "gpurun_out", "rank1.err"))


def test_full_size_checks_accept_a_consistent_block_and_reject_a_broken_one():
    """bench.verify_full_size(): the size-independent properties bench.py checks on the counter block of its full-size jobs (sums of the
    matrices against FilterStat, rows against the length histogram, one composition count per read and kind, reads credited per adapter
    against adapter_stats).  Here on the CPU checker's block and results for ragged reads with adapters: it must pass, and a block with
    ONE counter off by one must not."""
    import numpy as np
    import pytest
    import torch

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import bench
    import make_fixtures
    from oracle_engine import OracleEngine

    from faqcs_amd import _capi as capi
    from faqcs_amd import driver
    from faqcs_amd.options import parse_args

    opt = parse_args(["-u", "x", "-d", "y", "--adapter", "--polyA", "--min_L", "30"])
    rng = np.random.Generator(np.random.PCG64(99))
    reads = []
    for _ in range(1500):
        s, q = make_fixtures._adv_read(rng, 150)
        reads.append((b"@r", s.tobytes(), q.tobytes()))
    eng = OracleEngine(opt, 160, 33)
    seq, qual, offset, seg = driver.pack_segments([reads[:700], reads[700:]])
    res = eng.process(seq, qual, offset, seg)
    blk = eng.counters()
    n_ad = eng.holder.n_adapters if hasattr(eng, "holder") else len(opt.adapter)
    lay = capi.python_layout(160, n_ad)
    fs = blk[lay["filter_stats"][0]:lay["filter_stats"][0] + 32]
    r = torch.from_numpy(res.view(np.uint16).reshape(-1, 4).astype(np.int32).astype(np.int16))
    L = 0  # (ragged: the function takes the lengths from the block itself)
    bench.verify_full_size(blk, lay, fs, [r[:800], r[800:]], len(reads), L, "adapter", n_ad)
    for name in ("post_qual", "pre_comp", "adapter_stats", "post_len_hist"):
        bad = blk.copy()
        k = lay[name][0] + int(np.nonzero(blk[lay[name][0]:lay[name][0] + lay[name][1]])[0][0])
        bad[k] += 1
        with pytest.raises(SystemExit):
            bench.verify_full_size(bad, lay, fs, [r], len(reads), L, "adapter", n_ad)
