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
