"""Multi-process (world_size = 2, gloo, CPU) test of the only collective on the path: shards of reads are
processed independently and the additive u64 counter block is all-reduced (faqcs_amd/parallel.py).  The
per-rank engine here is the CPU checker standing in for the HIP engine -- the sharding / reduction logic under
test is identical (everything except k-mers is per-read results + integer sums)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, args, n_reads, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist

    import make_fixtures
    from oracle_engine import OracleEngine

    from faqcs_amd import driver, parallel
    from faqcs_amd.options import parse_args

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    opt = parse_args(["-u", "x", "-d", "y"] + args)
    rng = np.random.Generator(np.random.PCG64(1234))
    reads = []
    for _ in range(n_reads):
        s, q = make_fixtures._adv_read(rng, 150)
        reads.append((b"@r", s.tobytes(), q.tobytes()))
    # shard on 8-read boundaries so the adapter pre-pass groups (trim.cpp:977-1071) stay inside one shard
    groups = n_reads // 8
    lo, hi = parallel.shard_bounds(groups, rank, world)
    mine = reads[lo * 8:(hi * 8 if rank < world - 1 else n_reads)]
    eng = OracleEngine(opt, 256, 33)
    seq, qual, offset, seg = driver.pack_segments([mine])
    res = eng.process(seq, qual, offset, seg)
    total = parallel.allreduce_counters_host(eng.counters())
    if rank == 0:
        ref = OracleEngine(opt, 256, 33)
        s2, q2, o2, g2 = driver.pack_segments([reads])
        res_all = ref.process(s2, q2, o2, g2)
        ok = bool((total == ref.counters()).all()) and bool((res == res_all[: len(mine)]).all())
        with open(out, "w") as f:
            f.write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("args", [[], ["--adapter", "--polyA"]], ids=["plain", "adapter"])
def test_two_rank_allreduce_equals_single_process(args, tmp_path):
    import torch.multiprocessing as mp

    port = _free_port()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, port, args, 1203, out), nprocs=2, join=True)
    assert open(out).read() == "ok"


def test_shard_bounds_cover_everything():
    from faqcs_amd import parallel

    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
