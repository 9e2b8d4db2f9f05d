"""Multi-process (world_size = 2, 3, 4 and 8; gloo, CPU) tests of the only collective on the path: shards of reads are
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


@pytest.mark.parametrize("world,args", [(3, []), (4, ["--adapter", "--polyA"]), (8, [])], ids=["w3_plain", "w4_adapter", "w8_plain"])
def test_more_ranks_allreduce_equals_single_process(world, args, tmp_path):
    """The same job on 3, 4 and 8 ranks (gloo, CPU): shard counts that do not divide the 150 adapter groups of the input evenly
    (1 203 reads: the last rank also takes the 3 reads behind the last whole group), the 8-rank shape of BASELINE configs[3]."""
    import torch.multiprocessing as mp

    port = _free_port()
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, port, args, 1203, out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_schedule_slices_for_every_rank_count():
    """bench.py --config kmer --gpus N hands rank r the epochs of ITS shard of the global sequence of trim() calls (rank-major
    order): for 1 ... 8 ranks the slices tile the schedule, and the epoch of a segment never depends on how many ranks there are."""
    from faqcs_amd import parallel

    sizes_one = [32768] * 13 + [1234]
    for world in range(1, 9):
        sizes = sizes_one * world
        ep, pts = parallel.rarefaction_schedule(sizes, 100000, 40)
        per = len(sizes) // world
        got = []
        for r in range(world):
            got.extend(ep[r * per:(r + 1) * per])
        assert got == ep and len(ep) == len(sizes)
        total = 0
        for e, sz in zip(ep, sizes):
            assert e == parallel.EPOCH_NONE or e == sum(1 for p in pts if p <= total)  # the points taken before this call
            total += sz


def test_shard_bounds_cover_everything():
    from faqcs_amd import parallel

    for n in (0, 1, 7, 64, 1001):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


@pytest.mark.parametrize("split,subset,seg", [(300, 4, 333), (700, 10, 333), (5000, 10, 100), (100, 3, 64), (64, 50, 64)])
def test_rarefaction_schedule_matches_the_oracle(split, subset, seg):
    """parallel.rarefaction_schedule (the host rule the multi-GPU k-mer path runs on GLOBAL read counts) against the
    points of the single-process oracle (trim.cpp:157-185), and the epoch identity the exchange relies on:
    distinct(point i) = #{keys whose first epoch <= i}, total(point i) = #{occurrences with epoch <= i}."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_fixtures
    from oracle_engine import OracleEngine

    from faqcs_amd import driver, parallel
    from faqcs_amd.options import parse_args

    opt = parse_args(["-u", "x", "-d", "y", "--kmer_rarefaction", "--split_size", str(split), "--subset", str(subset), "--qc_only", "-m", "9"])
    rng = np.random.Generator(np.random.PCG64(5))
    reads = []
    for _ in range(1500):
        s, q = make_fixtures._adv_read(rng, 60)
        reads.append((b"@r", s.tobytes(), q.tobytes()))
    segs = [reads[i:i + seg] for i in range(0, len(reads), seg)]
    epochs, points = parallel.rarefaction_schedule([len(x) for x in segs], opt.split_size, opt.num_subsample)
    ora = OracleEngine(opt, 256, 33)
    ora.process(*driver.pack_segments(segs))
    got = ora.kmer_points()
    assert [int(p["num_seq"]) for p in got] == points
    # epoch identity with plain python sets (k = 9, --qc_only: the raw reads are counted)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    first, total_by_epoch = {}, {}
    for e, sg in zip(epochs, segs):
        if e == parallel.EPOCH_NONE:
            continue
        for _d, sq, _q in sg:
            u = sq.upper()
            for i in range(len(u) - 8):
                w = u[i:i + 9]
                if w.strip(b"ACGT"):
                    continue
                key = min(w, w.translate(comp)[::-1])
                first[key] = min(first.get(key, e), e)
                total_by_epoch[e] = total_by_epoch.get(e, 0) + 1
    for i, p in enumerate(got):
        assert int(p["distinct_kmer"]) == sum(1 for v in first.values() if v <= i)
        assert int(p["total_kmer"]) == sum(v for k, v in total_by_epoch.items() if k <= i)
