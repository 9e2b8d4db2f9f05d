#!/usr/bin/env python3
"""bench.py -- throughput of the FaQCs per-read hot path on MI355X.

A *step* is one pass of the trim kernel (trim_lds for the headline shape; + adapter_overlap for --config adapter) over the whole
synthetic data set resident in HBM: BASELINE.json configs[1] (100 M pairs of 2x150 bp, BWA_plus -q 5
--min_L 50) per GPU, generated on the device by faqcs_synth_fill.  Inputs are in HBM
when the timed region starts; every step ends with the job's one collective, the all-reduce of the counter
block (world_size > 1).  value = reads processed by all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     algorithmic bytes (2L + 4 + 8 per read) / average kernel time measured with HIP
               events on the library's compute stream; peak = 8 TB/s HBM3E
  cpu_baseline the real reference binary (oracle/_ref/FaQCs_ref -t <cores>, kind "reference") or the plain-C
               port (kind "port") timed on a bounded sample of the same workload on this host.
  e2e          (N = 1, plain / adapter) faqcs_mi from FASTQ files in /dev/shm to trimmed FASTQ + QC.stats.txt, reads/s.

--gpus N without WORLD_SIZE in the environment: this process starts N rank processes (before it touches the GPU) and relays
rank 0's line; under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  --config kmer --gpus N runs the
owner-partitioned k-mer exchange of DESIGN.md section 6.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=float, default=float(os.environ.get("FAQCS_BENCH_PAIRS", 0)),
                    help="pairs per GPU resident in HBM (default: BASELINE's sizes -- 100 M for plain / adapter on one GPU (configs[1], [2]), "
                         "125 M per GPU for plain on 8 GPUs (configs[3]: 1 B pairs), 25 M per GPU for kmer (configs[4]: 200 M pairs on 8 GPUs))")
    ap.add_argument("--read-len", type=int, default=None, help="default 150 (250 for --config kmer)")
    ap.add_argument("--config", choices=["plain", "adapter", "kmer", "replaceN"], default="plain",
                    help="plain = BASELINE configs[1] (the headline), adapter = configs[2], kmer = configs[4]'s shape on one GPU")
    ap.add_argument("--batch-reads", type=int, default=1 << 25, help="reads per submission, clamped to what a < 4 GiB arena holds (u32 offsets): "
                    "28.6 M reads of 150 bases; fewer, larger launches = fewer seams between launches (2^24: -2.8 %% on the default line)")
    ap.add_argument("--at-frac", type=float, default=None, help="A+T fraction of the synthetic bases (default: uniform ACGT); 0.9 makes most "
                    "reads dinucleotide candidates of the low-complexity filter (an AT-rich genome)")
    ap.add_argument("--kmer-table-log2", type=int, default=31, help="log2 of the k-mer table's slots per GPU (--config kmer; 2^31 slots = 34 GB: the "
                    "tests that put eight ranks on one GPU pass 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run-memory", action="store_true", help="--config kmer: print every rank's HBM budget (reads, table, group buffers: "
                    "faqcs_kmer_memory_plan) and stop before anything is allocated; exit 1 when a rank's share of its device cannot hold it")
    ap.add_argument("--no-other-configs", action="store_true", help="the default single-GPU run (config plain, 2x150) also runs BASELINE's adapter and k-mer "
                    "configurations, two steps each, and reports them under \"configs\"; this switches that off")
    ap.add_argument("--e2e-pairs", type=float, default=float(os.environ.get("FAQCS_BENCH_E2E_PAIRS", -1)),
                    help="pairs of the same workload written as FASTQ to /dev/shm for the end-to-end (files in, files out) run of faqcs_mi; 0 = skip; "
                         "default: 16 M pairs when /dev/shm has 64 GB free (20 GB of files; the fixed 0.4 s of HIP start-up weighs less), else 8 M")
    return ap.parse_args()


def effective_cpus():
    """CPUs the process may use: the cgroup quota when there is one (the GPU boxes show 256 hardware threads and allow 16), else os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(opt_args, hs, hq, L, n_sample):
    """Times the reference (or the oracle port) on a bounded sample: returns the cpu_baseline object."""
    cores = os.cpu_count() or 1
    ref = os.path.join(ROOT, "oracle", "_ref", "FaQCs_ref")
    port = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_engine import OracleEngine

        from faqcs_amd.options import parse_args

        opt = parse_args(["-1", "a", "-2", "b", "-d", "x", "--ascii", "33"] + opt_args)
        m = min(n_sample, 200000)
        eng = OracleEngine(opt, 256, 33)
        off = (np.arange(m + 1, dtype=np.uint64) * L).astype(np.uint32)
        t0 = time.perf_counter()
        eng.process(hs, hq, off, np.array([0, m], dtype=np.uint32))
        port = m / (time.perf_counter() - t0) / 1e6
    except Exception as e:  # the checker is optional for the bench
        print("cpu_baseline: oracle port unavailable (%s)" % e, file=sys.stderr)
    if os.path.exists(ref):
        try:
            tmp = tempfile.mkdtemp(prefix="faqcs_bench_")
            half = n_sample // 2
            for mate, lo in ((1, 0), (2, half)):
                with open(os.path.join(tmp, "r%d.fq" % mate), "wb") as f:
                    chunk = []
                    for i in range(half):
                        a = (lo + i) * L
                        chunk.append(b"@SYN:%d/%d\n" % (i, mate) + hs[a:a + L].tobytes() + b"\n+\n" + hq[a:a + L].tobytes() + b"\n")
                    f.write(b"".join(chunk))
            # The reference's -t scaling is not monotone (its per-call `omp critical` merge of ~135 k counters per thread grows with the
            # team, FaQCs trim.cpp:120-154): every thread count of {1, 8, 16, all cores} is timed on the same sample and the BEST is reported
            tried = {}
            for t in sorted({1, min(8, cores), min(16, cores), cores}):
                cmd = [ref, "-1", os.path.join(tmp, "r1.fq"), "-2", os.path.join(tmp, "r2.fq"), "-d", os.path.join(tmp, "out%d" % t),
                       "-t", str(t), "--ascii", "33", "--trim_only"] + opt_args
                t0 = time.perf_counter()
                subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=600)
                tried[t] = round(2 * half / (time.perf_counter() - t0) / 1e6, 4)
            subprocess.run(["rm", "-rf", tmp])
            best = max(tried, key=lambda t: tried[t])
            return {"value": tried[best], "unit": "M reads/s", "cores": best, "kind": "reference",
                    "threads_tried": {str(t): v for t, v in tried.items()}, "host_cores": cores, "host_cpus_allowed_by_cgroup": effective_cpus(),
                    "sample": "%d pairs of the same synthetic 2x%d workload as uncompressed FASTQ, whole-process wall clock of "
                              "FaQCs v2.10 --trim_only (parse+trim+write; the reference cannot separate them) at -t 1 / 8 / 16 / all cores; "
                              "value = the best of them (-t %d)" % (half, L, best),
                    "port_value": None if port is None else round(port, 4)}
        except Exception as e:
            print("cpu_baseline: reference run failed (%s)" % e, file=sys.stderr)
    if port is not None:
        return {"value": round(port, 4), "unit": "M reads/s", "cores": 1, "kind": "port",
                "sample": "%d reads of the same workload through oracle/faqcs_oracle.c (single thread, compute only)" % min(n_sample, 200000)}
    return None


def e2e_run(opt_args, hs, hq, L, n_pairs):
    """End to end through files: the first n_pairs pairs of the workload as two uncompressed FASTQ files in /dev/shm ->
    faqcs_amd/faqcs_mi (FaQCs command line on the HIP library) -> trimmed FASTQ + QC.stats.txt in /dev/shm.  Whole-process wall
    clock, HIP start-up included.  hs / hq: host copies of 2 * n_pairs reads (mate 1 = the first half)."""
    cli = os.path.join(ROOT, "faqcs_amd", "faqcs_mi")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    tmp = tempfile.mkdtemp(prefix="faqcs_e2e_", dir=base)
    try:
        n = int(n_pairs)
        paths = []
        for mate in (1, 2):
            lo = (mate - 1) * n
            idw = 9
            head = np.frombuffer(b"@SYN:", np.uint8)
            tail = np.frombuffer(b"/%d\n" % mate, np.uint8)
            rec = len(head) + idw + len(tail) + L + 3 + L + 1
            a = np.empty((n, rec), np.uint8)
            c = 0
            a[:, c:c + len(head)] = head; c += len(head)
            ids = np.arange(n, dtype=np.int64)
            for k in range(idw):
                a[:, c + idw - 1 - k] = 48 + (ids // 10 ** k) % 10
            c += idw
            a[:, c:c + len(tail)] = tail; c += len(tail)
            a[:, c:c + L] = hs[lo * L:(lo + n) * L].reshape(n, L); c += L
            a[:, c:c + 3] = np.frombuffer(b"\n+\n", np.uint8); c += 3
            a[:, c:c + L] = hq[lo * L:(lo + n) * L].reshape(n, L); c += L
            a[:, c] = 10
            pth = os.path.join(tmp, "r%d.fq" % mate)
            a.tofile(pth)
            paths.append(pth)
            del a
        cmd = [cli, "-1", paths[0], "-2", paths[1], "-d", os.path.join(tmp, "out"), "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + opt_args
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, FAQCS_MI_TIMING="1"))
        dt = time.perf_counter() - t0
        marks = {}
        for line in r.stderr.decode(errors="replace").splitlines():  # "[faqcs_mi    0.385 s] device context(s) ready"
            if line.startswith("[faqcs_mi") and " s] " in line:
                try:
                    marks[line.split(" s] ", 1)[1].strip()] = float(line[9:].split(" s]")[0])
                except ValueError:
                    pass
        if r.returncode != 0:
            return {"error": r.stderr.decode(errors="replace")[-300:]}
        out_bytes = sum(os.path.getsize(os.path.join(tmp, "out", f)) for f in os.listdir(os.path.join(tmp, "out")))
        # the same command in ONE process (FAQCS_MI_NO_FORK=1): its wall clock includes the release of the GPU context, the pinned buffers and
        # the file mappings, which the default mode leaves to a worker process after the command has returned (report both)
        subprocess.run(["rm", "-rf", os.path.join(tmp, "out")])
        t0 = time.perf_counter()
        r2 = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, env=dict(os.environ, FAQCS_MI_NO_FORK="1"))
        dt_nofork = time.perf_counter() - t0
        in_bytes = sum(os.path.getsize(p) for p in paths)
        pipe = None
        if "first pair parsed" in marks and "outputs written" in marks and marks["outputs written"] > marks["first pair parsed"]:
            pipe = round(2 * n / (marks["outputs written"] - marks["first pair parsed"]) / 1e6, 3)
        one = round(2 * n / dt_nofork / 1e6, 3) if r2.returncode == 0 else None
        # value = the WHOLE-PROCESS figure (one process, teardown included: what the machine pays); the forked mode's figure -- what the
        # caller waits for -- is reported beside it
        return {"value": one, "unit": "M reads/s", "seconds": round(dt_nofork, 3), "pairs": n,
                "value_as_the_caller_sees_it": round(2 * n / dt / 1e6, 3), "seconds_as_the_caller_sees_it": round(dt, 3),
                "pipeline_value": pipe, "stage_marks_s": marks,
                "value_one_process": one, "seconds_one_process": round(dt_nofork, 3),
                "input_GB": round(in_bytes / 1e9, 3), "output_GB": round(out_bytes / 1e9, 3),
                "what": "faqcs_mi: uncompressed FASTQ in /dev/shm -> parse -> pinned SoA -> HIP trim -> trimmed FASTQ + QC.stats.txt in /dev/shm; "
                        "value = the command in ONE process (FAQCS_MI_NO_FORK=1), HIP start-up and teardown included; value_as_the_caller_sees_it = the default mode: the command returns when every output file is complete and a worker process releases the GPU context, the pinned buffers and the mappings afterwards; pipeline_value = the same reads over the interval from the first "
                        "parsed pair to the last output byte (faqcs_mi's own stage marks).  ONE draw: five runs in a row on one box gave 19.7 - 29.5 as the caller "
                        "sees it (profiles/r6u/e2e_five_runs.txt) -- the box's page allocation for 8 GB of output decides, not the code"}
    except Exception as e:
        return {"error": str(e)}
    finally:
        subprocess.run(["rm", "-rf", tmp])


def free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, real_stdout):
    """`bench.py --gpus N` without a launcher: start N rank processes (one per GPU) BEFORE this process touches the GPU
    (torch.cuda.device_count() does not initialise it) and relay rank 0's JSON line.  FAQCS_BENCH_SHARE_GPU=1 lets the
    ranks share device 0 over gloo (a 1-GPU box can then exercise the N-rank path; never a measurement)."""
    import torch

    share = os.environ.get("FAQCS_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if not share and have < a.gpus:
        raise SystemExit("bench: --gpus %d but only %d GPU(s) are visible" % (a.gpus, have))
    port = free_port()
    procs, errs = [], []
    err_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(err_dir, exist_ok=True)
    except OSError:
        err_dir = tempfile.gettempdir()
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if share else r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if share:
            env["FAQCS_BENCH_BACKEND"] = "gloo"
        errs.append(open(os.path.join(err_dir, "rank%d.err" % r), "wb"))  # every rank's stderr is kept
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[-1]))
    # Poll every rank: the first one that exits non-zero takes the job down within seconds (the others would sit in the rendezvous
    # or in a collective until a timeout otherwise).  Only child processes started here are signalled.
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for r, f in enumerate(errs):
            f.close()
            try:
                tail = open(f.name, "rb").read()[-600:].decode(errors="replace")
            except OSError:
                tail = ""
            if tail.strip():
                sys.stderr.write("bench: rank %d stderr (%s), last lines:\n%s\n" % (r, f.name, tail))
        raise SystemExit(rc if rc > 0 else 1)
    reader.join(timeout=10)
    for f in errs:
        f.close()
    real_stdout.write((chunks[0] if chunks else b"").decode())
    real_stdout.flush()
    raise SystemExit(0)


def main():
    a = parse()
    if a.at_frac is not None:
        os.environ["FAQCS_SYNTH_AT"] = repr(a.at_frac)
    # stdout carries exactly ONE JSON line: whatever a library prints there (gloo, the HIP runtime) goes to stderr instead
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if a.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        launch_ranks(a, real_stdout)  # (does not return)
    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        raise SystemExit("bench: --gpus %d disagrees with WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (a.gpus, world))
    backend = os.environ.get("FAQCS_BENCH_BACKEND", "nccl")
    # under torch.distributed.run the process group is formed even for ONE rank: `--nproc-per-node 1` on a one-GPU box then runs the very
    # code an 8-GPU launch runs (RCCL communicator, counter all-reduce on a torch-owned tensor, MAX of the step times)
    use_dist = world > 1 or "WORLD_SIZE" in os.environ
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        rdzv_timeout = datetime.timedelta(seconds=int(os.environ.get("FAQCS_BENCH_RDZV_TIMEOUT", "120")))  # a missing rank fails the job, it does not hang it
        if os.environ.get("FAQCS_BENCH_DIE_RANK") == str(rank):  # (test hook: this rank dies before the rendezvous)
            raise SystemExit(7)
        if backend == "nccl":
            if torch.cuda.device_count() <= local:
                raise SystemExit("bench: rank %d wants GPU %d but %d are visible" % (rank, local, torch.cuda.device_count()))
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=rdzv_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=rdzv_timeout)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import __graft_entry__ as g

    if rank == 0:
        g.build()
    if use_dist:
        dist.barrier()

    env = {"rank": rank, "world": world, "local": local, "dev": dev, "use_dist": use_dist, "backend": backend, "real_stdout": real_stdout}
    if a.e2e_pairs < 0:
        try:
            import shutil

            a.e2e_pairs = 16e6 if shutil.disk_usage("/dev/shm").free > 64e9 else 8e6
        except OSError:
            a.e2e_pairs = 8e6
    out, keep = run_workload(env, a, a.config, a.pairs, a.steps, a.warmup, a.read_len, want_host_sample=(rank == 0 and world == 1))
    if rank == 0:
        L = out["config"]["read_len"]
        opt_args = keep["opt_args"]
        if world == 1 and not a.no_cpu_baseline:
            ns = min(200000, keep["sample_reads"])
            pad = np.zeros(64, np.uint8)
            out["cpu_baseline"] = cpu_baseline(opt_args, np.concatenate([keep["hs"][: ns * L], pad]), np.concatenate([keep["hq"][: ns * L], pad]), L, ns)
        if world == 1 and a.e2e_pairs > 0 and a.config != "kmer" and os.path.exists(os.path.join(ROOT, "faqcs_amd", "faqcs_mi")):
            ne = int(min(a.e2e_pairs, keep["sample_reads"] // 2))
            out["e2e"] = e2e_run(opt_args, keep["hs"][: 2 * ne * L], keep["hq"][: 2 * ne * L], L, ne)
    keep.clear()
    # The other single-GPU configurations of BASELINE.json, at their sizes, next to the headline (N = 1, default invocation):
    # configs[2] (--adapter --polyA on 100 M pairs) and configs[4]'s per-GPU share (25 M pairs 2x250 --kmer_rarefaction)
    if world == 1 and a.config == "plain" and a.read_len is None and not a.no_other_configs:
        others = {}
        for cfg in ("adapter", "kmer"):
            try:
                o2, k2 = run_workload(env, a, cfg, 0.0, 2, 1, None, want_host_sample=False)
                k2.clear()
                rf = o2["roofline"]
                others[cfg] = {"value": o2["value"], "unit": o2["unit"], "ms_per_step": o2["ms_per_step"], "steps": o2["steps"], "warmup": o2["warmup"],
                               "workload": o2["config"]["workload"], "pairs_per_gpu": o2["config"]["pairs_per_gpu"], "read_len": o2["config"]["read_len"],
                               "roofline": {k: rf[k] for k in ("kernel", "kernel_ms", "achieved", "frac", "traffic", "algorithmic_bytes_per_launch", "kernels_ms") if k in rf}}
                for k in ("valu", "kmer_counters"):
                    if k in rf:
                        others[cfg]["roofline"][k] = rf[k]
                if "kmer" in o2:
                    others[cfg]["kmer"] = o2["kmer"]
            except SystemExit as e:
                others[cfg] = {"error": str(e)}
        try:  # --replace_to_N_q is the one option that falls off trim_lds (its G -> N edit needs base and quality of a position together): the cliff, in the line
            o3, k3 = run_workload(env, a, "replaceN", 40e6, 2, 1, None, want_host_sample=False)
            k3.clear()
            others["replace_to_N_q"] = {"value": o3["value"], "unit": o3["unit"], "ms_per_step": o3["ms_per_step"], "workload": o3["config"]["workload"] + " --replace_to_N_q 20",
                                        "pairs_per_gpu": o3["config"]["pairs_per_gpu"], "kernel": o3["roofline"]["kernel"], "frac": o3["roofline"]["frac"],
                                        "relative_to_the_headline": round(o3["value"] / out["value"], 3)}
        except SystemExit as e:
            others["replace_to_N_q"] = {"error": str(e)}
        out["configs"] = others
    if rank == 0:
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if use_dist:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def verify_full_size(blk, lay, fs, results, n_reads, L, config, n_adapters):
    """Consistency of the counter block of ONE job with itself and with the per-read results on the device (outside the timed region).
    results: the batches' result tensors ([n, 4] int16 = start, kept, flags, 1 + credited adapter as u16)."""
    import torch

    from faqcs_amd import _capi as capi

    def part(name):
        o, n = lay[name]
        return blk[o:o + n].astype(np.int64)

    def fail(what, a, b):
        raise SystemExit("bench: full-size check failed: %s: %d != %d" % (what, int(a), int(b)))

    n_valid = kept_sum = 0
    hits = torch.zeros(n_adapters + 2, dtype=torch.int64, device=results[0].device) if n_adapters else None
    for r in results:
        u = r.to(torch.int32) & 0xffff
        valid = (u[:, 2] & 1) != 0
        n_valid += int(valid.sum().item())
        kept_sum += int((u[:, 1].to(torch.int64) * valid).sum().item())
        if hits is not None:
            hits += torch.bincount(u[:, 3].to(torch.int64).clamp(max=n_adapters + 1), minlength=n_adapters + 2)
    tl, tn = int(fs[capi.TOTAL_LENGTH]), int(fs[capi.TOTAL_NUMBER])
    ttl, ttn = int(fs[capi.TOTAL_TRIMMED_LENGTH]), int(fs[capi.TOTAL_TRIMMED_NUMBER])
    if n_valid != ttn: fail("valid results vs TOTAL_TRIMMED_NUMBER", n_valid, ttn)
    if kept_sum != ttl: fail("sum of kept lengths vs TOTAL_TRIMMED_LENGTH", kept_sum, ttl)
    for name, want in (("pre_qual", tl), ("post_qual", ttl), ("pre_base", tl), ("post_base", ttl), ("pre_base_qhist", tl), ("post_base_qhist", ttl)):
        if int(part(name).sum()) != want: fail("sum of " + name, part(name).sum(), want)
    for name, want in (("pre_len_hist", tn), ("post_len_hist", ttn), ("pre_read_qhist", tn), ("post_read_qhist", ttn)):
        if int(part(name).sum()) != want: fail("sum of " + name, part(name).sum(), want)
    ln = np.arange(lay["pre_len_hist"][1], dtype=np.int64)
    if int((part("pre_len_hist") * ln).sum()) != tl: fail("length histogram x length (pre)", (part("pre_len_hist") * ln).sum(), tl)
    if int((part("post_len_hist") * ln).sum()) != ttl: fail("length histogram x length (post)", (part("post_len_hist") * ln).sum(), ttl)
    # position p of the quality / base matrices holds one count per read that is longer than p
    R = lay["pre_len_hist"][1] - 1
    longer = tn - np.cumsum(part("pre_len_hist"))[:R]  # reads with length > p, p = 0 .. R - 1
    for name, width in (("pre_qual", capi.NQ), ("pre_base", capi.NBASE)):
        rows = part(name).reshape(R, width).sum(axis=1)
        if not (rows == longer).all():
            p = int(np.nonzero(rows != longer)[0][0])
            fail("row %d of %s vs the reads longer than %d" % (p, name, p), rows[p], longer[p])
    for name, want in (("pre_comp", tn), ("post_comp", ttn)):  # every read adds one count per kind
        per_kind = part(name).reshape(capi.NCOMP_BIN, capi.NCOMP_KIND).sum(axis=0)
        if not (per_kind == want).all(): fail("reads per kind of " + name, per_kind.min(), want)
    if hits is not None:  # a read credited to adapter j is one of adapter_stats[j]'s reads
        st = part("adapter_stats").reshape(n_adapters, 2)[:, 0]
        h = hits.cpu().numpy()[1:n_adapters + 1]
        if not (h == st).all():
            j = int(np.nonzero(h != st)[0][0])
            fail("reads credited to adapter %d vs adapter_stats" % j, h[j], st[j])


def run_workload(env, a, config, pairs, steps, warmup, read_len, want_host_sample):
    """One configuration: builds the resident data set, times `steps` steps, returns (the JSON object, a dict with a host
    sample of the data for the CPU baseline / the end-to-end run).  Frees its device memory before it returns."""
    import torch

    from faqcs_amd import _capi as capi
    from faqcs_amd import parallel
    from faqcs_amd.engine import HipEngine, _check
    from faqcs_amd.options import parse_args

    rank, world, local, dev, use_dist, backend = env["rank"], env["world"], env["local"], env["dev"], env["use_dist"], env["backend"]
    if use_dist:
        import torch.distributed as dist
    L = read_len or (250 if config == "kmer" else 150)
    if not pairs:
        pairs = 25e6 if config == "kmer" else (125e6 if (config == "plain" and world == 8) else 100e6)
    opt_args = {"adapter": ["--adapter", "--polyA"], "kmer": ["--kmer_rarefaction", "--split_size", "1000000", "--subset", "400"],
                "replaceN": ["--replace_to_N_q", "20"]}.get(config, [])  # replaceN: the one reference flag (trim.cpp:390-403) that leaves trim_lds for trim_filter_accumulate
    opt = parse_args(["-1", "r1", "-2", "r2", "-d", "out", "--ascii", "33", "-q", "5", "--min_L", "50", "--trim_only"] + opt_args)
    R_eng = 256 if L <= 256 else (capi.FAST_READ_LENGTH if L <= capi.FAST_READ_LENGTH else capi.MAX_READ_LENGTH)
    kmer_plan = None
    if config == "kmer":
        # A rank's HBM budget BEFORE anything is allocated: at full size -- 25 M pairs, 2^31 slots -- eight ranks cannot share one
        # GPU, and the first 8-GPU run should not find out by running out of memory half way.  faqcs_kmer_memory_plan is host-only.
        lib0 = capi.load_library()
        holder = capi.ParamsHolder(opt, R_eng, 33, kmer_table_slots=1 << a.kmer_table_log2)
        free0, total0 = torch.cuda.mem_get_info(dev)
        sharing = world if os.environ.get("FAQCS_BENCH_SHARE_GPU") == "1" else 1
        plan = (C.c_uint64 * 5)()
        reads_b = int(2 * pairs) * (2 * L + 12 + 8 + 1) + (1 << 30)
        _check(lib0, lib0.faqcs_kmer_memory_plan(C.byref(holder.p), max(0, free0 // sharing - reads_b), plan, 5))
        table_b, l1_b, l2_b, group_occ, misc_b = (int(x) for x in plan)
        occ = int(2 * pairs) * max(0, L - 30)
        kmer_plan = {"rank": rank, "device": local, "ranks_sharing_the_device": sharing, "device_total_GB": round(total0 / 1e9, 1), "device_free_GB": round(free0 / 1e9, 1),
                     "resident_reads_GB": round(reads_b / 1e9, 1), "kmer_table_GB": round(table_b / 1e9, 1), "group_buffers_GB": round((l1_b + l2_b) / 1e9, 1),
                     "small_arrays_GB": round(misc_b / 1e9, 2), "sum_GB": round((reads_b + table_b + l1_b + l2_b + misc_b) / 1e9, 1),
                     "group_takes_G_occurrences": round(group_occ / 1e9, 2), "pass_has_at_most_G_occurrences": round(occ / 1e9, 2),
                     "pass_counted_in_one_piece": bool(group_occ >= occ)}
        fits = reads_b + table_b + l1_b + l2_b + misc_b <= free0 // sharing
        kmer_plan["fits"] = bool(fits)
        sys.stderr.write("[bench] k-mer HBM budget: %s\n" % json.dumps(kmer_plan))
        if a.dry_run_memory or not fits:
            if not fits:
                sys.stderr.write("[bench] rank %d: %.1f GB do not fit its share of device %d (%.1f GB free / %d ranks): fewer --pairs, a smaller --kmer-table-log2, "
                                 "or one rank per GPU\n" % (rank, kmer_plan["sum_GB"], local, free0 / 1e9, sharing))
            if rank == 0 and a.dry_run_memory:
                env["real_stdout"].write(json.dumps({"dry_run_memory": kmer_plan}) + "\n")
                env["real_stdout"].flush()
            sys.exit(0 if fits else 1)
    eng = HipEngine(opt, R_eng, 33, device=local, kmer_table_slots=(1 << a.kmer_table_log2) if config == "kmer" else 0)
    lib = eng.lib

    # ---- resident synthetic data set ---------------------------------------------------------------------
    n_reads = int(2 * pairs)
    free, _ = torch.cuda.mem_get_info(dev)
    need = n_reads * (2 * L + 12) + (1 << 30)
    if need > free * 0.9:
        n_reads = int((free * 0.9 - (1 << 30)) // (2 * L + 12))
    batch = min(a.batch_reads, (0xFFFFFFFF - max(4096, L)) // L)
    batches = []
    first = rank * n_reads
    done = 0
    adapter_frac = 0.05 if config == "adapter" else 0.0
    while done < n_reads:
        m = min(batch, n_reads - done)
        seq = torch.empty(m * L + 64, dtype=torch.uint8, device=dev)
        qual = torch.empty(m * L + 64, dtype=torch.uint8, device=dev)
        off = torch.empty(m + 1, dtype=torch.int32, device=dev)
        res = torch.empty((m, 4), dtype=torch.int16, device=dev)
        if config == "kmer":  # windows of a fixed 50 Mbp genome, either strand, 0.5 % substitutions
            _check(lib, lib.faqcs_synth_fill_genome(local, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), m, L, 20260101, first + done, 50_000_000))
        else:
            _check(lib, lib.faqcs_synth_fill(local, seq.data_ptr(), qual.data_ptr(), off.data_ptr(), m, L, 20260101, first + done, adapter_frac))
        # the reference's 32 768-read trim() granularity only matters to the adapter pre-pass (groups of 8)
        seg = np.arange(0, m + capi.SEGMENT_READS, capi.SEGMENT_READS, dtype=np.uint32)
        seg[-1] = m
        if len(seg) > 1 and seg[-2] >= m:
            seg = seg[:-1]
        # the optional per-read flags of the batch layout (faqcs_batch::terminal_n: first / last base is 'N'), as a parser would set
        # them while laying the reads down; computed on the device here, with the rest of the synthetic input
        tn = torch.empty(m, dtype=torch.uint8, device=dev)
        _check(lib, lib.faqcs_terminal_n_flags(local, seq.data_ptr(), off.data_ptr(), m, tn.data_ptr()))
        bt = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), m, len(seg) - 1, seg.ctypes.data, L, tn.data_ptr())
        bt_noflags = capi.Batch(seq.data_ptr(), qual.data_ptr(), off.data_ptr(), m, len(seg) - 1, seg.ctypes.data, L, None)
        batches.append((seq, qual, off, res, seg, bt, m, tn, bt_noflags))
        done += m
    torch.cuda.synchronize()

    kmer_seen = [0, 0]
    kmer_wire = [0, 0]  # items sent over the process group, occurrences of the steps (the exchange's wire cost)

    # k-mers across ranks: every canonical k-mer has one owner rank; a rank buckets the (key, epoch) pairs of its
    # shard by owner, RCCL all-to-all moves them, the owner inserts.  The epoch of a 32 768-read segment comes from the GLOBAL
    # sampling schedule (rank-major order of the shards): parallel.rarefaction_schedule.
    kx, kmer_epochs, kmer_points_seq = None, None, None
    if config == "kmer" and world > 1:
        sizes = []
        for _r in range(world):
            for bb in batches:  # (every rank holds batches of the same shape)
                seg = bb[4]
                sizes.extend(int(seg[i + 1] - seg[i]) for i in range(len(seg) - 1))
        ep, kmer_points_seq = parallel.rarefaction_schedule(sizes, opt.split_size, opt.num_subsample)
        per_rank = len(sizes) // world
        kmer_epochs = ep[rank * per_rank:(rank + 1) * per_rank]
        kx = parallel.KmerExchange(eng, rank, world, opt.num_subsample)
    kmer_last = {}
    # The counter all-reduce under the nccl (= RCCL) backend: the library's own communicator reduces the block IN PLACE on the engine's compute
    # stream (faqcs_comm_allreduce_counters) -- but only after the FIRST step has run both forms on the same block and found them bit-identical
    # on every rank: the torch-staged form (export into a torch tensor, dist.all_reduce, import) is the reference, the native form is then used
    # for the timed steps.  A failure to set it up, or a mismatch, keeps the torch-staged form and says so in collective.what.
    # FAQCS_BENCH_NATIVE_RCCL=0 skips the native form altogether.  (No run with two or more GPUs has happened in any round: this is how
    # the first one validates the native collective without a code change.)
    native_rccl = False
    native_state = {"validate": use_dist and dist.get_backend() == "nccl" and os.environ.get("FAQCS_BENCH_NATIVE_RCCL", "1") not in ("", "0"), "note": None}

    def validate_native():
        """Both collectives on the block the first step left: True when the native one may be used."""
        n = eng.n_counters
        before = torch.empty(n, dtype=torch.int64, device=dev)
        eng.counters_export(before.data_ptr(), n)
        parallel.allreduce_counters_device(eng)  # torch-staged (no native communicator yet)
        want = torch.empty(n, dtype=torch.int64, device=dev)
        eng.counters_export(want.data_ptr(), n)
        verdict = []  # (ok, why) from the thread below

        def native_once():
            try:
                torch.cuda.set_device(dev)
                eng.counters_import(before.data_ptr(), n)  # the step's own block again
                parallel.native_comm_init(eng)
                parallel.allreduce_counters_device(eng)  # native, in place
                eng.sync()
                got = torch.empty(n, dtype=torch.int64, device=dev)
                eng.counters_export(got.data_ptr(), n)
                if torch.equal(got, want):
                    verdict.append((1, ""))
                else:
                    verdict.append((0, "the blocks differ in %d of %d words on rank %d" % (int((got != want).sum()), n, rank)))
            except Exception as e:  # no librccl, communicator set-up failed, ...
                verdict.append((0, "%s: %s" % (type(e).__name__, e)))

        # The library's communicator has never met a second rank on real links (no multi-GPU node was available to any round): its set-up and
        # first all-reduce run in a thread with a deadline, so that a rendezvous that never completes costs this run the native collective
        # and not its result -- the ranks then agree (the MIN below, over torch's communicator) to stay with the torch-staged all-reduce.
        import threading

        limit = float(os.environ.get("FAQCS_BENCH_NATIVE_TIMEOUT", "120"))
        th = threading.Thread(target=native_once, daemon=True)
        th.start()
        th.join(limit)
        ok, why = verdict[0] if verdict else (0, "no answer from the native communicator within %.0f s on rank %d" % (limit, rank))
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            eng.has_comm = False  # allreduce_counters_device() goes back to the staging tensor
            eng.counters_import(want.data_ptr(), n)
            native_state["note"] = "native RCCL all-reduce NOT used (%s); torch-staged all-reduce instead" % (why or "another rank reported a failure")
            return False
        native_state["note"] = "validated on the first step against the torch-staged all-reduce: bit-identical blocks on all %d ranks" % world
        return True

    coll = [0.0, 0]  # seconds in the counter all-reduce (export, all-reduce, import; behind a sync, so it is the collective alone), calls

    def step(flags=True):
        _check(lib, lib.faqcs_reset_counters(eng.ctx))  # one step = one job: its counter block starts at zero
        e0 = 0
        for bb in batches:
            res, seg, bt = bb[3], bb[4], (bb[5] if flags else bb[8])
            if kx is not None:
                eng.kmer_set_epochs(kmer_epochs[e0:e0 + len(seg) - 1])
                e0 += len(seg) - 1
            _check(lib, lib.faqcs_submit_device(eng.ctx, C.byref(bt), res.data_ptr()))
            if kx is not None:
                # pipelined: this submission's kernels are enqueued; NOW the items of the previous one are received and handed to the owner side
                # (its insert runs behind these kernels), then this submission's outbox goes on the wire and travels under the next one's kernels
                kx.exchange_end()
                kmer_wire[0] += kx.exchange_begin()[0]
        if config == "kmer" and kx is None:
            # the job's k-mer pass ends INSIDE the step: its k-mers are counted here (in one piece when they fit the group buffers),
            # the histogram of counts is read and the next job starts on an empty table (FaQCs.cpp:518-537)
            eng.kmer_end_table()
            kmer_seen[0], kmer_seen[1] = eng.kmer_totals()  # (of the pass that just ended)
        if kx is not None:  # the additive epoch histograms -> the job's rarefaction points (one small all-reduce); fresh tables
            pts, _hist = kx.finish(kmer_points_seq, n_reads * world)
            kmer_last["points"] = len(pts)
            kmer_last["distinct"], kmer_last["total"] = (pts[-1][1], pts[-1][2]) if pts else (0, 0)
            kmer_wire[1] += int(kmer_last["total"]) // world  # (this rank's share of the occurrences the last point has seen)
        nonlocal native_rccl
        if native_state["validate"]:
            native_state["validate"] = False
            eng.sync()
            native_rccl = validate_native()
        elif use_dist and native_rccl:  # enqueued behind the kernels: no sync in front of it, the one behind it is the step's own
            tc = time.perf_counter()
            parallel.allreduce_counters_device(eng)
            eng.sync()
            coll[0] += time.perf_counter() - tc  # (here: the tail of the kernels + the collective)
            coll[1] += 1
        elif use_dist:
            eng.sync()
            tc = time.perf_counter()
            parallel.allreduce_counters_device(eng)  # the job's one collective: the 1.1 MB counter block, once per job (= step)
            coll[0] += time.perf_counter() - tc
            coll[1] += 1
        else:
            eng.sync()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    ranks_seen = [[rank, local]]
    if use_dist:
        got = [None] * world
        dist.all_gather_object(got, [rank, local, torch.cuda.get_device_properties(local).name])
        ranks_seen = got

    for _ in range(warmup):
        step()
    kt = capi.KernelTimes()
    lib.faqcs_kernel_report(eng.ctx, C.byref(kt))  # reset the kernel timers
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    _check(lib, lib.faqcs_kernel_report(eng.ctx, C.byref(kt)))
    # outside the timed region: the (all-reduced) counter block of the last step must account for every read of the job
    blk = eng.counters()
    lay = capi.python_layout(eng.holder.max_read_length, eng.holder.n_adapters)
    fs = blk[lay["filter_stats"][0]:lay["filter_stats"][0] + 32]
    if int(fs[capi.TOTAL_NUMBER]) != n_reads * world or int(fs[capi.TOTAL_LENGTH]) != n_reads * world * L:
        raise SystemExit("bench: counter block does not add up: %d reads counted, %d expected" % (int(fs[capi.TOTAL_NUMBER]), n_reads * world))
    # ... and, at the job's full size, the accumulators must agree with each other and with the per-read results (size-independent
    # properties: the oracle compares every counter on batches it can finish, tests/test_gpu_parity.py; this is the same block on 2 x 10^8 reads)
    if world == 1:
        verify_full_size(blk, lay, fs, [bb[3] for bb in batches], n_reads, L, config, eng.holder.n_adapters)
    # the same job with batches that carry no terminal_n flags (the kernel then looks at the two end bases of every read itself):
    # outside the timed region, reported beside the headline so that the effect of the flags is visible
    noflag_value = None
    if world == 1 and config == "plain":
        fence()
        t1 = time.perf_counter()
        for _ in range(2):
            step(flags=False)
        fence()
        noflag_value = round(2 * n_reads / (time.perf_counter() - t1) / 1e6, 3)
        lib.faqcs_kernel_report(eng.ctx, C.byref(capi.KernelTimes()))

    out, keep = None, {"opt_args": opt_args}
    if rank == 0:
        total_reads = n_reads * world * steps
        value = total_reads / dt / 1e6
        reads_per_launch = n_reads / max(1, len(batches))
        bytes_per_read = 2 * L + 8 + 8  # seq + qual once, offset + length in (8), start / len / flags out (8): 316 B at L = 150
        alg_bytes = reads_per_launch * bytes_per_read
        trim_kernel = (kt.trim_kernel or b"").decode() or "trim"
        # the roofline entry describes the DOMINANT kernel of the configuration: adapter_overlap for --config adapter
        # (its algorithmic bytes: the bases once + 6 bytes of result per read), the trim kernel otherwise
        dominant, dom_ms = trim_kernel, kt.trim_ms
        if config == "adapter" and kt.adapter_ms > kt.trim_ms:
            dominant, dom_ms = "adapter_overlap", kt.adapter_ms
            alg_bytes = reads_per_launch * (L + 4 + 6)
        kmer_k = 31
        if config == "kmer" and (kt.kmer_ms + kt.kmer_insert_ms) > kt.trim_ms:
            # + (L - k + 1) x 16 bytes per read (8-byte key + 8-byte slot read-modify-write): 3 520 B/read at L = 250, k = 31
            dominant = "skm_extract16 + skm_extract + skm_split + skm_combine" if kx is None else "skm_extract16 + skm_outbox + skm_items + skm_split + skm_combine"
            if os.environ.get("FAQCS_KMER_DIRECT") == "1" and kx is None:
                dominant = "kmer_count"
            dom_ms = kt.kmer_ms + kt.kmer_insert_ms
            alg_bytes = reads_per_launch * (L - kmer_k + 1) * 16
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src, traffic_at = None, None, None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import load_if_current  # a stored counter file counts only if it was measured on THESE kernel sources

        tf = os.path.join(ROOT, "profiles", "traffic_%s.json" % config)
        if L == 150 and dominant != "adapter_overlap":  # (measured on the 2x150 shape only)
            # NOT measured in this run: rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE separately, gfx950 correction applied)
            # of the same kernel, stored per read by profiles/pmc_traffic2.sh and scaled to this run's launch size
            tj, why = load_if_current(tf)
            if tj is None:
                traffic_src = "null: " + why
            elif tj.get("kernel", trim_kernel) != trim_kernel:
                traffic_src = "null: %s holds %s, this run's kernel is %s" % (os.path.basename(tf), tj.get("kernel"), trim_kernel)
            else:
                traffic = int(tj["hbm_bytes_per_read"] * reads_per_launch)
                traffic_at = tj.get("reads_per_launch")
                traffic_src = "profiles/traffic_%s.json (rocprofv3 --pmc, %s; source_sha256 %s... = this build)" % (config, tj.get("tag", "stored per read"), tj["source_sha256"][:10])
        out = {
            "metric": "M reads/sec (paired 2x%dbp)" % L, "value": round(value, 3), "unit": "M reads/s", "n_gpus": world if world == 1 else dist.get_world_size(),
            "ranks_seen": ranks_seen,
            "reduced_block": {"reads_counted": int(fs[capi.TOTAL_NUMBER]), "reads_expected": int(n_reads * world),
                              "note": "TOTAL_NUMBER of the counter block after the job's all-reduce against ranks x reads per rank (a mismatch ends the run)"},
            "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("synthetic %.0fM-pair 2x%dbp Q33 reads" + ("" if a.at_frac is None else " (A+T = %g of the bases)" % a.at_frac) + " resident in HBM per GPU, BWA_plus -q 5 --min_L 50%s, "
                                    "1 step = 1 pass (%s%s%s + counter all-reduce)")
                                   % (n_reads / 2e6, L, {"adapter": " --adapter --polyA (5 percent read-through)", "kmer": " --kmer_rarefaction, genome-sampled reads"}.get(config, ""),
                                      {"adapter": "adapter_overlap, then ", "kmer": ""}.get(config, ""), trim_kernel, {"kmer": " + the k-mer kernels"}.get(config, "")),
                       "pairs_per_gpu": n_reads // 2, "read_len": L, "launches_per_step": len(batches) * (2 if config == "adapter" else 1),
                       "M_pairs_per_s": round(value / 2, 3), "terminal_n_supplied": True},
        }
        if noflag_value is not None:
            out["config"]["value_without_terminal_n_flags"] = noflag_value
        if kmer_plan is not None:
            out["config"]["hbm_budget"] = kmer_plan
        if coll[1]:
            out["collective"] = {"what": ("all-reduce(sum) of the %d-word u64 counter block, once per job, IN PLACE on the engine's compute stream by the library's own RCCL "
                                          "communicator (faqcs_comm_allreduce_counters); ms_per_step includes the wait for the step's last kernels; %s" % (eng.n_counters, native_state["note"])) if native_rccl else
                                         ("all-reduce(sum) of the %d-word u64 counter block, once per job: device-to-device export into a torch tensor, "
                                          "dist.all_reduce (%s), import%s" % (eng.n_counters, backend, "; " + native_state["note"] if native_state["note"] else "")),
                                 "ms_per_step": round(coll[0] / coll[1] * 1e3, 4), "calls": coll[1]}
        if config == "kmer" and kx is not None:
            out["kmer"] = {"G_inserts_per_s": round(kmer_last.get("total", 0) / (dt / steps) / 1e9, 3), "distinct_at_last_point": int(kmer_last.get("distinct", 0)),
                           "occurrences_at_last_point": int(kmer_last.get("total", 0)), "points": int(kmer_last.get("points", 0)),
                           "note": "owner-partitioned tables: runs of k-mers that share their minimizer (16-byte super-k-mer items, about 8 occurrences each) "
                                   "grouped by the owner of the minimizer's partition on the device, all-to-all over the process group, owner-side combine; "
                                   "distinct / total from the all-reduced epoch histograms.  The exchange is pipelined: the items of submission i travel, and the "
                                   "owner combines them, under the trim and extraction kernels of submission i + 1 (exchange_marks: seconds since the first mark)",
                           "wire_bytes_per_occurrence": round(16.0 * kmer_wire[0] / max(1, kmer_wire[1]), 3),
                           "exchange_marks": kx.marks[:24]}
        elif config == "kmer":
            d_, t_ = eng.kmer_totals()
            out["kmer"] = {"G_inserts_per_s": round(t_ / (dt / steps) / 1e9, 3), "distinct_per_step": int(d_), "occurrences_per_step": int(t_),
                           "points_per_step": len(eng.kmer_points()) // max(1, steps + warmup),
                           "note": "canonical 31-mers of the kept reads: runs of k-mers that share their minimizer travel as ONE 16-byte item (about 8 occurrences) and stay "
                                   "in HBM until the pass ends; then they are split 2^19 ways by the minimizer (a sort of 8 192-item tiles in LDS) and every fine "
                                   "partition is expanded and counted in LDS -- its keys go straight into the histogram of counts and the keys-by-first-epoch "
                                   "histogram, the device table is not touched (DESIGN.md section 4.4)"}
        out.update({
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_measured_at_reads_per_launch": traffic_at, "kernel": dominant,
                         "kernel_ms": round(dom_ms, 4), "launches": int(kt.n_launches),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "kernels_ms": {trim_kernel: round(kt.trim_ms, 4), "adapter_overlap": round(kt.adapter_ms, 4),
                                        "kmer kernels (per submission)": round(kt.kmer_ms, 4), "kmer_insert_items": round(kt.kmer_insert_ms, 4)}},
        })
        if noflag_value is not None and value > 0:  # (VERDICT r5 hygiene: the conservative figure beside the headline's)
            out["roofline"]["frac_without_terminal_n_flags"] = round(achieved / HBM_PEAK_GBS * noflag_value / value, 5)
        if config == "kmer" and "kmer" in out and dom_ms > 0:
            occ = out["kmer"].get("occurrences_per_step", out["kmer"].get("occurrences_at_last_point", 0)) / max(1, len(batches))
            rate = occ / (dom_ms * 1e-3) / 1e9
            kc = {"G_occurrences_per_s": round(rate, 3),
                  "note": "occurrences per submission / the k-mer kernels' time.  Rounds 1-3 paid one memory-side atomic per occurrence (13.5 G/s measured ceiling), "
                          "round 4 moved an 8-byte item per occurrence through two scatter passes (61 B of HBM traffic per occurrence), round 5 a 16-byte run "
                          "of ~8 occurrences and one table update per distinct key and group (22 B); round 6 counts a pass in one piece at its end and does not "
                          "touch the table: the counters of its kernels are in profiles/r6*/pmc_skm*.txt"}
            kj, why = load_if_current(os.path.join(ROOT, "profiles", "kmer_counters.json"))
            if kj is not None:
                kc.update(kj)
            else:
                kc["counters"] = "null: " + why
            out["roofline"]["kmer_counters"] = kc
        if dominant == "adapter_overlap":
            # adapter_overlap is bound by integer VALU issue, not by HBM: the work is L x sum|adapter| cell updates per
            # read (62 850 for the 9 built-ins + polyA at L = 150); the instruction count per read comes from the PMC profile
            cells = 62850.0 * L / 150.0
            # VALU and scalar issue: SQ_INSTS_VALU / SQ_INSTS_SALU + SQ_INSTS_BRANCH per read of adapter_overlap from its latest PMC profile
            # (profiles/adapter_counters.json, stored like the k-mer counters) x reads/s.  The vector side is priced against the MEASURED issue
            # rate of the instruction class the kernel is made of (v_bitop3, v_bcnt, v_alignbit, DPP, compares: 4 cycles per wave instruction =
            # 575 G/s chip-wide; only plain VOP1/VOP2 integer ops reach 1 000 G/s -- profiles/r3a/valu_lds_peak.txt), the scalar side against
            # one scalar instruction per clock and CU (measured with trim_long).
            ac, ac_why = load_if_current(os.path.join(ROOT, "profiles", "adapter_counters.json"))
            ac = ac or {}
            valu = {"cell_updates_per_s": round(reads_per_launch * cells / (dom_ms * 1e-3) / 1e12, 3), "unit": "T cell updates/s",
                    "measured_peak_G_wave_instr_per_s": 575.0, "G_wave_instr_per_s": None, "frac": None}
            if ac.get("valu_per_read"):
                reads_per_s = reads_per_launch / (dom_ms * 1e-3)
                valu_per_read = float(ac["valu_per_read"]) * L / 150.0
                g_instr = reads_per_s * valu_per_read / 1e9
                scal_per_read = (float(ac.get("salu_per_read", 0.0)) + float(ac.get("branches_per_read", 0.0))) * L / 150.0
                n_cu = torch.cuda.get_device_properties(0).multi_processor_count
                clk = float(getattr(torch.cuda.get_device_properties(0), "clock_rate", 0) or 2.4e6) * 1e3  # (kHz; MI355X: 2.4 GHz peak)
                valu.update({"G_wave_instr_per_s": round(g_instr, 1), "frac": round(g_instr / 575.0, 4), "valu_instr_per_read": round(valu_per_read, 1),
                             "scalar_instr_per_read": round(scal_per_read, 1),
                             "scalar_instr_per_cu_clock": round(reads_per_s * scal_per_read / (n_cu * clk), 3),
                             "counters_source": ac.get("counters_source"),
                             "note": "co-bound by vector issue (frac: wave instructions/s against the measured 575 G/s of 4-cycle instructions) and scalar issue "
                                     "(scalar_instr_per_cu_clock against 1 per clock and CU at the device's peak clock); peak from profiles/r3a/valu_lds_peak.txt"})
            out["roofline"]["valu"] = valu
        if want_host_sample:
            ns = int(min(max(400000, 2 * a.e2e_pairs if config != "kmer" else 0), batches[0][6]))
            keep["hs"] = batches[0][0][: ns * L].cpu().numpy()
            keep["hq"] = batches[0][1][: ns * L].cpu().numpy()
            keep["sample_reads"] = ns
    # release the device memory of this configuration (the next one builds its own data set)
    eng.close()
    del batches
    torch.cuda.empty_cache()
    return out, keep


if __name__ == "__main__":
    main()
