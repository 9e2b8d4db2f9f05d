#!/bin/bash
# Round-6 evidence on the GPU box (through gpurun, from the repo root): bash profiles/collect_r6.sh <tag>  -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/<tag>/ and the three stamped counter files into profiles/).
# Order matters: the counter files are measured FIRST and stamped with the sha256 of the kernel sources (tools/source_hash.py); bench.py
# then finds them current and prices `traffic` / `valu` / `kmer_counters` with them -- a file from another build reads "null: ..." there.
set -u
tag=${1:-r6}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
b() { timeout 900 python3 bench.py "$@" 2>> $out/bench.err < /dev/null; }
prof() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o $name -- "$@" < /dev/null > $out/prof_$name.log 2>&1
         f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/rocprofv3_kernel_stats_$name.csv; }
# 1. counters, stamped
[ -x profiles/microbench/trim_ab ] || g++ -O2 -std=c++17 -o profiles/microbench/trim_ab profiles/microbench/trim_ab.cpp -Iinclude -ldl -L/opt/rocm/lib -lamdhip64 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
timeout 900 bash profiles/pmc_traffic2.sh $tag/traffic faqcs_amd/libfaqcs_mi.so 28633087 > $out/traffic_plain.raw.json 2> $out/traffic.err < /dev/null
timeout 900 bash profiles/pmc_adapter.sh $tag/adapter_pmc > $out/pmc_adapter.txt 2>&1 < /dev/null
timeout 900 bash profiles/pmc_skm.sh $tag/skm_pmc 16e6 31 > $out/pmc_skm.txt 2>&1 < /dev/null
[ -x profiles/microbench/trim_ab ] && timeout 600 bash profiles/pmc_trim.sh $tag/trim_pmc faqcs_amd/libfaqcs_mi.so 16777216 150 > $out/pmc_trim_lds.txt 2>&1 < /dev/null
python3 - "$out" "$tag" <<'PY'
import json, os, re, sys
sys.path.insert(0, "tools")
from source_hash import stamp
out, tag = sys.argv[1], sys.argv[2]
def put(name, d):
    d = stamp(d)
    for p in (os.path.join(out, name), os.path.join("profiles", name)):
        json.dump(d, open(p, "w"), indent=1)
try:
    t = json.load(open(os.path.join(out, "traffic_plain.raw.json")))
    t["tag"] = tag + ", measured at the bench launch size"
    t["algorithmic_bytes_per_read"] = 316
    put("traffic_plain.json", t)
except Exception as e:
    print("traffic: %s" % e)
try:
    txt = open(os.path.join(out, "pmc_adapter.txt")).read()
    m = re.search(r"adapter_overlap_pair[^{]*(\{.*\})\s+per read", txt) or re.search(r"adapter_overlap[^{]*(\{.*\})\s+per read", txt)
    c = eval(m.group(1))
    put("adapter_counters.json", {"valu_per_read": c["SQ_INSTS_VALU"], "salu_per_read": c["SQ_INSTS_SALU"], "branches_per_read": c["SQ_INSTS_BRANCH"], "lds_per_read": c["SQ_INSTS_LDS"],
                                  "counters_source": "profiles/%s/pmc_adapter.txt (rocprofv3 --pmc over tools/ablate.py, 8 M reads of 150 bases, --adapter --polyA, 5 %% read-through; NOT measured in the bench run; scaled with the read length)" % tag})
except Exception as e:
    print("adapter: %s" % e)
try:
    txt = open(os.path.join(out, "pmc_skm.txt")).read()
    occ = float(re.search(r"occurrences (\d+), distinct keys (\d+) \(([\d.]+) per occurrence\)", txt).group(1))
    dpo = float(re.search(r"\(([\d.]+) per occurrence\)", txt).group(1))
    ker = {}
    for blk in re.split(r"\n(?=\w+: \d+ launches)", txt):
        name = blk.split(":")[0].strip()
        ker[name] = {m.group(1): float(m.group(2)) for m in re.finditer(r"\s+(\w+)\s+([\d.]+) per occurrence", blk)}
    rd = wr64 = wr = at = 0.0
    per_kernel = {}
    for k, c in ker.items():
        if not k.startswith("skm_"): continue
        rd_k, wr_k, w64_k = c.get("TCC_EA0_RDREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        rd += rd_k; wr += wr_k; wr64 += w64_k; at += c.get("TCC_EA0_ATOMIC_sum", 0.0)
        # reads: the item streams and the reads are wide coalesced loads (a 128-byte request tallied as one: x 128 B); the table probes of
        # skm_combine's write-out are single 64-byte sectors -- priced at 128 B too here, the conservative reading of the guide's rule
        per_kernel[k] = {"read_requests": rd_k, "write_requests": wr_k, "bytes_upper": rd_k * 128 + w64_k * 64 + (wr_k - w64_k) * 32,
                         "bytes_sector_reading": rd_k * 64 + w64_k * 64 + (wr_k - w64_k) * 32,
                         "valu": c.get("SQ_INSTS_VALU"), "salu": c.get("SQ_INSTS_SALU"), "branch": c.get("SQ_INSTS_BRANCH"), "lds": c.get("SQ_INSTS_LDS")}
    put("kmer_counters.json", {"TCC_EA0_ATOMIC_per_occurrence": round(at, 5), "fabric_read_requests_per_occurrence": round(rd, 4),
                               "fabric_write_requests_per_occurrence": round(wr, 4),
                               "hbm_bytes_per_occurrence": round(rd * 128 + wr64 * 64 + (wr - wr64) * 32, 1),
                               "read_request_bytes": "128: settled by profiles/microbench/req_width.hip (profiles/r6b/req_width*.txt: a random 16-byte load costs ONE fabric read request, and so do two loads of the two 64-byte sectors of one random 128-byte line, at the same 37.5 G requests/s)",
                               "distinct_keys_per_occurrence": dpo, "per_kernel_per_occurrence": per_kernel,
                               "counters_source": "profiles/%s/pmc_skm.txt (rocprofv3 --pmc, separate SQ / TCC passes over tools/kmer_bench.py, 16 M reads of 250 bases, 2^31 slots; NOT measured "
                                                  "in the bench run; hbm_bytes_per_occurrence prices a read request at 128 B and a write request at its counted width)" % tag})
except Exception as e:
    print("kmer: %s" % e)
PY
# 2. the driver's command: the headline + BASELINE's adapter and k-mer configurations + cpu_baseline (best of -t 1/8/16/all) + e2e
b > $out/bench_default_all_configs.json
[ "${2:-}" = counters ] && { ls -la $out; exit 0; }   # (bash profiles/collect_r6.sh <tag> counters: the stamped counter files and the driver's line only)
# 3. other shapes / configurations through the same harness
for L in 100 125 250 300; do b --read-len $L --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_${L}bp_40Mpairs.json; done
b --read-len 75 --pairs 100e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_75bp_100Mpairs.json
b --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer_250bp_25Mpairs.json
FAQCS_KMER_FINAL=0 b --config kmer --steps 2 --no-cpu-baseline > $out/bench_kmer_through_the_table_round5_path.json   # (A/B: every group into the table, as in round 5; same build, same box)
FAQCS_ADAPTER_PAIR=0 b --config adapter --steps 2 --no-cpu-baseline --no-other-configs --e2e-pairs 0 > $out/bench_adapter_one_read_per_wave.json   # (A/B: adapter_overlap instead of adapter_overlap_pair)
FAQCS_TAIL_FOLD=0 b --steps 10 --no-cpu-baseline --no-other-configs --e2e-pairs 0 > $out/bench_plain_fold_on_the_aux_stream.json   # (A/B: composition_histogram beside the next launch instead of in trim_lds's tail)
FAQCS_KMER_EXTRACT16=0 b --config kmer --steps 2 --no-cpu-baseline > $out/bench_kmer_general_extraction_kernel_only.json   # (A/B: skm_extract for every read)
FAQCS_KMER_DIRECT=1 b --config kmer --steps 2 --no-cpu-baseline > $out/bench_kmer_direct_one_atomic_per_occurrence.json   # rounds 1-3's path, same build, same box
FAQCS_BENCH_SHARE_GPU=1 b --config kmer --gpus 2 --pairs 4e6 --steps 2 --no-cpu-baseline --kmer-table-log2 29 > $out/bench_kmer_2ranks_shared_gpu.json  # (the N-rank path on one GPU: not a measurement)
FAQCS_BENCH_SHARE_GPU=1 b --config kmer --gpus 8 --pairs 5e5 --steps 2 --no-cpu-baseline --kmer-table-log2 28 > $out/bench_kmer_8ranks_shared_gpu.json
FAQCS_BENCH_SHARE_GPU=1 b --gpus 8 --pairs 8e6 --steps 2 --no-cpu-baseline --no-other-configs --e2e-pairs 0 > $out/bench_plain_8ranks_shared_gpu.json
# (a distinct-heavy input: 16 M reads of a 465 Mbp genome, 2^30 slots at load 0.72 -- every partition takes several LDS rounds per group.  Before the
#  device-scope fence between the rounds of skm_combine was dropped (82e208c) this line read 413.5 ms / 38.7 M reads/s on the same box type.)
FAQCS_KMER_STATS=1 timeout 300 python3 tools/kmer_bench.py 16e6 250 30 465e6 > $out/kmer_bench_distinct_heavy_16Mreads_465Mbp.txt 2>&1 < /dev/null
FAQCS_KMER_STATS=1 KMER_BENCH_TABLE=1 timeout 300 python3 tools/kmer_bench.py 16e6 250 30 465e6 > $out/kmer_bench_distinct_heavy_through_the_table.txt 2>&1 < /dev/null
timeout 120 python3 bench.py --config kmer --dry-run-memory > $out/bench_kmer_dry_run_memory_1rank.json 2> $out/bench_kmer_dry_run_memory_1rank.err < /dev/null
FAQCS_BENCH_SHARE_GPU=1 timeout 120 python3 bench.py --config kmer --gpus 8 --dry-run-memory > $out/bench_kmer_dry_run_memory_8ranks_one_gpu.json 2> $out/bench_kmer_dry_run_memory_8ranks_one_gpu.err < /dev/null; cat gpurun_out/rank0.err >> $out/bench_kmer_dry_run_memory_8ranks_one_gpu.err 2>/dev/null
# 4. rocprofv3 --kernel-trace --stats over shorter runs of the same commands
prof bench_plain_43Mpairs python3 bench.py --pairs 42949630 --steps 3 --no-cpu-baseline --e2e-pairs 0 --no-other-configs
prof bench_kmer_25Mpairs python3 bench.py --config kmer --steps 2 --no-cpu-baseline
prof bench_adapter_43Mpairs python3 bench.py --config adapter --pairs 42949630 --steps 2 --no-cpu-baseline --e2e-pairs 0
ls -la $out
# 5. compressed input end to end (plain / BGZF / ordinary gzip with the A/B switches of the inflate work), with the per-role thread accounting
FAQCS_E2E_GZ=1 timeout 1500 python3 tools/e2e_big.py 8e6 2>&1 < /dev/null | grep -E "^mapped|^streaming|input|threads .|main thread:|parsers:" > $out/e2e_gz_8Mpairs.txt
rm -rf /dev/shm/faqcs_e2e_big
# 6. the whole GPU suite on this build
timeout 900 python -m pytest tests -x -q -m gpu > $out/gpu_suite.txt 2>&1 < /dev/null
tail -2 $out/gpu_suite.txt
