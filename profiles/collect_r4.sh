#!/bin/bash
# Round-4 evidence on the GPU box (through gpurun, from the repo root): bash profiles/collect_r4.sh <tag>  -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/<tag>/).
set -u
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
b() { python3 bench.py "$@" 2>> $out/bench.err; }
# 1. the driver's command: the headline + BASELINE's adapter and k-mer configurations + cpu_baseline (best of -t 1/8/16/all) + e2e
b > $out/bench_default_all_configs.json
# 2. other shapes through the same harness
for L in 100 125 192 200 250 251 300; do b --read-len $L --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_${L}bp_40Mpairs.json; done
for L in 96 128; do b --read-len $L --pairs 60e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_${L}bp_60Mpairs.json; done
for L in 50 64 75; do b --read-len $L --pairs 100e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_${L}bp_100Mpairs.json; done
for L in 50 75; do FAQCS_TRIM_LDS4=0 b --read-len $L --pairs 100e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_${L}bp_100Mpairs_trim_tpr.json; done   # rounds 1-3's kernel for these lengths
FAQCS_TRIM_LDS16=0 b --read-len 250 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/bench_plain_250bp_40Mpairs_trim_filter_accumulate.json   # round 3's kernel for this length, same build, same box
FAQCS_KMER_DIRECT=1 b --config kmer --steps 2 --no-cpu-baseline > $out/bench_kmer_direct_one_atomic_per_occurrence.json   # rounds 1-3's path, same build, same box
b --config kmer --steps 3 --no-cpu-baseline > $out/bench_kmer_250bp_25Mpairs.json
FAQCS_BENCH_SHARE_GPU=1 b --config kmer --gpus 2 --pairs 4e6 --steps 2 --no-cpu-baseline > $out/bench_kmer_2ranks_shared_gpu.json  # (the N-rank path on one GPU: not a measurement)
# 3. rocprofv3 --kernel-trace --stats over shorter runs of the same commands
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_plain -o plain -- python3 bench.py --pairs 42949630 --steps 3 --no-cpu-baseline --e2e-pairs 0 --no-other-configs > $out/prof_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_kmer -o kmer -- python3 bench.py --config kmer --steps 2 --no-cpu-baseline > $out/prof_kmer.log 2>&1
cp $out/prof_plain/plain_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_plain_43Mpairs.csv
cp $out/prof_kmer/kmer_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_kmer_25Mpairs.csv
# 4. counters: the k-mer kernels per occurrence (SQ + TCC passes), trim_long per 64 positions
bash profiles/pmc_kmer_group.sh $tag/kmer_pmc 8e6 30 > $out/pmc_kmer_group.txt 2>&1
bash profiles/pmc_trim_long.sh $tag/trim_long > $out/pmc_trim_long.txt 2>&1
bash profiles/pmc_adapter.sh $tag/adapter_pmc > $out/pmc_adapter.txt 2>&1
[ -x profiles/microbench/trim_ab ] && bash profiles/pmc_trim.sh $tag/trim16_pmc faqcs_amd/libfaqcs_mi.so 8388608 250 > $out/pmc_trim_lds16.txt 2>&1
[ -x profiles/microbench/trim_ab ] && bash profiles/pmc_trim.sh $tag/trim4_pmc faqcs_amd/libfaqcs_mi.so 33554432 75 > $out/pmc_trim_lds4.txt 2>&1
[ -x profiles/microbench/trim_ab ] && bash profiles/pmc_trim.sh $tag/trim_pmc faqcs_amd/libfaqcs_mi.so 16777216 150 > $out/pmc_trim_lds.txt 2>&1
bash tools/e2e_probe.sh > $out/e2e_probe.txt 2>&1
ls -la $out
