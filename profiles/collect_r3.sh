#!/bin/bash
# Round-3 evidence on the GPU box (through gpurun, from the repo root): bash profiles/collect_r3.sh <tag>  -> gpurun_out/<tag>/
# (copy what is to be judged into profiles/<tag>/).  Needs the in-tree builds: libfaqcs_mi.so, libfaqcs_mi_stamps.so
# (profiles/build_stamps.sh), profiles/microbench/{trim_ab,valu_lds_peak,lds_mix}.
set -u
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
b() { python3 bench.py "$@" 2>> $out/bench.err; }
# 1. bench lines: the three BASELINE configurations at BASELINE's sizes, then other read lengths through the same harness
b > $out/bench_plain_100Mpairs.json
b --config adapter --e2e-pairs 0 > $out/bench_adapter_100Mpairs.json
b --config kmer --no-cpu-baseline > $out/bench_kmer_250bp_25Mpairs.json
for L in 100 125 250 300; do b --read-len $L --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_${L}bp_40Mpairs.json; done
for L in 50 75 128; do b --read-len $L --pairs 60e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_${L}bp_60Mpairs.json; done
b --read-len 600 --pairs 8e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_600bp_8Mpairs.json
b --at-frac 0.9 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_at90_40Mpairs.json
FAQCS_BENCH_SHARE_GPU=1 b --gpus 2 --pairs 10e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_2ranks_shared_gpu.json   # (the launch path on two ranks sharing the GPU: not a measurement)
FAQCS_BENCH_SHARE_GPU=1 b --config kmer --gpus 2 --pairs 4e6 --steps 2 --no-cpu-baseline > $out/bench_kmer_2ranks_shared_gpu.json
# 2. rocprofv3 --kernel-trace --stats over shorter runs of the same commands (same kernels, same launch size)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_plain -o plain -- python3 bench.py --pairs 42949630 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/prof_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_adapter -o adapter -- python3 bench.py --config adapter --pairs 14316543 --steps 2 --no-cpu-baseline --e2e-pairs 0 > $out/prof_adapter.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_kmer -o kmer -- python3 bench.py --config kmer --pairs 8e6 --steps 2 --no-cpu-baseline > $out/prof_kmer.log 2>&1
cp $out/prof_plain/plain_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_plain_43Mpairs.csv
cp $out/prof_adapter/adapter_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_adapter_14Mpairs.csv
cp $out/prof_kmer/kmer_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_kmer_8Mpairs.csv
# 3. the trim kernel: SQ counters per read (three --pmc passes), HBM traffic (four --pmc passes), section clocks (stamps build)
bash profiles/pmc_trim.sh $tag/pmc > $out/pmc_instruction_mix.txt 2>&1
bash profiles/pmc_traffic2.sh $tag/traffic > $out/traffic_plain.json 2> $out/traffic.err
TRIM_AB_STAMPS=1 ./profiles/microbench/trim_ab 16777216 150 3 profiles/microbench/libfaqcs_mi_stamps.so > $out/section_stamps.txt 2>&1
./profiles/microbench/trim_ab 16777216 150 4 profiles/microbench/libfaqcs_mi_r2.so faqcs_amd/libfaqcs_mi.so > $out/ab_round2_vs_round3.txt 2>&1
TRIM_AB_SYNC_EACH=1 ./profiles/microbench/trim_ab 16777216 150 4 faqcs_amd/libfaqcs_mi.so > $out/ab_no_composition_overlap.txt 2>&1
# 4. adapter_overlap and kmer_count counters (scripts of round 2)
bash profiles/pmc_adapter.sh $tag/adapter 0.05 > $out/pmc_adapter.txt 2>&1
if [ "${SKIP_KMER_PMC:-0}" != 1 ]; then bash profiles/pmc_kmer.sh $tag/kmer > $out/pmc_kmer_atomics.txt 2>&1; fi
# 5. instruction rates
./profiles/microbench/lds_mix > $out/lds_mix.txt 2>&1
ls -la $out
