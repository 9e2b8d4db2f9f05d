#!/bin/bash
# Collects the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh <tag>            -> gpurun_out/<tag>/...   (copy what you want judged into profiles/<tag>/)
# 1. bench.py default line (BASELINE configs[1], 100 M pairs) and the adapter configuration
# 2. rocprofv3 --kernel-trace --stats over a shorter bench run (same kernels, same launch size)
# 3. HBM traffic: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with other trace domains)
set -u
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "${SKIP_BENCH:-0}" != 1 ]; then
python3 bench.py > $out/bench_plain_100Mpairs.json 2> $out/bench_plain.err
FAQCS_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --pairs 10e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_2ranks_shared_gpu.json 2>> $out/bench_plain.err  # (FAQCS_BENCH_SHARE_GPU=1 in the environment: the launch path, not a measurement)
python3 bench.py --config adapter --pairs 20e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_adapter_20Mpairs.json 2> $out/bench_adapter.err
python3 bench.py --read-len 250 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_250bp_40Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 300 --pairs 20e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_300bp_20Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 100 --pairs 60e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_100bp_60Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 75 --pairs 60e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_75bp_60Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 50 --pairs 60e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_50bp_60Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 125 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_125bp_40Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --read-len 600 --pairs 8e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_600bp_8Mpairs.json 2>> $out/bench_plain.err
python3 bench.py --config adapter --read-len 300 --pairs 4e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_adapter_300bp_4Mpairs.json 2>> $out/bench_adapter.err
python3 bench.py --config kmer --no-cpu-baseline --steps 3 > $out/bench_kmer_250bp_10Mpairs.json 2>> $out/bench_plain.err
FAQCS_BENCH_SHARE_GPU=1 python3 bench.py --config kmer --gpus 2 --pairs 4e6 --steps 2 --no-cpu-baseline > $out/bench_kmer_2ranks_shared_gpu.json 2>> $out/bench_plain.err  # (the exchange path on two ranks sharing the GPU: not a measurement)
python3 bench.py --at-frac 0.9 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_at90_40Mpairs.json 2>> $out/bench_plain.err   # AT-rich genome: the dinucleotide counter on most reads
python3 bench.py --read-len 128 --pairs 40e6 --no-cpu-baseline --e2e-pairs 0 > $out/bench_plain_128bp_40Mpairs.json 2>> $out/bench_plain.err  # (stays on trim_tpr: LDS bank stride)
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_plain -o plain -- python3 bench.py --pairs 42949630 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/prof_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_adapter -o adapter -- python3 bench.py --config adapter --pairs 4e6 --steps 3 --no-cpu-baseline --e2e-pairs 0 > $out/prof_adapter.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o pmc -- python3 tools/ablate.py 0 16e6 > $out/pmc_$c.log 2>&1
done
python3 profiles/pmc_traffic.py $out 16000000 $tag > $out/traffic_plain.json 2> $out/traffic.err
# 4. instruction / wait / LDS counters per read: the trim kernel (two passes), adapter_overlap, kmer_count's atomics
bash profiles/pmc_insts.sh $tag/insts1 > $out/pmc_instruction_mix.txt 2>&1
PMC="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SMEM" bash profiles/pmc_insts.sh $tag/insts2 >> $out/pmc_instruction_mix.txt 2>&1
bash profiles/pmc_adapter.sh $tag/adapter 0.05 > $out/pmc_adapter.txt 2>&1
if [ "${SKIP_KMER_PMC:-0}" != 1 ]; then bash profiles/pmc_kmer.sh $tag/kmer > $out/pmc_kmer_atomics.txt 2>&1; fi
cp $out/prof_plain/plain_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_plain_43Mpairs.csv; cp $out/prof_adapter/adapter_kernel_stats.csv $out/rocprofv3_kernel_stats_bench_adapter_4Mpairs.csv
ls -la $out
