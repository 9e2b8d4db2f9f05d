#!/bin/bash
# kernel-level A/B of library variants on the k-mer bench (rocprofv3 --kernel-trace --stats; the step time of one bench run is noisy by +-3 ms):
#   bash tools/ab_kernel.sh <kernel name prefix> <lib> [<lib> ...]
export TMPDIR=/tmp
k=$1; shift
for lib in "$@"; do
  d=gpurun_out/abk_$(basename $lib .so)
  rm -rf $d
  FAQCS_MI_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o s -- python3 bench.py --config kmer --steps 3 --no-cpu-baseline < /dev/null > $d.log 2>&1
  echo "$lib: $(python3 tools/kstats.py $d 12 | grep "^$k" | head -1)"
done
