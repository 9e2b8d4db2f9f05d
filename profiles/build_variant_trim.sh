#!/bin/bash
# Like build_variant.sh, for faqcs_trim_kernel.hip: bash profiles/build_variant_trim.sh <name> -DFOO ...  -> profiles/microbench/libfaqcs_mi_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
cs=faqcs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $cs/faqcs_trim_kernel.hip -o /tmp/faqcs_trimk_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/libfaqcs_mi_$name.so $cs/faqcs_capi.o /tmp/faqcs_trimk_$name.o $cs/faqcs_trim_lds_kernel.o $cs/faqcs_trim_long_kernel.o $cs/faqcs_adapter_kernel.o $cs/faqcs_kmer_kernel.o $cs/faqcs_kmer_skm_kernel.o $cs/faqcs_synth_kernel.o
