#!/bin/bash
# Diagnostic builds of the library with trim_lds compiled under extra -D switches (results are WRONG with the FAQCS_LDS_NO_*
# switches: they exist to attribute LDS time): bash profiles/build_variant.sh <name> -DFOO [-DBAR ...]  -> profiles/microbench/libfaqcs_mi_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
cs=faqcs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function "$@" -c $cs/faqcs_trim_lds_kernel.hip -o /tmp/faqcs_lds_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/libfaqcs_mi_$name.so $cs/faqcs_capi.o $cs/faqcs_trim_kernel.o $cs/faqcs_trim_long_kernel.o $cs/faqcs_adapter_kernel.o $cs/faqcs_kmer_kernel.o $cs/faqcs_kmer_skm_kernel.o $cs/faqcs_synth_kernel.o /tmp/faqcs_lds_$name.o
