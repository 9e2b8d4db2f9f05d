#!/bin/bash
# Builds profiles/microbench/libfaqcs_mi_stamps.so: the product library with trim_lds compiled -DFAQCS_LDS_STAMPS (section clocks,
# read with tools/stamps.py through FAQCS_MI_LIB).  Diagnostic only; run after __graft_entry__.build().
set -e
cd "$(dirname "$0")/.."
cs=faqcs_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -DFAQCS_LDS_STAMPS -c $cs/faqcs_trim_lds_kernel.hip -o /tmp/faqcs_lds_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/microbench/libfaqcs_mi_stamps.so $cs/faqcs_capi.o $cs/faqcs_trim_kernel.o $cs/faqcs_trim_long_kernel.o $cs/faqcs_adapter_kernel.o $cs/faqcs_kmer_kernel.o $cs/faqcs_kmer_skm_kernel.o $cs/faqcs_synth_kernel.o /tmp/faqcs_lds_stamps.o
