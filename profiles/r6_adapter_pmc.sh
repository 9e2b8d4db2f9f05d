#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r6f
bash profiles/pmc_adapter.sh r6f/pmc_pair > gpurun_out/r6f/pmc_adapter_pair.txt 2>&1
FAQCS_ADAPTER_PAIR=0 bash profiles/pmc_adapter.sh r6f/pmc_single > gpurun_out/r6f/pmc_adapter_single.txt 2>&1
echo done
