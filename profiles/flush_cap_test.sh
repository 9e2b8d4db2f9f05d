#!/bin/bash
# trim_lds's flush rows and the per-block cap, exercised on a small launch: a build that flushes every 24 chunks (-DFAQCS_LDS_TEST_FLUSH_CHUNKS=24:
# 8 rows x 24 chunks x 256 blocks = 3 145 728 reads fill EVERY block's rows to the brim) must give the product build's result and counter hashes;
# one read more and the launch goes to another kernel.   bash profiles/build_variant.sh tflush -DFAQCS_LDS_TEST_FLUSH_CHUNKS=24 ; then on the GPU box:
for n in 3145728 3145408 1000000 3145729; do
  TRIM_AB_REPS=1 ./profiles/microbench/trim_ab $n 150 1 faqcs_amd/libfaqcs_mi.so faqcs_amd/libfaqcs_mi_tflush.so 2>&1 | grep -E "^faqcs"
done
TRIM_AB_TRIM5=3 TRIM_AB_REPS=1 ./profiles/microbench/trim_ab 3145728 150 1 faqcs_amd/libfaqcs_mi.so faqcs_amd/libfaqcs_mi_tflush.so 2>&1 | grep -E "^faqcs"
TRIM_AB_REPS=1 ./profiles/microbench/trim_ab 3145728 100 1 faqcs_amd/libfaqcs_mi.so faqcs_amd/libfaqcs_mi_tflush.so 2>&1 | grep -E "^faqcs"
