mkdir -p gpurun_out/r3g
TRIM_AB_REPS=1 ./profiles/microbench/trim_ab 16777216 150 1 faqcs_amd/libfaqcs_mi_blog.so > gpurun_out/r3g/blocklog.txt 2>&1
wc -l gpurun_out/r3g/blocklog.txt
