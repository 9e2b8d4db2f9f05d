for i in 1 2 3; do
for v in old new; do
  lib=$PWD/faqcs_amd/libfaqcs_mi.so; [ $v = old ] && lib=$PWD/profiles/microbench/libfaqcs_mi_old.so
  FAQCS_MI_LIB=$lib python3 bench.py --no-cpu-baseline --e2e-pairs 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
