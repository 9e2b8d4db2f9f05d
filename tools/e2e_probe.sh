cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import os,sys,subprocess,time
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
import numpy as np, make_fixtures
n=4_000_000; L=150
base="/dev/shm/e2ep"; os.makedirs(base,exist_ok=True)
seqs,quals=make_fixtures.headline_arrays(1_000_000,L)
for mate in (1,2):
    with open(base+"/r%d.fq"%mate,"wb") as f:
        done=0
        while done<n:
            m=min(500000,n-done)
            recs=[b"@SYN:%09d/%d\n"%(done+i,mate)+seqs[i].tobytes()+b"\n+\n"+quals[i].tobytes()+b"\n" for i in range(m)]
            f.write(b"".join(recs)); done+=m
PY
for mode in fork nofork; do
  for rep in 1 2; do
    rm -rf /dev/shm/e2ep/out
    s=$(date +%s.%N)
    if [ $mode = nofork ]; then export FAQCS_MI_NO_FORK=1; else unset FAQCS_MI_NO_FORK; fi
    FAQCS_MI_TIMING=1 faqcs_amd/faqcs_mi -1 /dev/shm/e2ep/r1.fq -2 /dev/shm/e2ep/r2.fq -d /dev/shm/e2ep/out --ascii 33 -q 5 --min_L 50 --trim_only 2> /dev/shm/e2ep/err.txt
    e=$(date +%s.%N)
    echo "$mode rep $rep wall $(python3 -c "print(round($e - $s, 3))") ; $(grep 'statistics written' /dev/shm/e2ep/err.txt) ; $(grep 'options parsed' /dev/shm/e2ep/err.txt)"
  done
done
s=$(date +%s.%N); faqcs_amd/faqcs_mi --version > /dev/null 2>&1; e=$(date +%s.%N); echo "version-only wall $(python3 -c "print(round($e - $s, 3))")"
rm -rf /dev/shm/e2ep
