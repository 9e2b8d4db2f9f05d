import numpy as np
M62=(1<<62)-1
def mix2(x):
    x = x.copy()
    x ^= x >> np.uint64(31); x = (x * np.uint64(0xff51afd7ed558ccd)) & np.uint64(M62)
    x ^= x >> np.uint64(29); x = (x * np.uint64(0xc4ceb9fe1a85ec53)) & np.uint64(M62)
    x ^= x >> np.uint64(32)
    return x
def mix1(x):
    x = x.copy()
    x ^= x >> np.uint64(31); x = (x * np.uint64(0x9E3779B97F4A7C15)) & np.uint64(M62)
    x ^= x >> np.uint64(29)
    return x
rng=np.random.default_rng(1)
def keys_from_seq(seq,k=31):
    # seq: array of codes 0..3 (A=0,T=1,C=2,G=3); enc = plane1<<32 | plane0, bit t = t-th base
    n=len(seq)-k+1
    p0=(seq&1).astype(np.uint64); p1=((seq>>1)&1).astype(np.uint64)
    w0=np.zeros(n,np.uint64); w1=np.zeros(n,np.uint64)
    for t in range(k):
        w0|=p0[t:t+n]<<np.uint64(t); w1|=p1[t:t+n]<<np.uint64(t)
    # rc: reversed planes with plane0 inverted
    r0=np.zeros(n,np.uint64); r1=np.zeros(n,np.uint64)
    for t in range(k):
        r0|=(np.uint64(1)-p0[t:t+n])<<np.uint64(k-1-t); r1|=p1[t:t+n]<<np.uint64(k-1-t)
    fwd=(w1<<np.uint64(32))|w0; rc=(r1<<np.uint64(32))|r0
    return np.minimum(fwd,rc)
def stats(h,name):
    for bits,label in ((8,'bucket'),(16,'partition')):
        c=np.bincount((h>>np.uint64(62-bits)).astype(np.int64),minlength=1<<bits)
        m=c.mean(); print(name,label,'mean %.1f max %d min %d  std/poisson %.3f'%(m,c.max(),c.min(),c.std()/np.sqrt(m)))
    # table slots: 2^24 slots, distinct keys: measure linear-probe cluster via occupancy of 2^20 coarse cells
    c=np.bincount((h>>np.uint64(62-22)).astype(np.int64),minlength=1<<22)
    m=c.mean(); print(name,'2^22 cells mean %.2f max %d std/poisson %.3f'%(m,c.max(),c.std()/np.sqrt(m)))
for gen in ('random','at_rich','repeats'):
    if gen=='random': seq=rng.integers(0,4,4_000_000)
    elif gen=='at_rich': seq=rng.choice(4,4_000_000,p=[0.45,0.45,0.05,0.05])
    else:
        unit=rng.integers(0,4,50_000); seq=np.concatenate([unit]*40+[rng.integers(0,4,2_000_000)])
    k=np.unique(keys_from_seq(seq.astype(np.int64)))
    print(gen,len(k),'distinct')
    stats(mix2(k),'mix2'); stats(mix1(k),'mix1')
