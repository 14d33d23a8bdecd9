import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import indelope_amd, oracle
from indelope_amd import _abi as A
import test_gpu_round2 as T
hip=indelope_amd.api(); hip.init(0); orc=oracle.get()
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
bad=0; n=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    w=int(rng.integers(0,63)) if rng.random()<0.6 else int(rng.choice([0,1,2,15,16,17,31,32,33,46,47,48,49,50,61,62,63,64,80,-1])); z=int(rng.choice([-1,5,20,60,200,400,1000])); flag=int(rng.choice([0,A.KSW_EZ_RIGHT,A.KSW_EZ_EXTZ_ONLY,A.KSW_EZ_RIGHT|A.KSW_EZ_REV_CIGAR]))
    go=int(rng.integers(2,9)); ge=int(rng.integers(1,4)); ma=int(rng.integers(1,4)); mi=-int(rng.integers(1,6))
    qs,ts=[],[]
    for _ in range(200):
        ql=int(rng.integers(1,800)); tl=int(rng.integers(1,800))
        q,t=T._pair(rng, ql if rng.random()<0.6 else min(ql,tl), tl, sub=float(rng.choice([0,0.01,0.05,0.2])), indel=float(rng.choice([0,0.01,0.05])), shift=int(rng.integers(0,max(1,tl//2))) if rng.random()<0.5 else None)
        qs.append(q); ts.append(t)
    kw=dict(match=ma, mismatch=mi, gap_open=go, gap_ext=ge, bw=w, z=z, flag=flag)
    ez,cg=hip.align_batch(qs,ts,**kw); ez2,cg2=orc.align_batch(qs,ts,**kw)
    for i in range(len(qs)):
        n+=1
        if ez[i].tolist()!=ez2[i].tolist() or cg[i].tolist()!=cg2[i].tolist():
            bad+=1
            if bad<5: print('DIFF',kw,len(qs[i]),len(ts[i]),ez[i],ez2[i])
    print(it, kw, 'mode', hip.b.debug_last_ksw_mode(), 'bad', bad, flush=True)
print('done', n, 'pairs', bad, 'differences')
sys.exit(1 if bad else 0)
