import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import indelope_amd, oracle, kats
import test_oracle_ksw2 as tk
hip = indelope_amd.api(); hip.init(0); o = oracle.get()
def cmp(qs, ts, **kw):
    a, ca = hip.align_batch(qs, ts, **kw); b, cb = o.align_batch(qs, ts, **kw)
    bad = 0
    for i in range(len(qs)):
        if a[i].tolist() != b[i].tolist() or ca[i].tolist() != cb[i].tolist():
            bad += 1
            if bad <= 4:
                print("case", i, len(qs[i]), len(ts[i]), kw); print("  gpu", a[i], hip_cig(ca[i])); print("  cpu", b[i], hip_cig(cb[i]))
    print("bad", bad, "of", len(qs))
def hip_cig(c): return "".join("%d%s" % (x >> 4, "MID"[x & 15]) for x in c.tolist())
cmp([kats.KSW_QRY], [kats.KSW_TGT], gap_open=4, gap_ext=1, bw=50, z=400, flag=0)
pairs = list(tk.cases(300, 200))
cmp([q for q, t in pairs], [t for q, t in pairs], gap_open=4, gap_ext=1, bw=50, z=400, flag=0)
# one pair per call (no persistent-loop reuse) vs the batch
bad = 0
for i, (q, t) in enumerate(pairs[:60]):
    a, ca = hip.align_batch([q], [t], gap_open=4, gap_ext=1, bw=50, z=400, flag=0)
    b, cb = o.align_batch([q], [t], gap_open=4, gap_ext=1, bw=50, z=400, flag=0)
    if a[0].tolist() != b[0].tolist() or ca[0].tolist() != cb[0].tolist():
        bad += 1
print("single-call bad", bad, "of 60")
