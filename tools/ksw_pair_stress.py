"""The pair sweep of ksw2 (ksw_pair.h: two alignments of equal contig length per wavefront) against the oracle.

Batches are built so that k_ksw_plan finds partners: a handful of contig lengths, several jobs of each, windows mostly longer
than qlen + w (what a pair needs) and sometimes shorter (those jobs take the single sweep), band widths 49..62, z-drops that
fire in every phase, scoring schemes inside ksw_narrow_ok().  Every field of every record and every CIGAR is compared; the
same batch is also run with ihp_debug_set("ksw_pair", 0) and must give the same bytes.
    python tools/ksw_pair_stress.py [seed] [batches]
"""
import sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import indelope_amd, oracle
from indelope_amd import _abi as A
import test_gpu_round2 as T


def batch(rng, w, n_len=6, per_len=(1, 9)):
    qs, ts = [], []
    for _ in range(n_len):
        ql = int(rng.integers(w + 32, 700)) if rng.random() < 0.85 else int(rng.integers(1, w + 40))
        for _ in range(int(rng.integers(*per_len))):
            u = rng.random()
            tl = ql + w + 1 + int(rng.integers(0, 200)) if u < 0.8 else ql + int(rng.integers(0, w + 2)) if u < 0.9 else int(rng.integers(1, 800))
            t = rng.integers(0, 4, tl)
            sub = float(rng.choice([0, 0.01, 0.05, 0.2, 0.6])); indel = float(rng.choice([0, 0.01, 0.03]))
            lo = int(rng.integers(0, max(1, tl - ql))) if rng.random() < 0.4 else 0
            src = t[lo:lo + ql]
            q = np.where(rng.random(len(src)) < sub, (src + rng.integers(1, 4, len(src))) % 4, src)
            if indel and len(q) > 40:
                for _ in range(int(indel * len(q)) + 1):
                    if len(q) <= 40:
                        break
                    a = int(rng.integers(10, len(q) - 20)); L = int(rng.integers(1, 30))
                    if rng.random() < 0.5:
                        q = np.concatenate([q[:a], rng.integers(0, 4, L), q[a:]])
                    else:
                        q = np.concatenate([q[:a], q[a + L:]])
            # the contig length of the group is what the plan pairs on: cut or pad to ql
            if len(q) >= ql:
                q = q[:ql]
            else:
                q = np.concatenate([q, rng.integers(0, 4, ql - len(q))])
            if rng.random() < 0.03:
                t = t.copy(); t[int(rng.integers(0, tl))] = 4                 # a wildcard in the window is the pair sweep's too
            if rng.random() < 0.02:
                q = q.copy(); q[int(rng.integers(0, ql))] = 4                 # one in the contig is not
            qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    order = rng.permutation(len(qs))
    return [qs[i] for i in order], [ts[i] for i in order]


def main():
    hip = indelope_amd.api(); hip.init(0); orc = oracle.get()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = n = tot_pairs = 0
    for it in range(nb):
        w = int(rng.integers(49, 63)) if rng.random() < 0.9 else int(rng.choice([40, 48, 63, 70]))
        z = int(rng.choice([-1, 5, 20, 60, 200, 400, 1000]))
        flag = int(rng.choice([0, 0, 0, A.KSW_EZ_EXTZ_ONLY, A.KSW_EZ_REV_CIGAR]))
        go = int(rng.integers(2, 9)); ge = int(rng.integers(1, 4)); ma = int(rng.integers(1, 4)); mi = -int(rng.integers(1, 6))
        if rng.random() < 0.5:
            go, ge, ma, mi = 4, 1, 1, -2
        qs, ts = batch(rng, max(w, 1))
        kw = dict(match=ma, mismatch=mi, gap_open=go, gap_ext=ge, bw=w, z=z, flag=flag, encoded=True)
        ez, cg = hip.align_batch(qs, ts, **kw)
        npairs = hip.b.debug_last_ksw_pairs(); tot_pairs += npairs
        ez2, cg2 = orc.align_batch(qs, ts, **kw)
        hip.debug_set(ksw_pair=0)
        ez3, cg3 = hip.align_batch(qs, ts, **kw)
        hip.debug_set()
        for i in range(len(qs)):
            n += 1
            d1 = ez[i].tolist() != ez2[i].tolist() or cg[i].tolist() != cg2[i].tolist()
            d2 = ez3[i].tolist() != ez2[i].tolist() or cg3[i].tolist() != cg2[i].tolist()
            if d1 or d2:
                bad += 1
                if bad < 12:
                    print('DIFF', 'pair' if d1 else '', 'single' if d2 else '', kw, len(qs[i]), len(ts[i]), '\n  hip ', ez[i], '\n  orc ', ez2[i],
                          '\n  cig ', cg[i].tolist()[:12], '\n  ocg ', cg2[i].tolist()[:12])
        print(it, {k: v for k, v in kw.items() if k != 'encoded'}, len(qs), 'jobs,', npairs, 'pairs; bad', bad, flush=True)
    print('done', n, 'alignments,', tot_pairs, 'pairs,', bad, 'differences')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
