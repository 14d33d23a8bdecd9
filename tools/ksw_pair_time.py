"""Wall time of ihp_ksw_extz2_batch on synthetic jobs: how the pair sweep's cost depends on how often the running maximum moves.
    python tools/ksw_pair_time.py [n_jobs] [qlen]
"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import indelope_amd

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    ql = int(sys.argv[2]) if len(sys.argv) > 2 else 247
    hip = indelope_amd.api(); hip.init(0)
    rng = np.random.default_rng(1)
    only = sys.argv[3] if len(sys.argv) > 3 else None
    bw = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    for name, sub in (("identical", 0.0), ("subst1", 0.01), ("unrelated", 1.0)):
        if only and name != only:
            continue
        qs, ts = [], []
        for i in range(n):
            t = rng.integers(0, 4, ql + 80 + max(0, bw - 50)).astype(np.uint8)
            q = t[:ql].copy() if sub < 1 else rng.integers(0, 4, ql).astype(np.uint8)
            if 0 < sub < 1:
                m = rng.random(ql) < sub
                q[m] = (q[m] + 1) % 4
            qs.append(q); ts.append(t)
        kw = dict(match=1, mismatch=-2, gap_open=4, gap_ext=1, bw=bw, z=400, flag=0, encoded=True)
        for pair in (1, 0):
            hip.debug_set(ksw_pair=pair)
            hip.align_batch(qs[:2000], ts[:2000], **kw)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); hip.align_batch(qs, ts, **kw); best = min(best, time.perf_counter() - t0)
            print(f"{name:18s} pair={pair}  {best*1e3:8.2f} ms wall for {n} jobs ({hip.b.debug_last_ksw_pairs()} pairs)", flush=True)
        hip.debug_set()

if __name__ == '__main__':
    main()
