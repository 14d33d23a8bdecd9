"""The alignment fallback's two-target sweep (ksw_duo.h) against the oracle and against the one-at-a-time sweep (ksw_wide.h).

Batches with tandem duplications send events to k_fallback; read lengths (60-321: three slots up to 192, five up to 320, the ring
sweep beyond), read counts, error rates, K and the fallback's
scoring vary per batch (inside and outside what ksw_duo_ok() takes).  Every field of every region is compared three ways:
duo sweep vs oracle, ihp_debug_set("fb_duo", 0) vs oracle.
    python tools/fb_stress.py [seed] [batches]
"""
import sys
import numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import indelope_amd, oracle
from indelope_amd import synth
from indelope_amd.host import BatchResult


def assert_same(got, exp):
    d = BatchResult.first_difference(got, exp)
    assert d is None, d


def main():
    hip = indelope_amd.api(); hip.init(0); orc = oracle.get()
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    bad = 0
    for it in range(nb):
        rl = int(rng.choice([60, 100, 150, 150, 150, 180, 192, 193, 250, 300, 320, 321]))
        K = int(rng.choice([21, 27, 31]))
        kw = dict(n_regions=int(rng.integers(20, 60)), read_len=rl, n_reads=(int(rng.integers(4, 24)), int(rng.integers(24, 80))),
                  err_rate=float(rng.choice([0, 1e-3, 5e-3, 2e-2])), config_id=int(rng.integers(100, 1 << 20)), K=K, dup_frac=float(rng.choice([0.3, 0.6, 0.9])))
        if rl > 150:
            kw["window_len"] = int(rl * 3 + 10)
        b, _ = synth.generate(**kw)
        pk = dict(K=K)
        u = rng.random()
        if u < 0.3:
            pk.update(fb_match=int(rng.integers(1, 4)), fb_mismatch=-int(rng.integers(1, 6)), fb_gap_open=int(rng.integers(2, 9)), fb_gap_ext=int(rng.integers(1, 4)))
        elif u < 0.4:
            pk.update(fb_flag=int(rng.choice([0x40, 0x80, 0xc0])))
        want = orc.run_regions(b, orc.params(**pk))
        fe = int(np.count_nonzero(want.events["fallback_needed"]))
        for duo in (1, 0):
            hip.debug_set(fb_duo=duo)
            got = hip.run_regions(b, hip.params(**pk))
            try:
                assert_same(got, want)
            except AssertionError as ex:
                bad += 1
                print("DIFF duo=%d" % duo, kw, pk, str(ex)[:400])
        hip.debug_set()
        print(it, kw, pk, "fallback events", fe, "bad", bad, flush=True)
    print("done", nb, "batches,", bad, "differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
