#!/usr/bin/env python3
"""Host threads running different batches with different parameters at the same time (the hints one batch leaves for the next,
the pools and the stream cache are process-wide), every result compared with the oracle.
usage: tools/thread_stress.py [threads] [configs per thread] [seed] [thread:config | -] [key=value ...]"""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelope_amd  # noqa: E402
import oracle  # noqa: E402
from indelope_amd import synth  # noqa: E402
from indelope_amd.host import BatchResult  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hip = indelope_amd.api()
hip.init(0)
for kv in sys.argv[5:]:                                      # library switches: key=value (ihp_debug_set)
    k_, v_ = kv.split("=")
    hip.debug_set(**{k_: int(v_)})
orc = oracle.get()
bad = []
lock = threading.Lock()


ONLY = tuple(int(x) for x in sys.argv[4].split(":")) if len(sys.argv) > 4 and sys.argv[4] != "-" else None     # "thread:config": replay that one alone


def worker(k):
    rng = np.random.default_rng(seed * 100 + k)
    for it in range(N):
        rl = int(rng.choice([100, 150, 151, 250, 300]))
        lo = int(rng.integers(2, 60))
        hi = int(rng.integers(lo, min(280, lo + rng.choice([10, 60, 200]))))
        cfg = dict(n_regions=int(rng.integers(5, 400)), read_len=rl, n_reads=(lo, hi), err_rate=float(rng.choice([0, 1e-3, 1e-2])),
                   config_id=5000 + 100 * k + it, dup_frac=float(rng.choice([0, 0.3])), seed=int(rng.integers(1, 2**31)))
        b, _ = synth.generate(**cfg)
        kw = dict(K=int(rng.choice([21, 27, 31])))
        if rng.random() < 0.3:
            kw.update(max_mismatch=1)
        if rng.random() < 0.3:
            kw.update(min_overlap_pct=float(rng.choice([0.5, 1.0])))
        if rng.random() < 0.3:
            kw.update(bw=int(rng.choice([20, 48, 62, -1])), zdrop=int(rng.choice([-1, 50, 1000])))
        if rng.random() < 0.3:
            kw.update(combine_min_support=int(rng.choice([1, 2, 4])), combine_min_overlap=int(rng.choice([17, 30, 80])))
        if rng.random() < 0.5:
            b = b.with_trim_bounds()
        mode = int(rng.integers(0, 3))
        plan = (bool(rng.integers(0, 2)), int(rng.integers(1, 4)), [bool(rng.random() < 0.5) for _ in range(3)])
        if ONLY and (k, it) != ONLY:
            continue
        if ONLY:
            print("replay", k, it, cfg, kw, "mode", mode, plan, "trim", b.trim_lo is not None, flush=True)
        if mode == 0:
            got = hip.run_regions(b, hip.params(**kw))
        else:                                                   # the batch API: several runs, eager fetch, a slab now and then
            slab = hip.make_slab(b.with_trim_bounds()) if mode == 2 and not (b.bases >= 97).any() else None
            h = hip.batch_upload_slab(slab, hip.params(**kw)) if slab else hip.batch_upload(b, hip.params(**kw))
            try:
                hip.batch_set_fetch(h, eager=plan[0])
                for j in range(plan[1]):
                    hip.batch_run(h)
                    if plan[2][j]:
                        hip.batch_sync(h)
                got = hip.batch_fetch(h)
            finally:
                hip.batch_free(h)
                if slab:
                    slab.free()
        exp = orc.run_regions(b, orc.params(**kw))
        d = BatchResult.first_difference(got, exp)
        if d is not None:
            with lock:
                bad.append((k, it, cfg, kw, d))
                print("DIFF thread", k, it, cfg, kw, d, flush=True)


if ONLY:
    for rep in range(5):
        worker(ONLY[0])
    print("replayed 5 times, %d differences" % len(bad))
    sys.exit(1 if bad else 0)
th = [threading.Thread(target=worker, args=(k,)) for k in range(T)]
for x in th:
    x.start()
for x in th:
    x.join()
print("done: %d threads x %d configs, %d differences" % (T, N, len(bad)))
sys.exit(1 if bad else 0)
