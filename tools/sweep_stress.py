#!/usr/bin/env python3
"""Randomised end-to-end sweep (decoded reads -> ROI scan -> batches -> assemble / ksw2 / tally -> filters -> VCF lines) on
synthetic chromosomes with planted indels: device against oracle, line for line, for several flush sizes.
usage: tools/sweep_stress.py [n] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import indelope_amd  # noqa: E402
import oracle  # noqa: E402
from indelope_amd import sweep  # noqa: E402
from test_sweep import make_target  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip = indelope_amd.api()
hip.init(0)
orc = oracle.get()
bad = 0
for it in range(n):
    kw = dict(seed=int(rng.integers(1, 1 << 30)), length=int(rng.choice([20_000, 60_000, 150_000])), every=int(rng.choice([1500, 4000, 9000])),
              n_per_site=int(rng.choice([8, 40, 120])), read_len=int(rng.choice([100, 150, 250])))
    ref, reads, cigars, truth = make_target(**kw)
    pk = dict(min_reads=3, min_ctg_len=73)
    if rng.random() < 0.3:
        pk.update(K=int(rng.choice([21, 31])))
    out = []
    for api in (orc, hip):
        for br in (10_000, int(rng.integers(1, 9))):
            lines, rois = sweep.call_target(api, reads, cigars, lambda a, b: ref[a:b], api.params(**pk), batch_regions=br, target_len=len(ref))
            out.append((lines, rois))
    ok = all(o == out[0] for o in out[1:])
    print(it, "ok " if ok else "DIFF", kw, pk, "lines", len(out[0][0]), "rois", len(out[0][1]), "sites", len(truth), flush=True)
    bad += not ok
print("done: %d targets, %d differences" % (n, bad))
sys.exit(1 if bad else 0)
