#!/usr/bin/env python3
"""Randomised parity sweep of the ROI evidence scan (row f4) against the oracle.  usage: tools/roi_stress.py [n] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import indelope_amd  # noqa: E402
import oracle  # noqa: E402
from test_roi import random_reads  # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
hip = indelope_amd.api()
hip.init(0)
orc = oracle.get()
bad = 0
for it in range(n_it):
    n = int(rng.choice([1, 5, 60, 400, 3000, 12000]))
    span = int(rng.choice([400, 3000, 20000, 150000]))
    gap = int(rng.choice([0, 0, 3, 25, 400]))
    hot = sorted(rng.integers(100, max(101, span - 100), int(rng.integers(0, 8))).tolist())
    st, en, cg, skip = random_reads(rng, n, max(span, 450), gap_every=gap, hot=hot, skip_frac=float(rng.choice([0, 0.05, 0.5])),
                                    event_frac=float(rng.choice([0, 0.3, 0.9])))
    kw = dict(read_skip=skip if rng.random() < 0.8 else None, origin=int(rng.choice([0, 12345, 3_000_000_000])), span=int(en.max() + rng.integers(0, 100)),
              min_event_support=int(rng.integers(0, 7)), min_read_coverage=int(rng.integers(0, 6)), max_read_coverage=int(rng.choice([1, 30, 600, 100000])))
    o = kw["origin"]
    exp = orc.gen_roi(st + o, en + o, cg, **kw)
    got = hip.gen_roi(st + o, en + o, cg, **kw)
    if got != exp:
        bad += 1
        if bad < 8:
            k2 = {k: v for k, v in kw.items() if k != "read_skip"}
            first = next((i for i in range(min(len(got), len(exp))) if got[i] != exp[i]), None)
            print("DIFF", it, n, span, gap, k2, "skip" if kw["read_skip"] is not None else "noskip", len(got), len(exp),
                  None if first is None else (first, got[first][:2], exp[first][:2], len(got[first][2]), len(exp[first][2]),
                                              [x for x in got[first][2] if x not in exp[first][2]][:5], [x for x in exp[first][2] if x not in got[first][2]][:5]))
print("done: %d scans, %d differences" % (n_it, bad))
sys.exit(1 if bad else 0)
