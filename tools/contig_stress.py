#!/usr/bin/env python3
"""Randomised parity sweep of the single-step contig procs (slide_align / insert / trim through the C ABI, the byte-based
device code) against the oracle.  usage: tools/contig_stress.py [n_pairs] [seed]"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import indelope_amd  # noqa: E402
import kats  # noqa: E402
import oracle  # noqa: E402
from indelope_amd.host import Contig  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
hip = indelope_amd.api()
hip.init(0)
orc = oracle.get()
bad = 0
for i in range(n):
    ln = int(rng.integers(17, 900))
    base = kats.rand_dna(rng, ln + 400)
    o = int(rng.integers(0, 250))
    t = Contig(base[50:50 + ln], 100, int(rng.integers(1, 9)))
    ql = int(rng.integers(1, 400))
    s0 = 50 + o - (int(rng.integers(0, 120)) if rng.random() < 0.5 else 0)
    qs = list(base[max(0, s0):max(0, s0) + ql])
    if not qs:
        continue
    for p in rng.integers(0, len(qs), rng.integers(0, 5)):
        qs[p] = "ACGT"[rng.integers(0, 4)]
    q = Contig("".join(qs), 7, int(rng.integers(1, 9)))
    for c in (q, t):
        hi = int(rng.choice([2, 4, 12, 300]))
        c._sup[:c.len] = rng.integers(1, hi + 1, c.len)
        c.nreads = int(rng.integers(1, 60))
    mo, mm, rule = int(rng.integers(1, 80)), int(rng.integers(0, 3)), int(rng.integers(0, 2))
    a = hip.slide_align(q, t, min_overlap=mo, max_mismatch=mm, allowed=rule)
    e = orc.slide_align(q, t, min_overlap=mo, max_mismatch=mm, allowed=rule)
    ok = (a.offset, a.corrections) == (e.offset, e.corrections) and (not e.aligned or (a.matches, a.mismatches) == (e.matches, e.mismatches))
    if ok and e.aligned:
        t1, q1, t2, q2 = copy.deepcopy(t), copy.deepcopy(q), copy.deepcopy(t), copy.deepcopy(q)
        hip.insert(t1, q1, a)
        orc.insert(t2, q2, e)
        ok = (t1.sequence, t1.support, t1.start, t1.nreads) == (t2.sequence, t2.support, t2.start, t2.nreads) and (q1.sequence, q1.support) == (q2.sequence, q2.support)
    if ok:
        ms = int(rng.integers(0, 6))
        t3, t4 = copy.deepcopy(t), copy.deepcopy(t)
        hip.trim(t3, ms)
        orc.trim(t4, ms)
        ok = (t3.sequence, t3.support, t3.start) == (t4.sequence, t4.support, t4.start)
    if not ok:
        bad += 1
        if bad < 6:
            print("DIFF", i, "tlen", t.len, "qlen", q.len, "mo", mo, "mm", mm, "rule", rule, a, e)
print("done: %d pairs, %d differences" % (n, bad))
sys.exit(1 if bad else 0)
