"""Regenerates the committed golden fixtures (run in the build container only).

  ksw2_golden.npz    inputs + every ksw_extz_t field + full CIGAR, produced by the REFERENCE'S OWN
                     C file (src/ksw2/csrc/ksw2_extz2_sse.c) compiled into oracle/_ref by oracle/Makefile.
  regions_golden.npz inputs + flat outputs of the per-region path for BASELINE config C1 and a handful
                     of small synthetic regions, produced by the CPU oracle (the Nim reference cannot
                     be built in this image: no nim/hts-nim/kmer) with ksw2 routed through the compiled
                     reference C.

  ksw2_pair_golden.npz  (round 5) batches built for the sweeps that carry the load: groups of 2-9 jobs of equal contig
                     length with windows of at least qlen + w + 1 bases (what k_ksw_plan pairs, ksw_pair.h), band
                     widths 49 / 50 / 57 / 62, z-drops that fire early / in the steady diagonals / in the tail, wildcards
                     in a window; and (read, window suffix, contig suffix) triples at the alignment fallback's settings
                     (gapo 5, unbanded, no z-drop: indelope.nim:318-344) for the two-target sweep (ksw_duo.h).  Every
                     expected field and CIGAR comes from the compiled reference C.

Usage: python tests/golden/make_golden.py [pairs]       (`pairs`: only the round-5 file)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc                      # noqa: E402
from indelope_amd import synth            # noqa: E402
from indelope_amd.host import BatchResult  # noqa: E402
import test_oracle_ksw2 as tk              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
FIELDS = tk.FIELDS


def ksw2():
    o = orc.get()
    assert o.ref_lib() is not None, "needs /root/reference (oracle/_ref)"
    q_all, t_all, q_off, t_off, par, ez, cig, cig_off = [], [], [0], [0], [], [], [], [0]
    for pi, kw in enumerate(tk.PARAMS):
        if kw["flag"] & (0x01 | 0x08 | 0x04):      # score-only / approx / generic: not on the GPU path
            continue
        for q, t in tk.cases(5000 + pi, 40):
            qe, te = o.encode(q), o.encode(t)
            f, c = o.ksw_ref(qe, te, **kw)
            q_all.append(qe); t_all.append(te)
            q_off.append(q_off[-1] + len(qe)); t_off.append(t_off[-1] + len(te))
            par.append([kw["gapo"], kw["gape"], kw["w"], kw["zdrop"], kw["flag"]])
            ez.append([f[k] for k in FIELDS]); cig.append(c); cig_off.append(cig_off[-1] + len(c))
    # contig-shaped cases at production settings (indelope.nim:221): contig vs window + indel
    rng = np.random.default_rng(99)
    kw = tk.PARAMS[0]
    for i in range(60):
        tl = int(rng.integers(300, 1100))
        t = tk.kats.rand_dna(rng, tl)
        q = tk.mutate(rng, t[:tl - int(rng.integers(63, 200))], i % 2)
        qe, te = o.encode(q), o.encode(t)
        f, c = o.ksw_ref(qe, te, **kw)
        q_all.append(qe); t_all.append(te)
        q_off.append(q_off[-1] + len(qe)); t_off.append(t_off[-1] + len(te))
        par.append([kw["gapo"], kw["gape"], kw["w"], kw["zdrop"], kw["flag"]])
        ez.append([f[k] for k in FIELDS]); cig.append(c); cig_off.append(cig_off[-1] + len(c))
    np.savez_compressed(os.path.join(HERE, "ksw2_golden.npz"),
                        q=np.concatenate(q_all), t=np.concatenate(t_all), q_off=np.array(q_off, np.int64),
                        t_off=np.array(t_off, np.int64), params=np.array(par, np.int32),
                        ez=np.array(ez, np.int64), cigar=np.concatenate(cig).astype(np.uint32),
                        cigar_off=np.array(cig_off, np.int64))
    print("ksw2_golden: %d cases" % len(par))


def _diverge(rng, src, where):
    """A query that follows `src` and leaves it for good at `where` (fraction of its length): the running maximum stops
    there and a small z-drop fires some diagonals later -- early, in the steady diagonals or in the tail of the sweep."""
    q = src.copy()
    a = int(len(q) * where)
    q[a:] = (q[a:] + rng.integers(1, 4, len(q) - a)) % 4
    return q


def ksw2_pairs():
    o = orc.get()
    assert o.ref_lib() is not None, "needs /root/reference (oracle/_ref)"
    rng = np.random.default_rng(2025)
    q_all, t_all, q_off, t_off, par, ez, cig, cig_off, grp = [], [], [0], [0], [], [], [], [0], []

    def add(g, qe, te, match, mismatch, gapo, gape, w, zdrop, flag):
        f, c = o.ksw_ref(qe, te, mat=o.matrix(match, mismatch), gapo=gapo, gape=gape, w=w, zdrop=zdrop, flag=flag)
        q_all.append(qe); t_all.append(te)
        q_off.append(q_off[-1] + len(qe)); t_off.append(t_off[-1] + len(te))
        par.append([match, mismatch, gapo, gape, w, zdrop, flag]); grp.append(g)
        ez.append([f[k] for k in FIELDS]); cig.append(c); cig_off.append(cig_off[-1] + len(c))

    # ---- the pair sweep: one group = one ihp_ksw_extz2_batch call, its jobs share (scoring, w, zdrop, flag)
    g = 0
    schemes = [(1, -2, 4, 1), (1, -2, 4, 1), (2, -3, 5, 2), (1, -4, 6, 1)]
    for w in (49, 50, 57, 62):
        for zdrop in (400, 5, 20, 60, -1):
            for flag in ((0, 0x40) if zdrop in (400, 20) else (0,)):            # KSW_EZ_EXTZ_ONLY on two of them
                ma, mi, go, ge = schemes[g % len(schemes)]
                for _ in range(3):                                                # three contig lengths per group
                    ql = int(rng.integers(w + 32, 420))
                    for k in range(int(rng.integers(2, 10))):                     # 2-9 jobs of that length
                        tl = ql + w + 1 + int(rng.integers(0, 160))
                        t = rng.integers(0, 4, tl).astype(np.uint8)
                        src = t[:ql].copy()
                        u = rng.random()
                        if u < 0.25:
                            q = _diverge(rng, src, float(rng.choice([0.05, 0.2, 0.5, 0.8, 0.95])))
                        elif u < 0.6 and ql > 120:                               # the event indelope is after
                            a = int(rng.integers(40, ql - 60)); L = int(rng.integers(5, 45))
                            q = np.concatenate([src[:a], src[a + L:], rng.integers(0, 4, L)]) if rng.random() < 0.5 else \
                                np.concatenate([src[:a], rng.integers(0, 4, L), src[a:]])[:ql]
                        else:
                            q = np.where(rng.random(ql) < float(rng.choice([0, 0.01, 0.05])), (src + rng.integers(1, 4, ql)) % 4, src)
                        q = q[:ql].astype(np.uint8)
                        if rng.random() < 0.08:
                            t = t.copy(); t[int(rng.integers(0, tl))] = 4        # a wildcard in a window: the pair sweep takes it
                        add(g, q, t, ma, mi, go, ge, w, zdrop, flag)
                g += 1
    n_pair = len(par)
    # ---- the two-target sweep of the alignment fallback: a read against a window suffix and against a contig suffix
    # (indelope.nim:336-344); consecutive cases 2i, 2i+1 share the read.  ihp_ksw_duo_batch runs them as one item.
    for i in range(120):
        rl = int(rng.choice([60, 100, 150, 150, 150, 151, 192, 193, 250, 320]))
        hap = rng.integers(0, 4, rl + 400).astype(np.uint8)
        lo = int(rng.integers(0, 60))
        read = hap[lo:lo + rl].copy()
        err = rng.random(rl) < float(rng.choice([0, 0.01, 0.03]))
        read[err] = (read[err] + rng.integers(1, 4, int(err.sum()))) % 4
        L = int(rng.integers(5, 40)); a = lo + int(rng.integers(20, max(21, rl - 20)))
        other = np.concatenate([hap[:a], hap[a + L:]]) if i % 2 else np.concatenate([hap[:a], rng.integers(0, 4, L).astype(np.uint8), hap[a:]])
        t0 = other[max(0, lo - int(rng.integers(0, 5))):][:int(rng.integers(rl + 20, rl + 300))]     # "window suffix"
        t1 = hap[max(0, lo - int(rng.integers(0, 5))):][:int(rng.integers(rl + 20, rl + 300))]       # "contig suffix"
        if i % 17 == 0:
            t0 = t0.copy(); t0[int(rng.integers(0, len(t0)))] = 4
        for t in (t0, t1):
            add(-1, read, np.ascontiguousarray(t, np.uint8), 1, -2, 5, 1, -1, -1, 0)
    np.savez_compressed(os.path.join(HERE, "ksw2_pair_golden.npz"),
                        q=np.concatenate(q_all), t=np.concatenate(t_all), q_off=np.array(q_off, np.int64),
                        t_off=np.array(t_off, np.int64), params=np.array(par, np.int32), group=np.array(grp, np.int32),
                        ez=np.array(ez, np.int64), cigar=np.concatenate(cig).astype(np.uint32),
                        cigar_off=np.array(cig_off, np.int64))
    zd = sum(1 for e in ez[:n_pair] if e[1])
    print("ksw2_pair_golden: %d pair-sweep cases in %d groups (%d with zdropped = 1), %d fallback cases" % (n_pair, g, zd, len(par) - n_pair))


def regions():
    o = orc.get()
    o.use_reference_ksw(True)
    out = {}
    sets = {"c1": synth.config("C1")[0],
            "small": synth.generate(24, read_len=150, n_reads=(8, 40), err_rate=2e-3, config_id=91)[0],
            "long": synth.generate(4, read_len=300, n_reads=(40, 40), err_rate=1e-3, n_events=2, window_len=1400,
                                   event_pos=500, config_id=95)[0],
            # tandem duplications: reads carry both k-mers, the alignment fallback (indelope.nim:312-372) votes
            "dup": synth.generate(10, read_len=150, n_reads=(12, 40), err_rate=1e-3, config_id=97, dup_frac=0.7)[0]}
    for name, b in sets.items():
        K = 31 if name == "long" else 27
        res = o.run_regions(b, o.params(K=K))
        for f in ("region_read_off", "read_off", "bases", "quals", "read_start", "read_stop", "mapq", "read_skip",
                  "ref_off", "ref_bases", "ref_origin"):
            out["%s.in.%s" % (name, f)] = getattr(b, f)
        for f in BatchResult.FIELDS:
            out["%s.out.%s" % (name, f)] = getattr(res, f)
        out["%s.K" % name] = np.int32(K)
        print(name, "regions", b.n_regions, "contigs", res.n_contigs, "events", res.n_events)
    o.use_reference_ksw(False)
    np.savez_compressed(os.path.join(HERE, "regions_golden.npz"), **out)


if __name__ == "__main__":
    if "pairs" in sys.argv[1:]:
        ksw2_pairs()
    else:
        ksw2()
        regions()
        ksw2_pairs()
