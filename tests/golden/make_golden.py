"""Regenerates the committed golden fixtures (run in the build container only).

  ksw2_golden.npz    inputs + every ksw_extz_t field + full CIGAR, produced by the REFERENCE'S OWN
                     C file (src/ksw2/csrc/ksw2_extz2_sse.c) compiled into oracle/_ref by oracle/Makefile.
  regions_golden.npz inputs + flat outputs of the per-region path for BASELINE config C1 and a handful
                     of small synthetic regions, produced by the CPU oracle (the Nim reference cannot
                     be built in this image: no nim/hts-nim/kmer) with ksw2 routed through the compiled
                     reference C.

Usage: python tests/golden/make_golden.py
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
    ksw2()
    regions()
