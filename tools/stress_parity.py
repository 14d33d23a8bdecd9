#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): many synthetic configurations through the HIP path and the
oracle, full result comparison.  usage: tools/stress_parity.py [n_configs] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import indelope_amd  # noqa: E402
import oracle  # noqa: E402
from indelope_amd import synth  # noqa: E402
from indelope_amd.host import BatchResult  # noqa: E402


WIDE_PARAMS = False
LONG_SHORT = None
ONLY = None
DEEP = False


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    hip = indelope_amd.api()
    hip.init(0)
    global WIDE_PARAMS
    for kv in sys.argv[3:]:                                      # library switches: key=value (ihp_debug_set); "params" = randomise ihp_params too
        if kv == "params":
            WIDE_PARAMS = True
            continue
        if kv == "lengths":                                      # very short and very long reads (the packed path takes 20 .. 960 bases)
            global LONG_SHORT
            LONG_SHORT = [20, 25, 36, 50, 400, 700, 959, 960, 961, 1200, 1500]
            continue
        if kv == "deep":                                         # regions of 200 .. 700 reads (gen_roi hands over up to 600: the wide combine
            global DEEP                                          # build; above 640 the byte-based passes), few regions per configuration
            DEEP = True
            continue
        if kv.startswith("only="):
            global ONLY
            ONLY = int(kv[5:])
            continue
        k, v = kv.split("=")
        hip.debug_set(**{k: int(v)})
    orc = oracle.get()
    bad = 0
    errs = 0
    t0 = time.time()
    for it in range(n):
        rl = int(rng.choice(LONG_SHORT if LONG_SHORT else [75, 100, 125, 150, 151, 200, 250, 300]))
        K = int(rng.choice([21, 25, 27, 31])) if rl >= 100 else 21
        lo = int(rng.integers(2, 40))
        hi = int(rng.integers(lo, min(300, lo + rng.choice([10, 60, 250]))))
        cfg = dict(n_regions=int(rng.integers(5, 120)), read_len=rl, n_reads=(lo, hi), err_rate=float(rng.choice([0, 1e-3, 3e-3, 1e-2])),
                   config_id=1000 + it, dup_frac=float(rng.choice([0, 0.2, 0.6])), seed=int(rng.integers(1, 2**31)))
        if rl >= 250 and rng.random() < 0.5:
            cfg.update(n_events=2, window_len=1400, event_pos=500)
        if DEEP:
            lo = int(rng.choice([200, 250, 257, 300, 500, 600, 641]))
            cfg.update(n_regions=int(rng.integers(3, 24)), read_len=int(rng.choice([75, 100, 150, 150, 250])), n_reads=(lo, int(lo + rng.choice([0, 40, 100]))),
                       err_rate=float(rng.choice([0, 1e-4, 2.5e-4, 1e-3, 3e-3])))
            cfg.pop("n_events", None); cfg.pop("window_len", None); cfg.pop("event_pos", None)
            K = 21 if cfg["read_len"] < 100 else K
        b, _ = synth.generate(**cfg)
        if DEEP and rng.random() < 0.5:                              # ordinary regions in front of and behind the deep ones
            from indelope_amd.host import concat_batches
            o1, _ = synth.generate(n_regions=int(rng.integers(1, 40)), read_len=cfg["read_len"], n_reads=(8, 120), err_rate=1e-3, config_id=5000 + it)
            b = concat_batches([o1, b, o1.slice(0, max(1, o1.n_regions // 2))])
        kw = dict(K=K)
        if os.environ.get("IHP_STRESS_NO_FALLBACK"):                # (replaying a refusal: does it come from the fallback's scratch?)
            kw["fallback"] = 0
        if rng.random() < 0.5:
            kw.update(min_reads=3, min_ctg_len=73)
        if WIDE_PARAMS and rng.random() < 0.7:                     # the parameters the CLI leaves at their defaults, too
            if rng.random() < 0.4: kw.update(bw=int(rng.choice([0, 8, 20, 47, 48, 49, 62, 70, -1])))
            if rng.random() < 0.3: kw.update(zdrop=int(rng.choice([-1, 20, 100, 1000])))
            if rng.random() < 0.3: kw.update(min_overlap_pct=float(rng.choice([0.4, 0.5, 0.8, 1.0])))
            if rng.random() < 0.2: kw.update(max_mismatch=int(rng.choice([1, 2])))
            if rng.random() < 0.3: kw.update(combine_min_support=int(rng.choice([1, 2, 4])))
            if rng.random() < 0.3: kw.update(combine_min_overlap=int(rng.choice([10, 17, 30, 80])))
            if rng.random() < 0.2: kw.update(gap_open=int(rng.choice([2, 6])), gap_ext=int(rng.choice([1, 2])), mismatch=int(rng.choice([-1, -4])))
            if rng.random() < 0.2: kw.update(ref_pad=int(rng.choice([0, 10, 60])))
            if rng.random() < 0.2: kw.update(max_pre_contigs=int(rng.choice([2, 5, 50])), max_events=int(rng.choice([1, 3])))
            if rng.random() < 0.2: kw.update(fallback=0)
            if rng.random() < 0.2: kw.update(ksw_flag=2)
            if rng.random() < 0.2: kw.update(min_mapq_assemble=int(rng.choice([0, 10, 30])), min_mapq_stop=int(rng.choice([0, 20])), min_mapq_tally=int(rng.choice([0, 10, 30])))
            if rng.random() < 0.2: kw.update(trim_min_qual=int(rng.choice([0, 3, 20])))
            if rng.random() < 0.2: kw.update(min_event_len=int(rng.choice([1, 3, 10])))
            if rng.random() < 0.2: kw.update(K=int(rng.choice([9, 13, 17, 31])))
        if rng.random() < 0.3:
            b.mapq = rng.choice(np.array([0, 5, 9, 10, 19, 20, 60], np.uint8), b.n_reads)
        if rng.random() < 0.4:
            q = b.quals.copy()
            hit = rng.random(len(q)) < 0.02
            q[hit] = 2
            for i in range(0, b.n_reads, 7):
                q[b.read_off[i]:b.read_off[i] + int(rng.integers(0, 40))] = 2
            b.quals = q
        if rng.random() < 0.5:
            b = b.with_trim_bounds()
        if rng.random() < 0.2:
            bases = b.bases.copy()
            hit = rng.random(len(bases)) < 0.003
            bases[hit] = rng.choice(np.frombuffer(b"Nacgt", np.uint8), int(hit.sum()))
            b.bases = bases
        if ONLY is not None and it != ONLY:
            continue
        if WIDE_PARAMS:
            print("    next", it, cfg, {k: v for k, v in kw.items()}, flush=True)
        try:
            got = hip.run_regions(b, hip.params(**kw))
            if DEEP or it % 4 == 0:                                  # the same results through the compact fetch (4-bit bases, byte supports + escapes)
                h = hip.batch_upload(b, hip.params(**kw))
                hip.batch_set_fetch(h, compact=True)
                hip.batch_run(h)
                g2 = hip.batch_fetch(h)
                hip.batch_free(h)
                d2 = BatchResult.first_difference(g2, got)
                if d2 is not None:
                    print("%3d DIFF compact fetch: %s" % (it, d2), flush=True)
                    bad += 1
        except Exception as e:                                   # an honest refusal (IHP_E_CAPACITY ...) is reported, not compared
            print("%3d ERR %s / %s  rl=%d reads=%s regions=%d %s" % (it, e, hip.b.last_hip_error().decode(), rl, cfg["n_reads"], b.n_regions, kw), flush=True)
            errs += 1
            continue
        exp = orc.run_regions(b, orc.params(**kw))
        d = BatchResult.first_difference(got, exp)
        ok = d is None and np.allclose(got.events["gl"], exp.events["gl"], rtol=1e-12)
        vg, ve = hip.call_variants(b, got, hip.params(**kw)), orc.call_variants(b, exp, orc.params(**kw))
        okv = [(x["filter"], x["start"], x["ref"], x["alt"], x["line"]) for x in vg] == [(x["filter"], x["start"], x["ref"], x["alt"], x["line"]) for x in ve]
        if WIDE_PARAMS:
            print("    params", {k: v for k, v in kw.items() if k != "K"})
        print("%3d %s rl=%d K=%d reads=%s err=%g dup=%g regions=%d contigs=%d events=%d fallback=%d variants=%d"
              % (it, "ok " if ok and okv else "DIFF", rl, K, cfg["n_reads"], cfg["err_rate"], cfg["dup_frac"], b.n_regions,
                 got.n_contigs, got.n_events, int((got.events["aligned"] == 1).sum()), sum(x["filter"] == 0 for x in vg)), d or "", flush=True)
        if not (ok and okv):
            print("    cfg", cfg, kw, flush=True)
            if ONLY is not None:                                  # once more through the batch calls, with the library's own account of the run
                hip.debug_set(verbose=1)
                h = hip.batch_upload(b, hip.params(**kw))
                hip.batch_run(h)
                hip.batch_sync(h)
                g2 = hip.batch_fetch(h)
                print("    again:", BatchResult.first_difference(g2, exp), "profile", hip.batch_profile(h)[[21, 23, 24, 25, 26, 28, 31]].tolist(), flush=True)
                hip.batch_free(h)
                hip.debug_set(verbose=0)
        bad += not (ok and okv)
    print("done: %d configs, %d differences, %d refused, %.1f s" % (n, bad, errs, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
