"""Round-4 GPU tests: two alignments per wavefront (ksw_pair.h) against the compiled reference and against the single sweep,
who-is-paired-with-whom independence, the per-shape launch plans on a stream of differently shaped batches, runs confirmed by
every call that lets go of them, the slab layout check, and a bounded pass of the stress tools with fresh seeds."""
import copy
import datetime
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from indelope_amd import synth
from indelope_amd import _abi as A
from indelope_amd.host import BatchResult, IhpError

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def assert_same(got, exp):
    d = BatchResult.first_difference(got, exp)
    assert d is None, d
    np.testing.assert_allclose(got.events["gl"], exp.events["gl"], rtol=1e-12)


def pair_batch(rng, w, n_len=8, per_len=(2, 10), wild=True):
    """Jobs k_ksw_plan can pair: a handful of contig lengths, several jobs of each, windows mostly longer than qlen + w."""
    qs, ts = [], []
    for _ in range(n_len):
        ql = int(rng.integers(w + 32, 600))
        for _ in range(int(rng.integers(*per_len))):
            u = rng.random()
            tl = ql + w + 1 + int(rng.integers(0, 150)) if u < 0.85 else ql + int(rng.integers(0, w + 1))
            t = rng.integers(0, 4, tl)
            sub = float(rng.choice([0, 0.01, 0.05, 0.3]))
            lo = int(rng.integers(0, max(1, tl - ql))) if rng.random() < 0.3 else 0
            src = t[lo:lo + ql]
            q = np.where(rng.random(len(src)) < sub, (src + rng.integers(1, 4, len(src))) % 4, src)
            if rng.random() < 0.5 and ql > 120:                           # an indel in the middle: the event indelope is after
                a = int(rng.integers(40, ql - 60)); L = int(rng.integers(5, 40))
                q = np.concatenate([q[:a], q[a + L:], rng.integers(0, 4, L)]) if rng.random() < 0.5 else np.concatenate([q[:a], rng.integers(0, 4, L), q[a:]])[:ql]
            q = q[:ql] if len(q) >= ql else np.concatenate([q, rng.integers(0, 4, ql - len(q))])
            if wild and rng.random() < 0.05:
                t = t.copy(); t[int(rng.integers(0, tl))] = 4             # a wildcard in the window: the pair sweep takes it
            if wild and rng.random() < 0.03:
                q = q.copy(); q[int(rng.integers(0, ql))] = 4             # one in the contig: that job takes the single sweep
            qs.append(q.astype(np.uint8)); ts.append(t.astype(np.uint8))
    order = rng.permutation(len(qs))
    return [qs[i] for i in order], [ts[i] for i in order]


@pytest.mark.parametrize("w,z,flag,score", [(50, 400, 0, (1, -2, 4, 1)), (49, -1, 0, (1, -2, 4, 1)), (62, 20, A.KSW_EZ_EXTZ_ONLY, (2, -3, 5, 2)),
                                            (55, 5, A.KSW_EZ_REV_CIGAR, (1, -4, 6, 1)), (50, 60, 0, (3, -1, 2, 3)), (57, 1000, 0, (1, -2, 4, 1))])
def test_pair_sweep_matches_the_reference_and_the_single_sweep(hip, oracle, w, z, flag, score):
    """ksw_pair.h: jobs of equal contig length share a wavefront (16-bit halves of every DP register); every field of every
    record and every CIGAR equals the compiled reference's (src/ksw2/csrc/ksw2_extz2_sse.c), and the bytes do not change when
    the pairing is switched off."""
    rng = np.random.default_rng(w * 1000 + (z & 0xff))
    ma, mi, go, ge = score
    kw = dict(match=ma, mismatch=mi, gap_open=go, gap_ext=ge, bw=w, z=z, flag=flag, encoded=True)
    for _ in range(4):
        qs, ts = pair_batch(rng, w)
        ez, cg = hip.align_batch(qs, ts, **kw)
        assert hip.b.debug_last_ksw_pairs() > 0
        ez2, cg2 = oracle.align_batch(qs, ts, **kw)
        hip.debug_set(ksw_pair=0)
        try:
            ez3, cg3 = hip.align_batch(qs, ts, **kw)
            assert hip.b.debug_last_ksw_pairs() == 0
        finally:
            hip.debug_set()
        for i in range(len(qs)):
            assert ez[i].tolist() == ez2[i].tolist() and cg[i].tolist() == cg2[i].tolist(), (i, len(qs[i]), len(ts[i]), ez[i], ez2[i])
            assert ez3[i].tolist() == ez2[i].tolist() and cg3[i].tolist() == cg2[i].tolist(), (i, "single")


def test_results_do_not_depend_on_who_is_paired_with_whom(hip, oracle):
    """k_ksw_plan hands out ranks with atomics: the partner of a job differs from run to run and with the order of the jobs.
    The same jobs in three orders, and each job next to copies of itself, give the same records."""
    rng = np.random.default_rng(44)
    qs, ts = pair_batch(rng, 50, n_len=5, per_len=(3, 12), wild=False)
    kw = dict(match=1, mismatch=-2, gap_open=4, gap_ext=1, bw=50, z=400, flag=0, encoded=True)
    ez0, cg0 = oracle.align_batch(qs, ts, **kw)
    for seed in (1, 2, 3):
        order = np.random.default_rng(seed).permutation(len(qs))
        ez, cg = hip.align_batch([qs[i] for i in order], [ts[i] for i in order], **kw)
        for k, i in enumerate(order):
            assert ez[k].tolist() == ez0[i].tolist() and cg[k].tolist() == cg0[i].tolist()
    ez, cg = hip.align_batch([q for q in qs for _ in range(2)], [t for t in ts for _ in range(2)], **kw)
    for i in range(len(qs)):
        for c in range(2):
            assert ez[2 * i + c].tolist() == ez0[i].tolist() and cg[2 * i + c].tolist() == cg0[i].tolist()


def test_the_pair_launch_inside_the_region_path(hip, oracle):
    """ihp_batch_run: k_ksw_plan_count / _place, the single sweep for the jobs without a partner on the second stream, k_ksw_pair
    for the rest: C2- and C5-shaped batches against the oracle with the pairing on and off, at a size where most jobs pair."""
    for name, n, K in (("C2", 1500, 27), ("C5", 400, 31)):
        b, _ = synth.config(name, n_regions=n)
        exp = oracle.run_regions_mt(b, oracle.params(K=K), 16)
        for pair in (1, 0):
            hip.debug_set(ksw_pair=pair)
            try:
                assert_same(hip.run_regions(b, hip.params(K=K)), exp)
            finally:
                hip.debug_set()


def test_a_stream_of_differently_shaped_batches_keeps_its_plans_apart(hip, oracle):
    """The launch plan of a run (tiers, retry launches, the roomy ksw2 launch) comes from the last batch OF THE SAME SHAPE: shape A,
    then B (which needs the retry route), then A again -- every result equals the oracle's, A is never repeated because of B, and
    B is repeated at most once (the first time its shape is seen behind a clean batch of that shape it is not: nothing is
    speculated without a plan)."""
    a, _ = synth.config("C2", n_regions=500)
    exp_a = oracle.run_regions_mt(a, oracle.params(K=27), 16)
    bb, _ = synth.generate(300, n_reads=(8, 96), err_rate=2e-3, config_id=77, dup_frac=0.2)
    bases = bb.bases.copy()
    idx = np.random.default_rng(3).integers(0, len(bases), 40)
    bases[idx[:20]] = ord("N")
    bases[idx[20:]] |= 0x20
    bb.bases = bases
    exp_b = oracle.run_regions_mt(bb, oracle.params(K=27), 16)
    c5, _ = synth.config("C5", n_regions=200)
    exp_c = oracle.run_regions_mt(c5, oracle.params(K=31), 16)
    hip.debug_set()                                             # forgets every plan
    reruns = {"A": 0, "B": 0, "C": 0}
    runs = {"A": 0, "B": 0, "C": 0}
    for which in "ABACABCAABBA":
        b, exp, K = {"A": (a, exp_a, 27), "B": (bb, exp_b, 27), "C": (c5, exp_c, 31)}[which]
        h = hip.batch_upload(b, hip.params(K=K))
        try:
            hip.batch_run(h)
            hip.batch_sync(h)
            prof = hip.batch_profile(h)
            reruns[which] += int(prof[31]); runs[which] += 1
            assert_same(hip.batch_fetch(h), exp)
        finally:
            hip.batch_free(h)
    assert reruns["A"] == 0 and reruns["C"] == 0 and reruns["B"] == 0, (reruns, runs)


def test_release_outputs_and_summary_dev_confirm_the_run(hip, oracle):
    """a run that left launches out is only final once somebody has checked its counters.  ihp_batch_release_outputs
    and ihp_batch_summary_dev now do (they used to wait for the stream only): the records that stay behind after a release are
    those of the repeated run."""
    from indelope_amd.dist import summaries_from_result
    clean, _ = synth.generate(64, n_reads=(30, 40), err_rate=0.0, config_id=9)
    dirty = copy.copy(clean)
    bases = clean.bases.copy()
    idx = np.random.default_rng(5).integers(0, len(bases), 12)
    bases[idx[:6]] = ord("N")
    bases[idx[6:]] |= 0x20
    dirty.bases = bases
    exp = oracle.run_regions(dirty)
    want = summaries_from_result(exp)
    for how in ("release", "summary_dev"):
        hip.debug_set()
        for _ in range(3):                                      # (CLEAN_MIN batches in a row: only then is the retry route left out)
            h = hip.batch_upload(clean)
            hip.batch_run(h); hip.batch_sync(h)
            hip.batch_free(h)
        h = hip.batch_upload(dirty)
        try:
            hip.batch_run(h)                                    # speculates on the clean batches' plan
            if how == "release":
                hip.batch_release_outputs(h)
            ptr, n = hip.batch_summary_dev(h)
            rec = hip.copy_to_host(ptr, n * A.SUMMARY_DTYPE.itemsize).view(A.SUMMARY_DTYPE)
            assert np.array_equal(rec, want), how
        finally:
            hip.batch_free(h)


def test_a_slab_layout_that_is_not_the_librarys_is_refused(hip, oracle):
    """ihp_batch_upload_slab follows the offsets of the caller's ihp_slab_layout: every one of them is compared with
    ihp_slab_layout_for first, unknown flag bits are refused."""
    import ctypes as C
    b, _ = synth.config("C2", n_regions=50)
    slab = hip.make_slab(b.with_trim_bounds())
    try:
        h = hip.batch_upload_slab(slab)
        hip.batch_free(h)
        for field in ("read_off", "ref_bases", "bases4", "bytes"):
            bad = copy.copy(slab)
            lay = A.SlabLayout()
            C.memmove(C.byref(lay), C.byref(slab.layout), C.sizeof(lay))
            setattr(lay, field, getattr(lay, field) + 64)
            bad.layout = lay
            with pytest.raises(IhpError) as e:
                hip.batch_upload_slab(bad)
            assert e.value.code == A.IHP_E_ARG, field
        bad = copy.copy(slab)
        bad.flags = slab.flags | 0x40
        with pytest.raises(IhpError) as e:
            hip.batch_upload_slab(bad)
        assert e.value.code == A.IHP_E_ARG
    finally:
        slab.free()


@pytest.mark.parametrize("cfg", [
    dict(n_regions=60, n_reads=(24, 64), err_rate=1e-3, config_id=141, dup_frac=0.6),
    dict(n_regions=40, read_len=100, n_reads=(20, 60), err_rate=2e-3, config_id=142, K=21, dup_frac=0.8),
    dict(n_regions=40, read_len=192, n_reads=(16, 48), err_rate=5e-3, config_id=143, K=31, dup_frac=0.7, window_len=590),   # the longest read of the three-slot build
    dict(n_regions=40, read_len=193, n_reads=(16, 48), err_rate=1e-3, config_id=144, K=31, dup_frac=0.7, window_len=590),   # one more: five slots
    dict(n_regions=30, read_len=320, n_reads=(16, 40), err_rate=2e-3, config_id=146, K=31, dup_frac=0.7, window_len=980),   # the longest read the sweep takes
    dict(n_regions=30, read_len=321, n_reads=(16, 40), err_rate=1e-3, config_id=147, K=31, dup_frac=0.7, window_len=980),   # one more: ksw_wide.h
    dict(n_regions=50, n_reads=(8, 40), err_rate=2e-2, config_id=145, dup_frac=0.9),
])
@pytest.mark.parametrize("scoring", [dict(), dict(fb_match=2, fb_mismatch=-4, fb_gap_open=7, fb_gap_ext=2), dict(fb_flag=0xc0)])
def test_fallback_two_target_sweep_matches_oracle_and_single_sweeps(hip, oracle, cfg, scoring):
    """the two alignments of a fallback item (read vs window, read vs contig; indelope.nim:336-344) share one
    sweep (ksw_duo.h).  Same votes as the oracle and as the one-at-a-time sweeps, whatever the read length and the scoring."""
    b, _ = synth.generate(**cfg)
    K = cfg.get("K", 27)
    want = oracle.run_regions(b, oracle.params(K=K, **scoring))
    try:
        for duo in (1, 0):
            hip.debug_set(fb_duo=duo)
            assert_same(hip.run_regions(b, hip.params(K=K, **scoring)), want)
    finally:
        hip.debug_set()
    if cfg["config_id"] in (141, 143):
        assert np.count_nonzero(want.events["fallback_needed"]) > 0           # the case exists in what was compared


def _reorder_reads(b, mode, seed):
    """The same regions with the reads of every region in another order (ragged arrays rebuilt read by read)."""
    rng = np.random.default_rng(seed)
    order = []
    for r in range(b.n_regions):
        i0, i1 = int(b.region_read_off[r]), int(b.region_read_off[r + 1])
        idx = np.arange(i0, i1)
        order.append(idx[::-1] if mode == "reverse" else rng.permutation(idx))
    order = np.concatenate(order) if order else np.zeros(0, np.int64)
    lens = np.diff(b.read_off)[order]
    read_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pick = np.concatenate([np.arange(b.read_off[i], b.read_off[i + 1]) for i in order]) if len(order) else np.zeros(0, np.int64)
    c = copy.copy(b)
    c.read_off = read_off
    c.bases = b.bases[pick]
    c.quals = b.quals[pick] if b.quals is not None and len(b.quals) == len(b.bases) else b.quals
    c.read_start, c.read_stop = b.read_start[order], b.read_stop[order]
    c.mapq, c.read_skip = b.mapq[order], b.read_skip[order]
    if b.trim_lo is not None:
        c.trim_lo, c.trim_hi = b.trim_lo[order], b.trim_hi[order]
    return c


@pytest.mark.parametrize("mode", ["reverse", "random"])
@pytest.mark.parametrize("cfg", [
    dict(n_regions=150, n_reads=(24, 64), err_rate=1e-3, config_id=151),
    dict(n_regions=80, n_reads=(16, 200), err_rate=2e-3, config_id=152),
    dict(n_regions=40, read_len=300, n_reads=(32, 64), err_rate=1e-3, n_events=2, window_len=1400, event_pos=500, config_id=153, K=31),
])
def test_reads_in_any_order_extend_contigs_on_either_side(hip, oracle, cfg, mode):
    """The read phase's query offsets (contig.nim:114-135: the read reaches past a contig's START) are only looked at where they
    can still beat what the target offsets found (round 4).  A position-sorted batch hardly uses them: the same regions with
    their reads reversed -- every read then extends its contig to the left -- and in random order use both forms of that walk
    (a few offsets against all contig heads; every contig against all offsets)."""
    b, _ = synth.generate(**cfg)
    b = _reorder_reads(b, mode, cfg["config_id"])
    K = cfg.get("K", 27)
    got = hip.run_regions(b, hip.params(K=K))
    assert_same(got, oracle.run_regions(b, oracle.params(K=K)))
    assert (got.status == 0).all()


def test_batches_with_hardly_any_bases(hip, oracle):
    """The pipelined 2-bit packing loads without asking whether a read has bases: a batch whose reads are all trimmed away to
    nothing / a few bases long takes the plain kernel or clamps, and gives what the oracle gives (no contigs)."""
    b, _ = synth.generate(6, n_reads=(3, 5), err_rate=0.0, config_id=161)
    b = b.with_trim_bounds()
    for keep in (0, 3):
        c = copy.copy(b)
        nr = c.n_reads
        c.read_off = (np.arange(nr + 1) * keep).astype(np.int64)
        c.bases = np.frombuffer(b"ACG" * nr, np.uint8).copy()[:keep * nr] if keep else np.zeros(0, np.uint8)
        c.quals = np.full(len(c.bases), 40, np.uint8)
        c.read_stop = c.read_start + keep
        c.trim_lo = np.zeros(nr, np.int32); c.trim_hi = np.full(nr, keep, np.int32)
        got = hip.run_regions(c, hip.params(K=27))
        assert_same(got, oracle.run_regions(c, oracle.params(K=27)))


def test_stress_tools_with_todays_seeds():
    """The randomised harnesses under tools/ found every device bug of round 3 and none of the fixed-seed tests did: a bounded
    pass of each (fresh seed from the date, a few seconds apiece) runs where the driver can see it.  The seed is printed on
    failure for replay."""
    day = datetime.date.today()
    seed = day.year * 10000 + day.month * 100 + day.day
    env = dict(os.environ, PYTHONPATH=ROOT)
    jobs = [("ksw_pair_stress.py", [str(seed), "12"]), ("ksw_stress.py", [str(seed), "4"]), ("stress_parity.py", ["16", str(seed), "params"]),
            ("thread_stress.py", ["3", "6", str(seed)]), ("sweep_stress.py", ["3", str(seed)]), ("fb_stress.py", [str(seed), "8"])]
    t0 = time.time()
    for tool, args in jobs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
        tail = (r.stdout + r.stderr)[-1500:]
        assert r.returncode == 0, "tools/%s %s (seed %d) failed:\n%s" % (tool, " ".join(args), seed, tail)
        assert "DIFF" not in r.stdout, "tools/%s seed %d:\n%s" % (tool, seed, tail)
    assert time.time() - t0 < 600
