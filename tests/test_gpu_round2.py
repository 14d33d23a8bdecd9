"""Round-2 GPU tests: full-size C2 against the oracle region by region, one C4 shard at full size, the error paths of
the C ABI (pool overflows, unsupported flags, bad struct sizes), the Contig `trim` KATs on the device, 1-base reads,
buffers handed back between runs, batches driven from threads that never called ihp_init."""
import ctypes as C
import threading

import numpy as np
import pytest

import hand_vectors
from indelope_amd import Contig, synth
from indelope_amd import _abi as A
from indelope_amd import dist as idist
from indelope_amd.host import BatchResult, IhpError, RegionBatch

pytestmark = pytest.mark.gpu


def assert_same(got, exp):
    d = BatchResult.first_difference(got, exp)
    assert d is None, d
    np.testing.assert_allclose(got.events["gl"], exp.events["gl"], rtol=1e-12)
    np.testing.assert_allclose(got.events["qual"], exp.events["qual"], rtol=1e-12, atol=1e-12)


def region_slice(res, lo, hi):
    """The part of a BatchResult that belongs to regions [lo, hi), offsets rebased (for slice-against-oracle checks)."""
    import copy
    r = copy.copy(res)
    c0, c1 = int(res.contig_off[lo]), int(res.contig_off[hi])
    b0, b1 = int(res.ctg_seq_off[c0]), int(res.ctg_seq_off[c1])
    w0, w1 = int(res.cigar_off[c0]), int(res.cigar_off[c1])
    e0, e1 = int(res.event_off[c0]), int(res.event_off[c1])
    h0, h1 = int(res.hit_off[e0]), int(res.hit_off[e1])
    r.n_regions, r.n_contigs, r.n_events, r.n_hits = hi - lo, c1 - c0, e1 - e0, h1 - h0
    r.status, r.n_contigs_pre = res.status[lo:hi], res.n_contigs_pre[lo:hi]
    r.contig_off = res.contig_off[lo:hi + 1] - c0
    for f in ("ctg_start", "ctg_nreads", "aln_flags", "aln_ref_start", "aln_ref_len", "aln_ez"):
        setattr(r, f, getattr(res, f)[c0:c1])
    r.ctg_seq_off = res.ctg_seq_off[c0:c1 + 1] - b0
    r.ctg_seq, r.ctg_support = res.ctg_seq[b0:b1], res.ctg_support[b0:b1]
    r.cigar_off, r.cigar = res.cigar_off[c0:c1 + 1] - w0, res.cigar[w0:w1]
    r.event_off, r.events = res.event_off[c0:c1 + 1] - e0, res.events[e0:e1]
    r.hit_off = res.hit_off[e0:e1 + 1] - h0
    r.ref_hit, r.alt_hit = res.ref_hit[h0:h1], res.alt_hit[h0:h1]
    return r


# ------------------------------------------------------------------------------------------ full-size parity
def test_full_size_c2_every_region_against_the_oracle(hip, oracle):
    """BASELINE configs[1] at full size: all 10 000 regions bit-identical to the oracle (every thread of the host)."""
    b, _ = synth.config("C2")
    got = hip.run_regions(b)
    exp = oracle.run_regions_mt(b, None, 64)
    assert_same(got, exp)


def test_c3_and_c5_shapes_at_size(hip, oracle):
    """C3 (16-256 reads per region) and C5 (300 bp reads, two events, K = 31) on 2 000 / 1 000 regions."""
    b, _ = synth.config("C3", n_regions=2000)
    assert_same(hip.run_regions(b), oracle.run_regions_mt(b, None, 64))
    b, _ = synth.config("C5", n_regions=1000)
    p, po = hip.params(K=31), oracle.params(K=31)
    assert_same(hip.run_regions(b, p), oracle.run_regions_mt(b, po, 64))


def test_c4_shard_full_size(hip, oracle):
    """BASELINE configs[3]: one rank's share of the 5 M regions (625 000 regions, 40 M reads) resident on one GPU --
    determinism, conservation, oracle parity on slices, and the per-region records the multi-GPU gather moves."""
    world, rank = 8, 5
    total = synth.CONFIGS["C4"]["n_regions"]
    bounds = idist.shard_bounds(np.ones(total), world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    assert hi - lo == 625_000
    b, _ = synth.config("C4", n_regions=hi - lo, first_region=lo)
    b = b.with_trim_bounds()
    h = hip.batch_upload(b)
    try:
        hip.batch_run(h)
        hip.batch_sync(h)
        s1 = hip.batch_summary_host(h, b.n_regions).view(np.int32).reshape(-1, idist.SUMMARY_WORDS).copy()
        res = hip.batch_fetch(h)
        hip.batch_run(h)                                   # determinism: the second run leaves the same records
        hip.batch_sync(h)
        s2 = hip.batch_summary_host(h, b.n_regions).view(np.int32).reshape(-1, idist.SUMMARY_WORDS)
        assert np.array_equal(s1, s2)
    finally:
        hip.batch_free(h)
    assert (res.status == 0).all()
    # conservation: every read is inserted exactly once; trim can only drop contigs, never add reads
    nz = np.diff(res.contig_off) > 0
    per_region = np.add.reduceat(res.ctg_nreads, res.contig_off[:-1][nz])
    assert per_region.max() <= 64 and per_region.min() >= 1
    assert res.ctg_support.min() >= 1
    assert (np.diff(res.contig_off) <= 64).all() and (res.n_contigs_pre >= np.diff(res.contig_off)).all()
    ev = res.events[res.events["status"] == A.IHP_EV_TALLIED]
    assert len(ev) > 0.9 * b.n_regions
    assert (ev["ref_support"] + ev["alt_support"] - ev["both_found"] <= 64).all()
    # the records of the gather are the fetched results, region by region
    sv = s1.view(A.SUMMARY_DTYPE).reshape(-1)
    assert np.array_equal(sv["status"], res.status) and np.array_equal(sv["n_contigs_pre"], res.n_contigs_pre)
    assert np.array_equal(sv["n_contigs"], np.diff(res.contig_off))
    done = (res.aln_flags & A.IHP_ALN_DONE) != 0
    assert np.array_equal(sv["n_aligned"], np.add.reduceat(np.append(done, 0).astype(np.int64), res.contig_off[:-1])
                          * nz)                                 # regions without contigs have nothing aligned
    # oracle parity on slices across the shard (the shard's regions are independent of its neighbours)
    for s_lo in (0, 123_456, 311_111, 624_750):
        sub = b.slice(s_lo, s_lo + 250)
        exp = oracle.run_regions_mt(sub, None, 16)
        assert_same(region_slice(res, s_lo, s_lo + 250), exp)
        expsum = idist.summaries_from_result(exp).view(np.int32).reshape(-1, idist.SUMMARY_WORDS)
        assert np.array_equal(s1[s_lo:s_lo + 250], expsum)


# ------------------------------------------------------------------------------------------------ contig API
def test_contig_trim_on_the_device(hip, oracle):
    """trim(c, min_support) (contig.nim:49-68) through ihp_contig_trim itself: hand-checked cases + random vs oracle."""
    for fn in hand_vectors.TRIM_KATS:
        fn(hip)
    rng = np.random.default_rng(5)
    for _ in range(40):
        n = int(rng.integers(1, 300))
        seq = "".join(rng.choice(list("ACGT"), n))
        sup = rng.integers(0, 6, n).astype(np.uint32)
        ms = int(rng.integers(0, 6))
        a, b = Contig(seq, 100), Contig(seq, 100)
        a._sup[:n] = sup
        b._sup[:n] = sup
        a.nreads = b.nreads = 7
        hip.trim(a, ms)
        oracle.trim(b, ms)
        assert (a.sequence, a.support, a.start, a.nreads) == (b.sequence, b.support, b.start, b.nreads)


@pytest.mark.parametrize("vec", hand_vectors.REGION_VECTORS, ids=lambda v: v.__name__)
def test_hand_derived_vectors(hip, vec):
    """Hand-derived expectations (tests/hand_vectors.py: derivations in the comments) through the batched path."""
    vec(hip)


def test_one_base_reads(hip, oracle):
    """trim() empties a 1-base read whatever its quality (a == high == 0, indelope.nim:28-30): the quality path, the
    stager's trim bounds and the no-qualities path agree with the oracle."""
    b = hand_vectors.batch_with_one_base_reads()
    exp = oracle.run_regions(b)
    assert_same(hip.run_regions(b), exp)
    tb = b.with_trim_bounds()
    ln = np.diff(b.read_off)
    assert (ln == 1).sum() >= 3 and (tb.trim_hi[ln == 1] == 0).all()
    assert_same(hip.run_regions(tb), exp)
    nq = RegionBatch(b.region_read_off, b.read_off, b.bases, None, b.read_start, b.read_stop, b.mapq, b.read_skip,
                     b.ref_off, b.ref_bases, b.ref_origin)
    assert_same(hip.run_regions(nq), oracle.run_regions(nq))


# ------------------------------------------------------------------------------------------------ error paths
def test_bad_struct_size_and_arguments(hip):
    b, _ = synth.generate(4, n_reads=(8, 8), config_id=70)
    p = hip.params()
    p.struct_size = 8
    with pytest.raises(IhpError) as e:
        hip.run_regions(b, p)
    assert e.value.code == A.IHP_E_ARG
    p = hip.params(K=32)
    with pytest.raises(IhpError) as e:
        hip.run_regions(b, p)
    assert e.value.code == A.IHP_E_ARG
    bad = RegionBatch(b.region_read_off + 1, b.read_off, b.bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip,
                      b.ref_off, b.ref_bases, b.ref_origin)
    with pytest.raises(IhpError) as e:
        hip.run_regions(bad)
    assert e.value.code == A.IHP_E_ARG


def _ksw_call(fn, q, t, m, mat, gapo, gape, w, z, flag):
    ez = A.KswExtz()
    fn(None, len(q), A.ptr(q, A.u8p), len(t), A.ptr(t, A.u8p), m, A.ptr(mat, A.i8p), gapo, gape, w, z, flag, C.byref(ez))
    cig = [int(ez.cigar[i]) for i in range(ez.n_cigar)]
    out = (ez.max, ez.zdropped, ez.max_q, ez.max_t, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q, ez.score, ez.n_cigar)
    if ez.cigar:
        C.CDLL(None).free(ez.cigar)
    return out, cig


def test_ksw_flags_of_the_ffi_seam(hip, oracle):
    """KSW_EZ_SCORE_ONLY / APPROX_MAX / APPROX_DROP / GENERIC_SC (ksw2.h:8-12; the reference path never sets them) through
    the drop-in `ksw_extz2_sse` symbol: every ksw_extz_t field and the CIGAR against the reference's own C compiled by
    oracle/Makefile (oracle/_ref).  The per-region path, which needs the CIGAR and the exact maximum, refuses them."""
    import test_oracle_ksw2 as tk
    for flag in (A.KSW_EZ_SCORE_ONLY, A.KSW_EZ_APPROX_MAX, A.KSW_EZ_GENERIC_SC):
        with pytest.raises(IhpError) as e:
            hip.run_regions(synth.generate(2, n_reads=(8, 8), config_id=71)[0], hip.params(ksw_flag=flag))
        assert e.value.code == A.IHP_E_UNSUPPORTED
    ref = oracle.ref_lib()
    if ref is None:
        pytest.skip("oracle/_ref (the compiled reference ksw2) is not built")
    S, R, G, AM, AD, X, RV = (A.KSW_EZ_SCORE_ONLY, A.KSW_EZ_RIGHT, A.KSW_EZ_GENERIC_SC, A.KSW_EZ_APPROX_MAX, A.KSW_EZ_APPROX_DROP,
                              A.KSW_EZ_EXTZ_ONLY, A.KSW_EZ_REV_CIGAR)
    rng = np.random.default_rng(77)
    gen = rng.integers(-6, 3, (5, 5)).astype(np.int8)          # a generic matrix: positive diagonal, no wildcard row
    gen[np.arange(5), np.arange(5)] = [2, 3, 1, 2, 1]
    gen = np.ascontiguousarray(gen.reshape(-1))
    plain = hip.matrix()
    sets = [(S, plain, 4, 1, 50, 400), (S | R, plain, 4, 2, -1, -1), (AM, plain, 4, 1, 50, 400), (AM | AD, plain, 4, 1, 50, 100),
            (AM | AD | X, plain, 4, 1, 30, 60), (AM | R | RV, plain, 5, 1, -1, -1), (G, gen, 4, 1, 50, 400), (G | R, gen, 6, 2, -1, 200),
            (G | AM | AD, gen, 4, 1, 40, 80), (G | S, gen, 4, 1, 50, 400), (0x100, plain, 4, 1, 50, 400)]
    n = 0
    for si, (flag, mat, go, ge, w, z) in enumerate(sets):
        for q, t in tk.cases(900 + si, 12):
            qe, te = hip.encode(q), hip.encode(t)
            got = _ksw_call(hip.cdll.ksw_extz2_sse, qe, te, 5, mat, go, ge, w, z, flag)
            assert hip.b.ksw_last_status() == 0
            exp = _ksw_call(ref.ksw_extz2_sse, qe, te, 5, mat, go, ge, w, z, flag)
            assert got == exp, (si, hex(flag), len(q), len(t), got, exp)
            n += 1
        if flag & (S | G | AM):
            assert hip.b.debug_last_ksw_mode() == 2
    assert n >= 120


@pytest.mark.parametrize("which", ["cigar", "events", "hits", "ksw"])
def test_pool_overflow_is_reported(hip, which):
    """CIGAR pool, event pool, hit pool and ksw2 traceback scratch too small (test hook ihp_debug_limits): the run
    reports IHP_E_CAPACITY at ihp_batch_sync and again at fetch, with the pool named in the message; nothing crashes and
    the next batch (limits reset) is fine."""
    if which == "cigar":        # CIGARs longer than the 32-word slot need the bump pool: many short events
        b = hand_vectors.batch_with_long_cigars()
        lim = dict(cigar_words=8)
    elif which == "events":
        b, _ = synth.generate(64, n_reads=(32, 48), err_rate=0.0, config_id=72)
        lim = dict(events=4)
    elif which == "hits":       # more tallied events in a region than its fixed hit slots
        b = hand_vectors.batch_with_many_events()
        lim = dict(hits=1)
    else:
        b, _ = synth.generate(32, n_reads=(32, 48), err_rate=0.0, config_id=73)
        lim = dict(ksw_bytes=4096)
    hip.debug_limits(**lim)
    try:
        h = hip.batch_upload(b)
    finally:
        hip.debug_limits()
    try:
        hip.batch_run(h)
        with pytest.raises(IhpError) as e:
            hip.batch_sync(h)
        assert e.value.code == A.IHP_E_CAPACITY and "overflow" in str(e.value)
        with pytest.raises(IhpError) as e:
            hip.batch_fetch(h)
        assert e.value.code == A.IHP_E_CAPACITY
    finally:
        hip.batch_free(h)
    ok = hip.run_regions(b)                          # same input, library sizing: fine
    assert (ok.status == 0).all()


# ---------------------------------------------------------------------------------------------- buffers / threads
def test_release_outputs_between_runs(hip, oracle):
    """ihp_batch_release_outputs: the inputs and the summary records stay, the next run takes buffers again and gives
    the same results; fetch without that run is refused."""
    b, _ = synth.generate(300, n_reads=(16, 64), err_rate=1e-3, config_id=74, dup_frac=0.1)
    exp = oracle.run_regions(b)
    h = hip.batch_upload(b)
    try:
        hip.batch_run(h)
        hip.batch_sync(h)
        hip.batch_release_outputs(h)
        with pytest.raises(IhpError):
            hip.batch_fetch(h)
        for _ in range(2):
            hip.batch_run(h)
            hip.batch_sync(h)
            assert_same(hip.batch_fetch(h), exp)
            hip.batch_release_outputs(h)
    finally:
        hip.batch_free(h)


def test_runs_need_no_memset(hip, oracle):
    """A batch is cleared once at upload; every later run relies on the previous run's last kernel having reset the
    counters, work queues and per-region hit counts: five runs in a row, identical results."""
    b, _ = synth.generate(500, n_reads=(16, 64), err_rate=2e-3, config_id=75, dup_frac=0.2)
    exp = oracle.run_regions(b)
    h = hip.batch_upload(b)
    try:
        for _ in range(5):
            hip.batch_run(h)
            hip.batch_sync(h)
        assert_same(hip.batch_fetch(h), exp)
    finally:
        hip.batch_free(h)


def test_threads_that_never_called_init(hip, oracle):
    """HIP's current device is per thread: batch calls from a fresh host thread bind it themselves."""
    b, _ = synth.generate(100, n_reads=(8, 64), err_rate=1e-3, config_id=76)
    exp = oracle.run_regions(b)
    out = []

    def worker():
        try:
            out.append(hip.run_regions(b))
        except Exception as e:          # pragma: no cover
            out.append(e)
    th = [threading.Thread(target=worker) for _ in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in out:
        assert not isinstance(r, Exception), r
        assert_same(r, exp)


def test_c_harness_drives_the_device():
    """The C harness (gcc, include/indelope_hip.h, no Python in between): the reference's ksw2 KAT through the drop-in
    symbol incl. cigar-buffer reuse, a hand-derived region through ihp_run_regions, the error convention."""
    import json
    import subprocess
    from test_abi_exports import _build_harness
    pr = subprocess.run([_build_harness(), "--gpu"], capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    # (the JSON object is the last line: librccl prints a banner on stdout when the harness forms its group of one)
    assert json.loads(pr.stdout.strip().splitlines()[-1])["gpu_checks"] is True


def _pair(rng, qlen, tlen, sub=0.01, indel=0.01, shift=None):
    """A query of ~qlen bases derived from a window of a random target of tlen bases."""
    t = rng.integers(0, 4, tlen)
    lo = 0 if shift is None else shift
    src = t[lo:lo + qlen].copy()
    q = []
    for b in src:
        u = rng.random()
        if u < indel / 2:
            continue
        if u < indel:
            q += [b, int(rng.integers(0, 4))]
            continue
        q.append(int((b + rng.integers(1, 4)) % 4) if rng.random() < sub else int(b))
    q = np.array(q if q else [0], np.uint8)
    L = "ACGT"
    return "".join(L[i] for i in q), "".join(L[i] for i in t)


def test_ksw_sweep_phases(hip, oracle):
    """The production sweep is laid out by phase (early / steady periods / tail runs cut by the query or the target);
    shapes that put the ends of the phases everywhere -- query longer and shorter than the target, jobs too short for a
    steady phase, z-drops in every phase, every band width the steady loop takes -- against the oracle, field by field."""
    rng = np.random.default_rng(20261002)
    R = A.KSW_EZ_RIGHT
    n = 0
    for w, z, flag, go, ge in ((50, 400, 0, 4, 1), (49, 40, 0, 4, 1), (62, 15, R, 4, 2), (55, 100, 0, 6, 1), (50, -1, 0, 4, 1),
                               (57, 25, R, 4, 1), (50, 8, 0, 4, 1), (61, 60, 0, 5, 2)):
        qs, ts = [], []
        for _ in range(150):
            shape = rng.integers(0, 6)
            if shape == 0:
                ql, tl = int(rng.integers(200, 420)), int(rng.integers(200, 420))             # either may be the longer one
            elif shape == 1:
                tl = int(rng.integers(120, 200)); ql = int(rng.integers(300, 500))          # cut by the target first
            elif shape == 2:
                ql = int(rng.integers(w + 20, w + 45)); tl = int(rng.integers(w + 20, 300))  # around the "room for the early phase" limit
            elif shape == 3:
                ql, tl = int(rng.integers(1, 90)), int(rng.integers(1, 90))                  # no steady phase at all
            elif shape == 4:
                ql = int(rng.integers(250, 400)); tl = ql + int(rng.integers(-3, 60))        # ends close together
            else:
                ql, tl = int(rng.integers(500, 700)), int(rng.integers(500, 700))            # many periods
            sub = float(rng.choice([0.0, 0.01, 0.05, 0.3]))                                   # 0.3: scores fall, z-drops
            q, t = _pair(rng, min(ql, tl) if rng.random() < 0.5 else ql, tl, sub=sub, indel=float(rng.choice([0.0, 0.01, 0.04])),
                         shift=int(rng.integers(0, max(1, tl // 3))) if rng.random() < 0.4 else None)
            if len(q) < ql:                                                                   # pad the query with unrelated bases
                q += "".join("ACGT"[i] for i in rng.integers(0, 4, ql - len(q)))
            qs.append(q); ts.append(t)
        kw = dict(match=1, mismatch=-2, gap_open=go, gap_ext=ge, bw=w, z=z, flag=flag)
        ez, cg = hip.align_batch(qs, ts, **kw)
        assert hip.b.debug_last_ksw_mode() == (4 if flag & R else 3)
        ez2, cg2 = oracle.align_batch(qs, ts, **kw)
        for i in range(len(qs)):
            assert ez[i].tolist() == ez2[i].tolist(), (w, z, flag, i, len(qs[i]), len(ts[i]), ez[i], ez2[i])
            assert cg[i].tolist() == cg2[i].tolist(), (w, z, flag, i, len(qs[i]), len(ts[i]))
        n += len(qs)
    assert n == 1200

