"""Per-region path of the CPU oracle: golden fixtures, determinism, planted-event recovery."""
import numpy as np
import pytest

import golden_util
from indelope_amd import synth
from indelope_amd import _abi as A
from indelope_amd.host import BatchResult


@pytest.mark.parametrize("name", ["c1", "small", "long", "dup"])
def test_golden_regions(oracle, name):
    golden_util.check_regions(oracle, name)


def test_golden_ksw2(oracle):
    assert golden_util.check_ksw2(oracle) >= 200


def test_c1_recovers_planted_event(oracle):
    """BASELINE config C1 end to end on the CPU path: the planted indel comes back as a HET call."""
    b, truth = synth.config("C1")
    res = oracle.run_regions(b)
    ev = res.events[res.events["status"] == A.IHP_EV_TALLIED]
    assert len(ev) == 1
    assert ev[0]["type"] == truth[0, 0] and ev[0]["len"] == truth[0, 1]
    assert ev[0]["tstart"] == b.ref_origin[0] + 250
    assert ev[0]["gt"] == A.IHP_GT_HET
    assert ev[0]["ref_support"] + ev[0]["alt_support"] == 48 and ev[0]["both_found"] == 0


def test_error_free_regions_recover_truth(oracle):
    b, truth = synth.generate(40, n_reads=(32, 64), err_rate=0.0, config_id=7)
    res = oracle.run_regions(b)
    hits = 0
    for r in range(b.n_regions):
        evs = []
        for c in range(res.contig_off[r], res.contig_off[r + 1]):
            evs += [e for e in res.events[res.event_off[c]:res.event_off[c + 1]] if e["status"] == 0]
        hits += any(e["type"] == truth[r, 0] and e["len"] == truth[r, 1] for e in evs)
    assert hits >= 36          # a few regions lose the event to coverage / band limits


def test_alignment_fallback(oracle):
    """indelope.nim:312-372: tandem duplications put the alt k-mer into the reference haplotype too, so reference
    reads carry both k-mers (both_found > 0) and the per-read alignments decide.  fallback=0 keeps the k-mer counts."""
    b, truth = synth.generate(30, n_reads=(24, 64), err_rate=1e-3, config_id=41, dup_frac=0.6)
    res = oracle.run_regions(b)
    off = oracle.run_regions(b, oracle.params(fallback=0))
    ev, ev0 = res.events, off.events
    fb = ev["aligned"] == 1
    assert fb.sum() >= 10 and (ev0["aligned"] == 0).all()
    assert np.array_equal(ev["fallback_needed"], ev0["fallback_needed"]) and np.array_equal(fb, ev["fallback_needed"] == 1)
    # the k-mer tally is kept beside the votes; without the fallback it is the result
    for k in ("ref_support", "alt_support", "both_found"):
        assert np.array_equal(ev["kmer_" + k], ev0[k])
        assert np.array_equal(ev[k][~fb], ev0[k][~fb])
    assert (ev["kmer_both_found"][fb] > 0).all() and (ev["both_found"][fb] == 0).all()
    # a read votes at most once, and the votes separate the haplotypes where the k-mers could not
    nreads = np.diff(b.region_read_off).max()
    assert ((ev["ref_support"] + ev["alt_support"])[fb] <= nreads).all()
    assert (ev["alt_support"][fb] < ev["kmer_alt_support"][fb]).all()
    assert (ev["gt"][fb] == A.IHP_GT_HET).sum() >= 0.7 * fb.sum()
    # mapq < 10 reads do not vote (:325); quality-trimmed starts shift the window (:328)
    b.mapq = b.mapq.copy()
    b.mapq[::3] = 9
    b.quals = b.quals.copy()
    for i in range(0, b.n_reads, 5):
        b.quals[b.read_off[i]:b.read_off[i] + 7] = 2
    low = oracle.run_regions(b)
    fb2 = low.events["aligned"] == 1
    assert fb2.sum() >= 5
    assert ((low.events["ref_support"] + low.events["alt_support"])[fb2] <= nreads - nreads // 3).all()


def test_first_hit_positions_are_consistent_with_the_counts(oracle):
    """hit_off/ref_hit/alt_hit: per tallied event one entry per read of the region; the counts of indelope.nim:301-311
    are the numbers of non-negative entries, and the window at a reported index is the k-mer or its reverse complement."""
    b, _ = synth.generate(40, n_reads=(12, 64), err_rate=1e-3, config_id=46, dup_frac=0.3)
    b.mapq = b.mapq.copy()
    b.mapq[::7] = 5
    res = oracle.run_regions(b)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    assert res.n_hits == res.hit_off[-1] and len(res.ref_hit) == res.n_hits == len(res.alt_hit)
    seen = 0
    for c in range(res.n_contigs):
        r = int(np.searchsorted(res.contig_off, c, side="right") - 1)
        r0, r1 = b.region_read_off[r], b.region_read_off[r + 1]
        for e in range(res.event_off[c], res.event_off[c + 1]):
            ev = res.events[e]
            h0, h1 = res.hit_off[e], res.hit_off[e + 1]
            if ev["status"] != A.IHP_EV_TALLIED:
                assert h0 == h1
                continue
            assert h1 - h0 == r1 - r0
            rh, ah = res.ref_hit[h0:h1], res.alt_hit[h0:h1]
            assert (rh >= 0).sum() == ev["kmer_ref_support"] and (ah >= 0).sum() == ev["kmer_alt_support"]
            assert ((rh >= 0) & (ah >= 0)).sum() == ev["kmer_both_found"]
            assert (rh[b.mapq[r0:r1] < 10] == -1).all() and (ah[b.mapq[r0:r1] < 10] == -1).all()
            for k, pos in ((ev["ref_kmer"], rh), (ev["alt_kmer"], ah)):
                for i in np.flatnonzero(pos >= 0)[:6]:
                    lo = b.read_off[r0 + i] + pos[i]
                    w = b.bases[lo:lo + len(k)].tobytes()
                    assert w == k or w == k.translate(comp)[::-1]
                    # first hit: no earlier window of the read matches
                    rd = b.bases[b.read_off[r0 + i]:b.read_off[r0 + i + 1]].tobytes()
                    first = min(x for x in (rd.find(k), rd.find(k.translate(comp)[::-1])) if x >= 0)
                    assert first == pos[i]
            seen += 1
    assert seen >= 20


def test_trim_bounds_equal_qualities(oracle):
    """RegionBatch.with_trim_bounds (the stager's trim(), indelope.nim:23-38) against orc_read_trim on every read."""
    b, _ = synth.generate(40, n_reads=(8, 40), err_rate=1e-3, config_id=48)
    rng = np.random.default_rng(1)
    q = rng.choice(np.array([2, 14, 15, 30], np.uint8), len(b.quals), p=[0.3, 0.1, 0.1, 0.5])
    for i in range(0, b.n_reads, 9):
        q[b.read_off[i]:b.read_off[i + 1]] = 2
    for i in range(4, b.n_reads, 9):
        q[b.read_off[i]:b.read_off[i + 1] - 1] = 2
    b.quals = q
    tb = b.with_trim_bounds()
    for i in range(b.n_reads):
        a, lo, hi = oracle.read_trim(q[b.read_off[i]:b.read_off[i + 1]])
        if lo == hi:
            assert tb.trim_lo[i] == tb.trim_hi[i] == a, i
        else:
            assert (tb.trim_lo[i], tb.trim_hi[i]) == (lo, hi) and a == lo, i
    assert BatchResult.first_difference(oracle.run_regions(b), oracle.run_regions(tb)) is None


def test_threads_do_not_change_results(oracle):
    b, _ = synth.generate(64, n_reads=(16, 64), err_rate=1e-3, config_id=8)
    a = oracle.run_regions(b)
    c = oracle.run_regions_mt(b, nthreads=4)
    assert BatchResult.first_difference(a, c) is None


def test_reference_ksw_plug_gives_identical_regions(oracle):
    if not oracle.use_reference_ksw(True):
        pytest.skip("oracle/_ref not built")
    try:
        b, _ = synth.generate(64, n_reads=(16, 64), err_rate=1e-3, config_id=9)
        a = oracle.run_regions(b)
    finally:
        oracle.use_reference_ksw(False)
    c = oracle.run_regions(b)
    assert BatchResult.first_difference(a, c) is None


def test_edge_regions(oracle):
    """Empty region, all reads filtered, reads emptied by the quality trim (indelope.nim:28-30 quirk)."""
    b, _ = synth.generate(3, n_reads=(12, 12), config_id=10)
    b.mapq = b.mapq.copy()
    b.quals = b.quals.copy()
    r0, r1 = b.region_read_off[1], b.region_read_off[2]
    b.mapq[r0:r1] = 3                                   # region 1: nothing assembles, nothing tallied
    r2 = b.region_read_off[2]
    b.quals[b.read_off[r2]:b.read_off[r2 + 1]] = 2      # first read of region 2 trimmed to nothing
    res = oracle.run_regions(b)
    assert res.n_contigs_pre[1] == 0 and res.contig_off[2] == res.contig_off[1]
    assert res.n_contigs_pre[2] >= 1
    # an empty batch and a batch with an empty region
    e = b.slice(0, 0)
    assert oracle.run_regions(e).n_contigs == 0
    one = b.slice(0, 1)                                 # a region with a reference window but no reads
    z8, z64 = np.zeros(0, np.uint8), np.zeros(0, np.int64)
    one.region_read_off, one.read_off = np.array([0, 0], np.int64), np.array([0], np.int64)
    one.bases, one.quals, one.mapq, one.read_skip, one.read_start, one.read_stop = z8, z8, z8, z8, z64, z64
    assert oracle.run_regions(one).n_contigs == 0


def test_oracle_reproduces_the_transcript_fixtures(oracle):
    """The committed fixtures of tests/golden/make_transcript_golden.py (both restatements agreed on them in the build
    container) replayed through the C oracle as built here."""
    import golden_util
    assert golden_util.check_transcript(oracle) >= 300


def test_oracle_reproduces_the_deep_fixtures(oracle):
    """tests/golden/deep_golden.npz: 300 regions of 257-600 reads (what gen_roi may hand over, indelope.nim:483-485, :515) whose
    expected results the C oracle and the Python transcription of the Nim sources produced identically (make_deep_golden.py:
    1 200 regions compared, 0 differences, profiles/r06_transcript_deep_diff.txt); the inputs are regenerated and hash-checked."""
    import golden_util
    assert golden_util.check_deep(oracle, threads=8) == 300
