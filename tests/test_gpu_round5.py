"""Round 5: the sweeps that carry the load against fixtures produced by the compiled reference C."""
import ctypes as C

import numpy as np
import pytest

import golden_util
from indelope_amd import _abi as A
from indelope_amd import synth
from indelope_amd.host import BatchResult, RegionBatch

pytestmark = pytest.mark.gpu


def test_pair_sweep_against_reference_generated_groups(hip):
    """k_ksw_pair (ksw_pair.h) meets the output of src/ksw2/csrc/ksw2_extz2_sse.c:113-388 directly: 447 jobs in 28 batches of
    equal-length groups (w 49 / 50 / 57 / 62, z-drops that fire early / steady / tail, four scoring schemes, wildcards in
    windows), every ksw_extz_t field and CIGAR; each batch must have run pairs."""
    n, pairs = golden_util.check_ksw2_pair_groups(hip)
    assert n >= 400 and pairs >= 150, (n, pairs)
    # the same bytes from the single sweep
    hip.debug_set(ksw_pair=0)
    try:
        n2, _ = golden_util.check_ksw2_pair_groups(hip, want_pairs=False)
        assert hip.b.debug_last_ksw_pairs() == 0 and n2 == n
    finally:
        hip.debug_set()


def test_two_target_sweep_against_the_reference(hip):
    """ksw_duo.h (both alignments of a fallback item in one sweep) against the compiled reference at the fallback's settings
    (gapo 5, unbanded, no z-drop: indelope.nim:318-319, ksw2.nim:159) on (read, window suffix, contig suffix) triples."""
    n, taken = golden_util.check_duo(hip)
    assert n == 120 and taken >= 100, (n, taken)


def test_regions_both_restatements_agree_on(hip):
    """tests/golden/transcript_golden.npz: ~360 regions across the parity parameter space (votes firing in combine, more than 20
    pre-combine contigs, a trailing D, the `alt_kmer == ref_kmer` retry, reads that empty under trim, mapping qualities around the
    three thresholds, the CLI's min_reads / min_ctg_len, K 21 / 27 / 31) whose expected results were produced identically by the C
    oracle and by the Python transcription of contig.nim / indelope.nim:157-372 / ksw2.nim:22-91 written from the Nim alone."""
    assert golden_util.check_transcript(hip) >= 300


def _same(got, exp):
    d = BatchResult.first_difference(got, exp)
    assert d is None, d


def _mixed_batch():
    """Reads with N / IUPAC codes, low mapping qualities, skippable reads, trimmed ends, a window with an N."""
    b, _ = synth.generate(200, n_reads=(8, 96), err_rate=2e-3, config_id=78, dup_frac=0.2)
    rng = np.random.default_rng(4)
    bases = b.bases.copy()
    bases[rng.integers(0, len(bases), 30)] = ord("N")
    q = b.quals.copy()
    for i in rng.integers(0, b.n_reads, 400):
        q[int(b.read_off[i]):int(b.read_off[i]) + int(rng.integers(1, 9))] = 3
    mapq = b.mapq.copy(); mapq[rng.integers(0, b.n_reads, 200)] = rng.choice([0, 5, 9, 19], 200)
    skip = b.read_skip.copy(); skip[rng.integers(0, b.n_reads, 50)] = 1
    return RegionBatch(b.region_read_off, b.read_off, bases, q, b.read_start, b.read_stop, mapq, skip, b.ref_off, b.ref_bases, b.ref_origin)


@pytest.mark.parametrize("which", ["C2", "C3", "C5", "mixed", "mixed_refN", "empty"])
def test_compact_slab_gives_the_same_results(hip, oracle, which):
    """ihp_batch_upload_slab2 (14 bytes per read instead of 34, windows 2 or 4 bits per base; the arrays of
    ihp_batch_in are made on the device by k_slab_expand) against the oracle on the separate arrays, and -- byte for byte --
    against the first slab form."""
    K = 31 if which == "C5" else 27
    if which in ("C2", "C3", "C5"):
        b, _ = synth.config(which, n_regions={"C2": 600, "C3": 300, "C5": 150}[which])
    elif which == "empty":
        b, _ = synth.generate(0)
    else:
        b = _mixed_batch()
        if which == "mixed_refN":
            ref = b.ref_bases.copy(); ref[np.random.default_rng(9).integers(0, len(ref), 5)] = ord("N")
            b = RegionBatch(b.region_read_off, b.read_off, b.bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip, b.ref_off, ref, b.ref_origin)
    exp = oracle.run_regions_mt(b, oracle.params(K=K), 16)
    bt = b.with_trim_bounds()
    s2 = hip.make_slab2(bt)
    try:
        assert bool(s2.flags & A.IHP_SLAB2_REF_2BIT) == (which != "mixed_refN")
        h = hip.batch_upload_slab2(s2, hip.params(K=K))
        try:
            hip.batch_run(h); hip.batch_sync(h)
            got = hip.batch_fetch(h)
            hip.batch_run(h); hip.batch_sync(h)                       # a second run of the resident batch
            again = hip.batch_fetch(h)
        finally:
            hip.batch_free(h)
        n2 = s2.layout.bytes
    finally:
        s2.free()
    _same(got, exp)
    _same(again, exp)
    if b.n_reads:
        s1 = hip.make_slab(bt)
        try:
            assert n2 < (0.8 if which == "C2" else 0.9) * s1.layout.bytes, (n2, s1.layout.bytes)
        finally:
            s1.free()


def test_a_compact_slab_that_is_not_the_librarys_is_refused(hip):
    """The layout is checked against ihp_slab2_layout_for before an offset is followed; unknown flags are refused; lengths that
    do not add up to a region's region_base_off step are caught on the device and every wait reports IHP_E_ARG."""
    from indelope_amd.host import IhpError
    b, _ = synth.config("C2", n_regions=40)
    s2 = hip.make_slab2(b.with_trim_bounds())
    try:
        h = C.c_void_p()
        bad = A.Slab2Layout.from_buffer_copy(s2.layout)
        bad.len += 64
        assert hip.b.batch_upload_slab2(C.byref(hip.params()), s2.n_regions, s2.n_reads, s2.ptr, C.byref(bad), s2.flags, C.byref(h)) == A.IHP_E_ARG
        bad = A.Slab2Layout.from_buffer_copy(s2.layout)
        bad.bytes -= 64
        assert hip.b.batch_upload_slab2(C.byref(hip.params()), s2.n_regions, s2.n_reads, s2.ptr, C.byref(bad), s2.flags, C.byref(h)) == A.IHP_E_ARG
        assert hip.b.batch_upload_slab2(C.byref(hip.params()), s2.n_regions, s2.n_reads, s2.ptr, C.byref(s2.layout), s2.flags | 8, C.byref(h)) == A.IHP_E_ARG
        assert hip.b.batch_upload_slab2(C.byref(hip.params()), s2.n_regions, s2.n_reads + 1, s2.ptr, C.byref(s2.layout), s2.flags, C.byref(h)) == A.IHP_E_ARG
        # one read's length off by one: the prefix sums no longer meet region_base_off
        mem = np.ctypeslib.as_array(C.cast(s2.ptr, C.POINTER(C.c_uint8)), (s2.layout.bytes,))
        ln = mem[s2.layout.len:s2.layout.len + 2 * s2.n_reads].view(np.uint16)
        ln[7] -= 1
        hb = hip.batch_upload_slab2(s2)
        try:
            hip.batch_run(hb)
            with pytest.raises(IhpError) as e:
                hip.batch_sync(hb)
            assert e.value.code == A.IHP_E_ARG
            hip.batch_run(hb)
            with pytest.raises(IhpError):
                hip.batch_sync(hb)                                    # it stays refused
        finally:
            hip.batch_free(hb)
    finally:
        s2.free()


@pytest.mark.parametrize("K", [7, 15, 16, 17, 21, 31])
def test_tally_on_packed_reads_for_every_kmer_width(hip, oracle, K):
    """k_tally rolls the k-mer code in two 32-bit halves and screens every window on the lower one (round 5): K below, at and above
    the 16 bases a half holds, reads of unequal lengths, duplications (both k-mers in a read), against the oracle."""
    b, _ = synth.generate(120, n_reads=(8, 70), err_rate=2e-3, config_id=500 + K, dup_frac=0.3)
    p = dict(K=K, min_ctg_len=40, min_reads=3)
    _same(hip.run_regions(b, hip.params(**p)), oracle.run_regions_mt(b, oracle.params(**p), 16))


@pytest.mark.parametrize("pf", [0, 1, 2])
def test_a_slab_through_either_packing_kernel(hip, oracle, pf):
    """A batch that came as a slab has its bases 4 bits each; k_prepack_fast<U, true> (the software pipeline of round 4, now also
    for that input: the ASCII bases the byte-based kernels read are written on the way) and the plain k_prepack give the oracle's
    results -- reads with N / IUPAC codes, trimmed ends, 300-base reads (more than one dword round per 16-lane group), and a
    batch whose reads hold three bases or none (the pipeline loads without asking)."""
    import copy
    hip.debug_set(prepack_fast=pf)
    try:
        long_reads, _ = synth.config("C5", n_regions=40)
        tiny, _ = synth.generate(6, n_reads=(3, 5), err_rate=0.0, config_id=161)
        tiny = tiny.with_trim_bounds()
        batches = [(_mixed_batch().with_trim_bounds(), 27), (long_reads.with_trim_bounds(), 31)]
        for keep in (0, 3):
            c = copy.copy(tiny)
            nr = c.n_reads
            c.read_off = (np.arange(nr + 1) * keep).astype(np.int64)
            c.bases = np.frombuffer(b"ACG" * nr, np.uint8).copy()[:keep * nr] if keep else np.zeros(0, np.uint8)
            c.quals = np.full(len(c.bases), 40, np.uint8)
            c.read_stop = c.read_start + keep
            c.trim_lo = np.zeros(nr, np.int32); c.trim_hi = np.full(nr, keep, np.int32)
            batches.append((c, 27))
        for b, K in batches:
            exp = oracle.run_regions_mt(b, oracle.params(K=K), 16)
            s2 = hip.make_slab2(b)
            try:
                h = hip.batch_upload_slab2(s2, hip.params(K=K))
                try:
                    hip.batch_run(h); hip.batch_sync(h)
                    _same(hip.batch_fetch(h), exp)
                finally:
                    hip.batch_free(h)
            finally:
                s2.free()
    finally:
        hip.debug_set()


def test_regions_with_more_contigs_than_the_short_table(hip, oracle):
    """The first combine tier runs a build whose contig table holds 32 entries; regions with more pre-combine contigs are filed
    under the second tier, and when the last batch of the shape had more than a hundredth of such regions the first tier runs
    the full-table build instead.  Batch after batch of such regions (the plan changes with the hints) and a batch without any
    in between give the oracle's results."""
    hip.debug_set()
    many, _ = synth.generate(240, n_reads=(120, 200), err_rate=1.2e-2, config_id=41)
    few, _ = synth.generate(240, n_reads=(120, 200), err_rate=0.0, config_id=42)
    exp_many, exp_few = oracle.run_regions_mt(many, oracle.params(K=27), 16), oracle.run_regions_mt(few, oracle.params(K=27), 16)
    assert (exp_many.n_contigs_pre > 32).mean() > 0.05, float((exp_many.n_contigs_pre > 32).mean())
    assert (exp_few.n_contigs_pre > 32).sum() == 0
    for b, exp in ((many, exp_many), (many, exp_many), (few, exp_few), (few, exp_few), (many, exp_many), (many, exp_many)):
        _same(hip.run_regions(b, hip.params(K=27)), exp)


def test_tally_records_and_the_jobs_that_find_the_array_full(hip, oracle):
    """k_tally_prep gathers the header of every job with events into one 128-byte record (a thread per job); k_tally takes the
    records, and the jobs that found the record array full work their header out themselves: the same events, k-mer counts and
    hit positions with room for every record, for three, and for none."""
    b = _mixed_batch()
    exp = oracle.run_regions_mt(b, oracle.params(K=27), 16)
    assert exp.n_events > 50
    try:
        for cap in (0, 3, 1):
            hip.debug_set(tally_rec_cap=cap)
            _same(hip.run_regions(b, hip.params(K=27)), exp)
    finally:
        hip.debug_set()
