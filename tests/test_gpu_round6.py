"""Round 6: the end-of-job gather behind the C ABI (RCCL, ihp_dist_*), the launch-plan and compact-slab regressions of ADVICE r5."""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from indelope_amd import _abi as A
from indelope_amd import dist as idist
from indelope_amd import synth
from indelope_amd.host import BatchResult, RegionBatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(got, exp):
    d = BatchResult.first_difference(got, exp)
    assert d is None, d


# ------------------------------------------------------------------------------------------------ multi-GPU behind the C ABI
def test_gather_in_a_group_of_one_through_the_c_entry_points(hip, oracle):
    """ihp_dist_unique_id / ihp_dist_init / ihp_dist_gather_summaries / ihp_dist_gather_payload / ihp_dist_finalize (dist_host.h:
    ncclCommInitRank, grouped send / recv to the root) on the hardware at hand: the gathered records are the batch's own
    (`summaries_from_result` of its fetched results, what the reference's main loop would walk: indelope.nim:601-608) and the
    gathered payload is what ihp_batch_fetch returns, which is what the oracle produces."""
    b, _ = synth.generate(300, n_reads=(24, 96), err_rate=1e-3, config_id=61, dup_frac=0.1)
    b = b.with_trim_bounds()
    comm = idist.Communicator(hip, 0, 1, lambda mine: mine)
    h = hip.batch_upload(b)
    try:
        assert hip.b.dist_rank(comm.h) == 0 and hip.b.dist_world(comm.h) == 1
        hip.batch_run(h)
        recs, counts = comm.gather_summaries(h, b.n_regions, root=0)          # sizes exchanged (ncclAllGather)
        res = hip.batch_fetch(h)
        want = idist.summaries_from_result(res)
        assert counts.tolist() == [b.n_regions] and len(recs) == b.n_regions
        assert recs.tobytes() == want.tobytes()
        recs2, _ = comm.gather_summaries(h, b.n_regions, root=0, counts=[b.n_regions])   # sizes known to every rank: no exchange
        assert recs2.tobytes() == want.tobytes()
        # the device-pointer form (a caller that keeps the records of several runs in a buffer of its own)
        ptr, n = hip.batch_summary_dev(h)
        recs3, _ = comm.gather_records(ptr, n, root=0, cap=n)
        assert recs3.tobytes() == want.tobytes()
        # a buffer that is too short: IHP_E_CAPACITY, and the communicator still works
        out = np.zeros(4, A.SUMMARY_DTYPE)
        nt = C.c_int64()
        assert hip.b.dist_gather_summaries(comm.h, h, 0, None, out.ctypes.data_as(C.c_void_p), 4, C.byref(nt), None) == A.IHP_E_CAPACITY
        assert nt.value == b.n_regions
        (got,), nbytes = comm.gather_payload(h, root=0)
        assert nbytes[0] > 0
        _same(got, res)
        _same(got, oracle.run_regions(b))
    finally:
        hip.batch_free(h)
        comm.close()


def test_two_ranks_from_c_with_no_python_in_them(hip):
    """tests/abi_harness.c --dist: two fresh processes, each a plain C program over include/indelope_hip.h -- the id travels in a
    file, ihp_init(0) + ihp_dist_init(rank, 2, id), a few regions each, ihp_dist_gather_summaries + ihp_dist_gather_payload to
    rank 0, which checks counts, rank order and the contigs' starts.  This box has ONE GPU, so both ranks bind device 0: RCCL
    either forms the group (then everything must check out) or refuses two ranks on one device at ncclCommInitRank -- in which
    case both ranks must come back with that refusal as IHP_E_HIP and its text, not hang and not crash."""
    from test_abi_exports import _build_harness
    exe = _build_harness()
    with tempfile.TemporaryDirectory() as td:
        idf = os.path.join(td, "id")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        procs = [subprocess.Popen([exe, "--dist", str(r), "2", idf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                o, e = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("the two-rank harness hung")
            outs.append((p.returncode, o, e))
    rc0, o0, e0 = outs[0]
    if rc0 == 0:
        rep = json.loads(o0.strip().splitlines()[-1])
        assert rep["ok"] is True and rep["world"] == 2 and rep["n_total"] == 5, (rep, e0)
        assert outs[1][0] == 0, outs[1]
        print("two ranks on one device: RCCL formed the group; gather verified")
    else:
        # refused: every rank reports it through the ABI's error convention
        for rc, o, e in outs:
            assert rc == 3, (rc, o, e)
            rep = json.loads(o.strip().splitlines()[-1])
            assert rep["dist_init"] == A.IHP_E_HIP and "rccl" in rep["error"], rep
        print("two ranks on one device: RCCL refuses at ncclCommInitRank (%s)" % json.loads(o0.strip().splitlines()[-1])["error"])


# ------------------------------------------------------------------------------------------------ ADVICE r5
def _many_contig_variant(b, region, n_mut=30):
    """The same batch with n_mut reads of one region carrying a substitution in their middle: each becomes a contig of its own in
    the read phase (min_overlap 132 of 150 always covers base 75), more than the 32 the first combine tier's short table holds."""
    bases = b.bases.copy()
    r0, r1 = int(b.region_read_off[region]), int(b.region_read_off[region + 1])
    assert r1 - r0 >= n_mut + 8
    for k, i in enumerate(range(r0 + 4, r0 + 4 + n_mut)):
        p = int(b.read_off[i]) + 60 + (k % 30)
        bases[p] = ord("ACGT"[("ACGT".index(chr(bases[p])) + 1 + k % 3) % 4])
    return RegionBatch(b.region_read_off, b.read_off, bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip, b.ref_off, b.ref_bases, b.ref_origin)


def test_a_rare_many_contig_region_does_not_make_every_other_batch_run_twice(hip, oracle):
    """ADVICE r5 (kernels.h:476): a batch WITHOUT a region of more than 32 contigs lets the next batch's first combine tier walk
    the other tiers' lists with the short-table build; a region of 33..64 contigs in that next batch was then refused by
    v3_take_over, landed on the retry list of a run that had left the retry launches out, and the whole run was repeated.  A stream
    that alternates batches with none and with one such region must not repeat anything: the short table serves a folding first
    tier only after CLEAN_MIN batches of the shape that filed nothing behind it (indelope_hip.hip, TierHint::clean_b / clean_c)."""
    raw, _ = synth.generate(600, n_reads=(64, 64), err_rate=1e-3, config_id=62)
    clean, dirty = raw.with_trim_bounds(), _many_contig_variant(raw, 7).with_trim_bounds()
    exp = {id(clean): oracle.run_regions(clean), id(dirty): oracle.run_regions(dirty)}
    assert 32 < exp[id(dirty)].n_contigs_pre[7] <= 60 and exp[id(clean)].n_contigs_pre.max() <= 32
    hip.debug_set()                                                  # (forgets every launch plan: none of this shape from an earlier test)
    reruns = 0
    for k in range(10):
        bt = dirty if k % 2 else clean
        h = hip.batch_upload(bt)
        try:
            hip.batch_run(h)
            hip.batch_sync(h)
            _same(hip.batch_fetch(h), exp[id(bt)])
            reruns += int(hip.batch_profile(h)[31])
        finally:
            hip.batch_free(h)
    assert reruns == 0, reruns
    # and behind a long clean streak (the short table is back) ONE such batch costs at most its own repeat, after which the plan is safe again
    for k in range(6):
        h = hip.batch_upload(clean)
        hip.batch_run(h); hip.batch_sync(h); hip.batch_free(h)
    seq, rr = [dirty, clean, dirty, clean], []
    for bt in seq:
        h = hip.batch_upload(bt)
        try:
            hip.batch_run(h); hip.batch_sync(h)
            _same(hip.batch_fetch(h), exp[id(bt)])
            rr.append(int(hip.batch_profile(h)[31]))
        finally:
            hip.batch_free(h)
    assert rr[0] <= 1 and rr[1:] == [0, 0, 0], rr


def test_a_malformed_compact_slab_stays_refused_however_many_runs_precede_the_first_wait(hip):
    """ADVICE r5 (indelope_hip.hip:2196): k_slab_expand raises its flag once, at upload; upload, run, RUN, sync used to lose it
    (the second run's k_summary reported the cleared word) and hand back results computed from clamped offsets with rc 0."""
    b, _ = synth.generate(40, n_reads=(16, 32), err_rate=1e-3, config_id=63)
    s2 = hip.make_slab2(b.with_trim_bounds())
    try:
        mem = np.ctypeslib.as_array(C.cast(s2.ptr, C.POINTER(C.c_uint8)), (s2.layout.bytes,))
        mem[s2.layout.len:s2.layout.len + 2 * s2.n_reads].view(np.uint16)[5] -= 1   # a region's lengths no longer add up to its region_base_off step
        h = hip.batch_upload_slab2(s2)
        try:
            hip.batch_run(h)
            hip.batch_run(h)
            assert hip.b.batch_sync(h) == A.IHP_E_ARG
            hip.batch_run(h)
            assert hip.b.batch_sync(h) == A.IHP_E_ARG                  # and it stays refused
            out = A.BatchOut()
            assert hip.b.batch_fetch(h, C.byref(out)) == A.IHP_E_ARG
        finally:
            hip.batch_free(h)
    finally:
        s2.free()


def test_a_soft_masked_window_takes_the_arrays(hip, oracle):
    """ADVICE r5 (nim/indelope_hip.nim fill_slab2): BAM's 4-bit alphabet has no lower case, so a window with soft-masked bases
    (hg19 / hg38) cannot travel in the compact slab -- the slab builders refuse it (Python: ValueError; Nim: nil, before a byte is
    written) and the caller hands over the arrays, whose kernels fold case as ksw2.nim:127-132 does."""
    b, _ = synth.generate(60, n_reads=(24, 48), err_rate=1e-3, config_id=64)
    ref = b.ref_bases.copy()
    for r in range(0, 60, 3):                                          # soft-mask a stretch of every third window
        lo = int(b.ref_off[r]) + 40
        ref[lo:lo + 120] = np.char.lower(ref[lo:lo + 120].view("S1")).view(np.uint8)
    low = RegionBatch(b.region_read_off, b.read_off, b.bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip, b.ref_off, ref, b.ref_origin).with_trim_bounds()
    with pytest.raises(ValueError):
        hip.make_slab2(low)
    _same(hip.run_regions(low), oracle.run_regions(low))


# ------------------------------------------------------------------------------------------------ regions of 257..600 reads
def test_deep_regions_region_by_region(hip, oracle):
    """gen_roi hands over up to 600 reads per roi (indelope.nim:483-485, :515) and the packed path stopped at 256
    (u8 supports, four record registers).  2 000 regions of the `deep` workload (n ~ logU[257, 600], 150 bp) region by region
    against the oracle -- contigs, every support (above 255 on the pile-up's core), ksw2 records, events, k-mer counts -- through
    the wide combine build (asm3_dev.h: 16-bit supports, ten record registers), with the byte-based passes left EMPTY: the
    profile's counters of k_assemble's regions ([3]) stay zero."""
    b, _ = synth.config("deep", n_regions=2000)
    b = b.with_trim_bounds()
    nr = np.diff(b.region_read_off)
    assert nr.min() >= 257 and nr.max() <= 600
    exp = oracle.run_regions_mt(b, oracle.params(K=27), 16)
    assert exp.ctg_support.max() > 255                                  # what a byte cannot hold
    hip.debug_set(profile=1)
    try:
        h = hip.batch_upload(b, hip.params(K=27))
        try:
            for _ in range(2):                                          # (the second run: on the plan the first one left)
                hip.batch_run(h)
                hip.batch_sync(h)
                got = hip.batch_fetch(h)
                _same(got, exp)
                prof = hip.batch_profile(h)
                assert prof[23] == 0, prof[20:32]                        # no region went back to the byte-based passes
        finally:
            hip.batch_free(h)
    finally:
        hip.debug_set()
    # mixed with ordinary regions in one batch, reads in BAM order: the same results whichever launch takes a region
    c3, _ = synth.config("C3", n_regions=400)
    d2, _ = synth.config("deep", n_regions=120)
    from indelope_amd.host import concat_batches
    mix = concat_batches([c3, d2, c3.slice(0, 50)]).with_trim_bounds()
    _same(hip.run_regions(mix), oracle.run_regions_mt(mix, oracle.params(K=27), 16))
    # the same regions at the other configs' error rate: the ones with more than 64 pre-combine contigs take the byte-based passes
    d3, _ = synth.config("deep_1e3", n_regions=150)
    _same(hip.run_regions(d3.with_trim_bounds()), oracle.run_regions_mt(d3.with_trim_bounds(), oracle.params(K=27), 16))


def test_deep_regions_both_restatements_agree_on(hip):
    """tests/golden/deep_golden.npz: 300 regions of 257-600 reads whose expected results the C oracle and the Python transcription
    of the Nim sources produced identically in the build container (1 200 compared, 0 differences); here through the device,
    with the reads' qualities (trim on the device) and with the stager's trim bounds."""
    import golden_util
    assert golden_util.check_deep(hip) == 300
    assert golden_util.check_deep(hip, trim_bounds=True) == 300


# ------------------------------------------------------------------------------------------------ full results in a third of the bytes
def test_compact_fetch_is_the_plain_fetch(hip, oracle):
    """IHP_FETCH_COMPACT: ihp_batch_fetch brings the contigs' bases 4 bits each and their supports a byte
    each with an escape list for the values above 254; ihp_out_contig gives every contig back as the plain fetch holds it, and
    ihp_call_variants (indelope.nim:375-428) on the compact form -- which expands only the contigs an insertion's allele is cut
    from -- returns the records of the plain form.  Workloads: C2-like with duplications, deep pile-ups (supports above 255:
    the escapes), reads with IUPAC codes (the 16-letter alphabet), lower-case reads (no 4-bit code: the plain form comes back)."""
    import ctypes as C
    from indelope_amd import _abi as A
    sets = []
    b, _ = synth.generate(300, n_reads=(24, 96), err_rate=2e-3, config_id=66, dup_frac=0.2)
    sets.append(("dup", b.with_trim_bounds(), True))
    d, _ = synth.config("deep", n_regions=60)
    sets.append(("deep", d.with_trim_bounds(), True))
    n = synth.generate(80, n_reads=(16, 48), err_rate=1e-3, config_id=67)[0]
    bases = n.bases.copy()
    idx = np.random.default_rng(8).integers(0, len(bases), 60)
    bases[idx[:30]] = ord("N"); bases[idx[30:]] = np.frombuffer(b"RYKMSW", np.uint8)[np.arange(30) % 6]
    sets.append(("iupac", RegionBatch(n.region_read_off, n.read_off, bases, n.quals, n.read_start, n.read_stop, n.mapq, n.read_skip, n.ref_off, n.ref_bases, n.ref_origin), True))
    low = bases.copy(); low[idx[:10]] |= 0x20
    sets.append(("lower", RegionBatch(n.region_read_off, n.read_off, low, n.quals, n.read_start, n.read_stop, n.mapq, n.read_skip, n.ref_off, n.ref_bases, n.ref_origin), False))
    for name, bt, want_compact in sets:
        h = hip.batch_upload(bt)
        try:
            hip.batch_run(h)
            plain = hip.batch_fetch(h)
            hip.batch_set_fetch(h, compact=True)
            comp = hip.batch_fetch(h)                                  # expanded contig by contig through ihp_out_contig
            assert comp.compact == want_compact, name
            _same(comp, plain)
            _same(plain, oracle.run_regions_mt(bt, oracle.params(), 8))
            raw = hip.batch_fetch(h, expand=False)
            if name == "deep":
                assert len(raw.sup_escape_idx) > 0 and plain.ctg_support.max() > 255
                assert np.array_equal(plain.ctg_support[raw.sup_escape_idx], raw.sup_escape_val)
            if want_compact:
                # the compact arrays are a third of the plain ones
                assert raw.ctg_seq4.nbytes + raw.ctg_sup8.nbytes + 12 * len(raw.sup_escape_idx) < 0.35 * (plain.ctg_seq.nbytes + plain.ctg_support.nbytes)
                v_plain, v_comp = hip.call_variants(bt, plain), hip.call_variants(bt, raw)
                assert len(v_plain) == len(v_comp) and all(repr(a) == repr(b_) for a, b_ in zip(v_plain, v_comp)), name
        finally:
            hip.batch_free(h)


@pytest.mark.parametrize("which", ["C2", "C5", "deep", "lengths"])
def test_compact_slab_with_the_reads_two_bits_each(hip, oracle, which):
    """IHP_SLAB2_BASES_2BIT: when every read base of a batch is upper-case A C G T the compact slab carries the reads in the
    library's own 2-bit packed form (half the bytes of BAM's 4-bit form; the device packs nothing, k_unpack_pk only writes the
    ASCII copy) -- the same results as the 4-bit form and as the oracle on the separate arrays; one N anywhere and the slab
    builder falls back to 4 bits."""
    K = 31 if which == "C5" else 27
    if which == "lengths":                                             # reads of 1 .. 40 bases around the 16-base word boundaries, some empty
        b, _ = synth.generate(60, read_len=150, n_reads=(20, 40), err_rate=1e-3, config_id=68)
        rng = np.random.default_rng(12)
        ln = rng.choice(np.array([0, 1, 2, 15, 16, 17, 31, 32, 33, 40, 150]), b.n_reads)
        ro = np.zeros(b.n_reads + 1, np.int64); ro[1:] = np.cumsum(ln)
        bases = np.concatenate([b.bases[int(b.read_off[i]):int(b.read_off[i]) + int(ln[i])] for i in range(b.n_reads)] + [np.zeros(0, np.uint8)])
        b = RegionBatch(b.region_read_off, ro, bases, np.full(len(bases), 30, np.uint8), b.read_start, b.read_start + ln, b.mapq, b.read_skip,
                        b.ref_off, b.ref_bases, b.ref_origin)
    else:
        b, _ = synth.config(which, n_regions={"C2": 500, "C5": 120, "deep": 40}[which])
    bt = b.with_trim_bounds()
    exp = oracle.run_regions_mt(bt, oracle.params(K=K), 16)
    sizes = {}
    for form in (None, False):
        s2 = hip.make_slab2(bt, bases_2bit=form)
        try:
            assert bool(s2.flags & A.IHP_SLAB2_BASES_2BIT) == (form is None)
            sizes[form] = s2.layout.bytes
            h = hip.batch_upload_slab2(s2, hip.params(K=K))
            try:
                for _ in range(2):
                    hip.batch_run(h); hip.batch_sync(h)
                    _same(hip.batch_fetch(h), exp)
            finally:
                hip.batch_free(h)
        finally:
            s2.free()
    if which != "lengths":
        assert sizes[None] < 0.66 * sizes[False], sizes
    # a single N: the 4-bit form, whatever was asked for
    bases = bt.bases.copy(); bases[len(bases) // 2] = ord("N")
    bn = RegionBatch(bt.region_read_off, bt.read_off, bases, bt.quals, bt.read_start, bt.read_stop, bt.mapq, bt.read_skip, bt.ref_off, bt.ref_bases, bt.ref_origin,
                     bt.trim_lo, bt.trim_hi)
    s2 = hip.make_slab2(bn)
    try:
        assert not (s2.flags & A.IHP_SLAB2_BASES_2BIT)
        h = hip.batch_upload_slab2(s2, hip.params(K=K))
        try:
            hip.batch_run(h); hip.batch_sync(h)
            _same(hip.batch_fetch(h), oracle.run_regions_mt(bn, oracle.params(K=K), 16))
        finally:
            hip.batch_free(h)
    finally:
        s2.free()


# ------------------------------------------------------------------------------------------------ the fallback's roomy launch
def test_fallback_items_that_do_not_fit_the_main_launch_take_the_roomy_one(hip, oracle):
    """A read of an event whose alignments need more traceback scratch than the alignment fallback's launch gives a wave used to
    refuse its batch (IHP_E_CAPACITY: met by the randomised runs with 960-base reads on a contig far longer than its window).  Now
    k_fallback puts such an item on a list and a second launch of it -- the roomy ksw2 launch's few workgroups, all the LDS, a large
    scratch -- votes for it (indelope.nim:336-356 all the same).  ihp_debug_set("fb_p_cap", n) cuts the main launch's scratch so that
    EVERY item goes that way; a batch that follows three clean runs of its shape (the roomy launches are then left out) is run
    again in full when it turns out to need them."""
    b, _ = synth.generate(48, n_reads=(24, 64), err_rate=1e-3, config_id=67, dup_frac=0.5)
    b = b.with_trim_bounds()
    exp = oracle.run_regions(b)
    assert int((exp.events["aligned"] == 1).sum()) > 5                      # the fallback has work
    hip.debug_set(fb_p_cap=1024)
    try:
        got = hip.run_regions(b)
        _same(got, exp)
        h = hip.batch_upload(b)
        try:
            hip.batch_run(h)
            _same(hip.batch_fetch(h), exp)
            prof = hip.batch_profile(h)
            assert int(prof[47]) > 100 and int(prof[31]) == 0                 # every item took the roomy launch; no hint yet: it was enqueued, no repeat
        finally:
            hip.batch_free(h)
    finally:
        hip.debug_set(fb_p_cap=0)
    # behind a streak of clean runs of the same shape the roomy launches are left out; the next batch that needs them pays one repeat
    clean, _ = synth.generate(48, n_reads=(24, 64), err_rate=1e-3, config_id=68, dup_frac=0.0)
    clean = clean.with_trim_bounds()
    hc = hip.batch_upload(clean)
    try:
        for _ in range(4):
            hip.batch_run(hc)
            hip.batch_sync(hc)
    finally:
        hip.batch_free(hc)
    hip.debug_set(fb_p_cap=1024)
    try:
        h = hip.batch_upload(b)
        try:
            hip.batch_run(h)
            _same(hip.batch_fetch(h), exp)
            prof = hip.batch_profile(h)
            assert int(prof[31]) == 1 and int(prof[47]) > 100                 # left out, found wanting, run again in full
        finally:
            hip.batch_free(h)
    finally:
        hip.debug_set(fb_p_cap=0)
