"""Round-3 GPU tests: every path switch of the library (ihp_debug_set) against the oracle on C2 / C3 / C5-shaped batches,
C5 at full size and a 20 000-region C3 slice, the RCCL branch of bench.py on the one GPU at hand (group of one: NCCL
process group, per-step gather, payload slabs), stage-time accumulation, and the report page of a freed batch."""
import json
import os
import subprocess
import sys

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


SHAPES = [("C2", 600, 27), ("C3", 400, 27), ("C5", 200, 31)]


@pytest.fixture(scope="module")
def shaped(oracle):
    out = []
    for name, n, K in SHAPES:
        b, _ = synth.config(name, n_regions=n)
        out.append((name, K, b, oracle.run_regions_mt(b, oracle.params(K=K), 16)))
    # a batch with everything the packed path refuses in it (lower case, N, short and long reads) beside clean regions
    b, _ = synth.generate(300, n_reads=(8, 96), err_rate=2e-3, config_id=77, dup_frac=0.2)
    rng = np.random.default_rng(3)
    bases = b.bases.copy()
    idx = rng.integers(0, len(bases), 40)
    bases[idx[:20]] = ord("N")
    bases[idx[20:]] |= 0x20
    b.bases = bases
    out.append(("mixed", 27, b, oracle.run_regions_mt(b, oracle.params(K=27), 16)))
    return out


@pytest.mark.parametrize("knobs", [dict(asm_v1=1), dict(tally_pk=0), dict(lpt=0), dict(no_rich=1), dict(asm_v1=1, tally_pk=0),
                                    dict(comb_occ=8), dict(asm_waves=6, asmr_waves=12, ksw_waves=9, tally_waves=5),
                                    dict(comb_waves=2), dict(comb_waves=4, comb_occ=4), dict(comb_waves=1)],
                         ids=lambda k: ",".join("%s=%d" % kv for kv in k.items()))
def test_results_do_not_depend_on_the_path(hip, shaped, knobs):
    """`asm_v1` forces the byte-based passes on class-1 input, `tally_pk=0` the ASCII tally, `lpt=0` the combine
    launch without cost classes and arena tiers, `no_rich=1` keeps read-rich regions on the byte passes; occupancies are free
    parameters; `comb_waves` = waves per workgroup of the combine kernel (wave 0 runs the region, the others share its
    best_match calls)."""
    hip.debug_set(**knobs)
    try:
        for name, K, b, exp in shaped:
            assert_same(hip.run_regions(b, hip.params(K=K)), exp)
    finally:
        hip.debug_set()


def test_unknown_knob_is_refused(hip):
    with pytest.raises(IhpError) as e:
        hip.debug_set(no_such_switch=1)
    assert e.value.code == A.IHP_E_ARG


def test_c5_full_size_and_c3_20000(hip, oracle):
    """BASELINE configs[4] at full size (10 000 regions of 64 x 300 bp, K = 31) and 20 000 regions of configs[2]
    (16-256 reads per region: every arena tier and assembly class) region by region against the oracle."""
    b, _ = synth.config("C5")
    assert_same(hip.run_regions(b, hip.params(K=31)), oracle.run_regions_mt(b, oracle.params(K=31), 64))
    b, _ = synth.config("C3", n_regions=20_000)
    assert_same(hip.run_regions(b), oracle.run_regions_mt(b, None, 64))


def test_stage_times_add_up_without_reading_them_in_the_loop(hip):
    b, _ = synth.config("C2", n_regions=2000)
    h = hip.batch_upload(b.with_trim_bounds())
    try:
        hip.batch_set_timing(h, True)
        for _ in range(3):
            hip.batch_run(h)
            hip.batch_sync(h)
            hip.batch_sync(h)                                   # a second wait does not count the run twice
        ms, n = hip.batch_kernel_ms_mean(h)
        one = hip.batch_kernel_ms(h)
        assert n == 3 and all(m > 0 for m in ms[:3])
        assert 0.5 * one[0] < ms[0] < 2.0 * one[0]
        ms2, n2 = hip.batch_kernel_ms_mean(h, reset=True)
        assert n2 == 3 and hip.batch_kernel_ms_mean(h)[1] == 0
    finally:
        hip.batch_free(h)


def test_free_while_running_does_not_disturb_the_next_batch(hip, oracle):
    """a batch freed with a run in flight must not hand its report slot to another batch before the run is over."""
    b, _ = synth.config("C2", n_regions=1500)
    small, _ = synth.generate(8, n_reads=(20, 30), err_rate=0.0, config_id=5)
    exp = oracle.run_regions(small)
    for _ in range(6):
        h = hip.batch_upload(b)
        hip.batch_run(h)
        hip.batch_free(h)                                       # run in flight
        h2 = hip.batch_upload(small)
        hip.batch_run(h2)
        hip.batch_sync(h2)                                      # would raise IHP_E_CAPACITY on a spurious overflow flag
        assert_same(hip.batch_fetch(h2), exp)
        hip.batch_free(h2)


def test_profile_after_release_is_an_argument_error(hip):
    b, _ = synth.generate(16, n_reads=(20, 30), err_rate=0.0, config_id=5)
    h = hip.batch_upload(b)
    try:
        hip.batch_run(h)
        hip.batch_sync(h)
        hip.batch_release_outputs(h)
        with pytest.raises(IhpError) as e:
            hip.batch_profile(h)
        assert e.value.code == A.IHP_E_ARG
        hip.batch_run(h)                                        # takes buffers again
        hip.batch_sync(h)
        hip.batch_profile(h)
    finally:
        hip.batch_free(h)


def _bench(args, launcher):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable]
    if launcher:                                                # a fresh child started by torch.distributed.run, as the driver does
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", "29611"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    pr = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert pr.returncode == 0, (pr.stdout[-1500:], pr.stderr[-3000:])
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, pr.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_branch_on_one_gpu_weak():
    """bench.py under torch.distributed.run --nproc-per-node 1 with --force-dist: torch's process group for the launch and the
    barriers, the gather itself through the library's own communicator (round 6: ihp_dist_gather_records / _payload, librccl behind
    the C ABI); rank 0 checks what it gathered against summaries_from_result / its own fetched results."""
    out = _bench(["--gpus", "1", "--force-dist", "--payload", "--verify-gather", "--regions", "3000", "--steps", "3", "--warmup", "1",
                  "--no-cpu", "--no-e2e", "--no-other"], launcher=True)
    g = out["gather_check"]
    # (round 5: the block's one gather carries the records of every region the block processed: steps x regions)
    assert g["backend"].startswith("ihp_dist") and g["backend"].endswith("nccl") and g["records"] == 9000 == g["regions_processed"] and g["records_identical_to_own_results"]
    assert g["payload_identical_to_own_results"] and g["payload_bytes"] > 0
    assert out["oracle_check"]["identical"] and out["n_gpus"] == 1
    assert out["config"]["gather"] == {"per": "block", "records_per_rank_per_gather": 9000, "regions_processed_per_rank_per_gather": 9000,
                                       "bytes_per_rank_per_gather": 9000 * 32}
    # round 3's form (one gather behind every step) is still there for comparison across rounds
    out = _bench(["--gpus", "1", "--force-dist", "--gather-per-step", "--verify-gather", "--regions", "3000", "--steps", "3", "--warmup", "1",
                  "--no-cpu", "--no-e2e", "--no-other", "--no-check"], launcher=True)
    g = out["gather_check"]
    assert g["records"] == 3000 and g["records_identical_to_own_results"] and out["config"]["gather"]["per"] == "step"


def test_rccl_branch_on_one_gpu_strong():
    """The strong-scaling form (C4 regions walked in resident chunks, two in flight, one gather per step) in a group of one."""
    out = _bench(["--force-dist", "--verify-gather", "--scaling", "strong", "--config", "C4", "--regions", "40000", "--chunk", "10000",
                  "--steps", "2", "--warmup", "1", "--no-cpu", "--no-e2e"], launcher=False)
    g = out["gather_check"]
    assert g["backend"].startswith("ihp_dist") and g["records"] == 40000
    assert out["oracle_check"]["identical"] and out["scaling"] == "strong"


def test_slab_upload_with_4bit_bases_gives_the_same_results(hip, shaped, oracle):
    """ihp_batch_upload_slab (one page-locked slab, BAM 4-bit bases, trim bounds) against the separate ASCII arrays of
    ihp_batch_in and against the oracle; IUPAC codes other than ACGT travel too (a read that holds one leaves the packed path)."""
    for name, K, b, exp in shaped:
        if name == "mixed":
            bases = b.bases.copy()
            bases[(bases >= 97) & (bases <= 122)] -= 32                  # BAM has no lower case
            b = type(b)(b.region_read_off, b.read_off, bases, b.quals, b.read_start, b.read_stop, b.mapq, b.read_skip,
                        b.ref_off, b.ref_bases, b.ref_origin)
            exp = oracle.run_regions_mt(b, oracle.params(K=K), 16)
        bt = b.with_trim_bounds()
        slab = hip.make_slab(bt)
        try:
            h = hip.batch_upload_slab(slab, hip.params(K=K))
            try:
                hip.batch_run(h)
                hip.batch_sync(h)
                got = hip.batch_fetch(h)
                hip.batch_set_fetch(h, no_bases=True)
                lean = hip.batch_fetch(h)
            finally:
                hip.batch_free(h)
        finally:
            slab.free()
        assert_same(got, exp)
        # the lean fetch: everything but the contigs' bases and supports
        assert len(lean.ctg_seq) == 0 and len(lean.ctg_support) == 0
        for f in ("status", "n_contigs_pre", "contig_off", "ctg_start", "ctg_nreads", "ctg_seq_off", "aln_flags", "aln_ref_start",
                  "aln_ref_len", "aln_ez", "cigar_off", "cigar", "event_off", "hit_off", "ref_hit", "alt_hit"):
            assert np.array_equal(getattr(lean, f), getattr(got, f)), f
        assert np.array_equal(lean.events, got.events)


def test_a_run_without_the_retry_launches_is_checked_and_repeated(hip, shaped, oracle):
    """ihp_batch_run leaves the roomy combine launch and the byte-based overflow passes out when the last batch OF THE SAME
    SHAPE needed none of them (round 4: the plan is remembered per shape); the wait (sync / fetch / summary) looks at the run's
    counters and repeats the run in full if a region did -- and the batch after that one is not speculated on again."""
    clean, _ = synth.generate(64, n_reads=(30, 40), err_rate=0.0, config_id=9)
    exp_clean = oracle.run_regions(clean)
    # the same batch with a few bases the packed path refuses: same shape, needs the retry route
    import copy
    dirty = copy.copy(clean)
    bases = clean.bases.copy()
    rng = np.random.default_rng(5)
    idx = rng.integers(0, len(bases), 12)
    bases[idx[:6]] = ord("N")
    bases[idx[6:]] |= 0x20
    dirty.bases = bases
    exp_dirty = oracle.run_regions(dirty)
    try:
        for how in ("fetch", "sync", "summary"):
            hip.debug_set()                                     # forgets every plan
            for _ in range(3):                                  # CLEAN_MIN batches in a row leave "nothing of this shape needs the retry route"
                h = hip.batch_upload(clean)                     # (round 6: one such batch is not taken as a promise about the next)
                hip.batch_run(h); hip.batch_sync(h)
                assert_same(hip.batch_fetch(h), exp_clean)
                hip.batch_free(h)
            for attempt in range(2):
                h = hip.batch_upload(dirty)
                try:
                    hip.batch_run(h)
                    if how == "sync":
                        hip.batch_sync(h)
                    elif how == "summary":
                        from indelope_amd.dist import summaries_from_result
                        rec = hip.batch_summary_host(h, dirty.n_regions)
                        assert np.array_equal(rec, summaries_from_result(exp_dirty))
                    prof = hip.batch_profile(h)
                    # lower case / N reads leave the packed path; the first such batch was speculated on and repeated, the second
                    # finds the plan of the repeated run and is not
                    # ([21]: bit 0 = the retry launches were left out of the last run, bit 1 = the roomy ksw2 launch: that one may be)
                    assert prof[23] > 0 and prof[31] == (1 if attempt == 0 else 0) and (prof[21] & 1) == 0, (how, attempt, prof[20:32])
                    assert_same(hip.batch_fetch(h), exp_dirty)
                finally:
                    hip.batch_free(h)
        # the test hook: every run that left the launches out is repeated; results are the same, and so with the switch off
        for kn in (dict(spec_fail=1), dict(no_spec=1)):
            hip.debug_set(**kn)
            for _ in range(5):
                assert_same(hip.run_regions(clean), exp_clean)
            hip.debug_set()
    finally:
        hip.debug_set()


def test_eager_fetch_and_batches_in_flight(hip, shaped, oracle):
    """IHP_FETCH_EAGER (the run counts its results itself, the fetch is one enqueue and one wait) gives what the plain fetch
    gives; ihp_batch_upload_slab only enqueues its copy, so two batches of one thread are in flight before the first is fetched."""
    name, K, b, exp = shaped[0]
    bt = b.with_trim_bounds()
    slabs = [hip.make_slab(bt) for _ in range(2)]
    try:
        hs = []
        for sl in slabs:
            h = hip.batch_upload_slab(sl, hip.params(K=K))
            hip.batch_set_fetch(h, eager=True)
            hip.batch_run(h)
            hs.append(h)
        for h in hs:
            assert_same(hip.batch_fetch(h), exp)                # no ihp_batch_sync in between: the fetch waits
            hip.batch_run(h)                                    # a second run counts again
            hip.batch_sync(h)
            assert_same(hip.batch_fetch(h), exp)
            hip.batch_set_fetch(h, no_bases=True)               # eager off again: the fetch counts by itself
            hip.batch_run(h)
            lean = hip.batch_fetch(h)
            assert len(lean.ctg_seq) == 0 and np.array_equal(lean.events, exp.events)
            hip.batch_release_outputs(h)
            with pytest.raises(IhpError) as e:
                hip.batch_fetch(h)
            assert e.value.code == A.IHP_E_ARG
            hip.batch_free(h)
    finally:
        for sl in slabs:
            sl.free()
    with pytest.raises(IhpError):
        hip.b.batch_set_fetch  # noqa: B018  (binding exists)
        h = hip.batch_upload(bt)
        try:
            hip._chk_hip(hip.b.batch_set_fetch(h, 64), "batch_set_fetch")   # an unknown flag is refused
        finally:
            hip.batch_free(h)


def test_tiers_cut_again_from_the_last_batch_give_the_same_results(hip, oracle):
    """The first combine tier of a run is sized from the histogram the last batch of the same shape left (C5: most regions need
    more than 0.3 x read bases predicts, the second run has one roomier tier): first run, adapted runs and `no_hint` agree."""
    b, _ = synth.config("C5", n_regions=700)
    exp = oracle.run_regions_mt(b, oracle.params(K=31), 16)
    h = hip.batch_upload(b, hip.params(K=31))
    try:
        for _ in range(3):
            hip.batch_run(h)
            hip.batch_sync(h)
            assert_same(hip.batch_fetch(h), exp)
        hip.debug_set(no_hint=1)
        hip.batch_run(h)
        assert_same(hip.batch_fetch(h), exp)
    finally:
        hip.debug_set()
        hip.batch_free(h)


def test_ksw_band_widths_at_the_edges_of_the_sweeps(hip, oracle):
    """Band widths where the narrow sweep changes its plan: 0 (diagonal 1 has no cell: the reference stops there), 1, 47 / 48 /
    49 (an uncut band of 48 is 48 and 47 cells wide in turn: the lean tail loop may only take it once a sequence end cuts it;
    from 49 on the steady loop runs), 62; found by tools/ksw_stress.py, seed 1."""
    import test_gpu_round2 as T
    rng = np.random.default_rng(4848)
    qs, ts = [], []
    for _ in range(120):
        ql, tl = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        q, t = T._pair(rng, ql if rng.random() < 0.6 else min(ql, tl), tl, sub=float(rng.choice([0, 0.01, 0.05, 0.2])),
                       indel=float(rng.choice([0, 0.01, 0.05])), shift=int(rng.integers(0, max(1, tl // 2))) if rng.random() < 0.5 else None)
        qs.append(q)
        ts.append(t)
    for w in (0, 1, 47, 48, 49, 62):
        for flag in (0, A.KSW_EZ_RIGHT, A.KSW_EZ_EXTZ_ONLY):
            for z in (-1, 200):
                kw = dict(match=1, mismatch=-5, gap_open=5, gap_ext=1, bw=w, z=z, flag=flag)
                ez, cg = hip.align_batch(qs, ts, **kw)
                ez2, cg2 = oracle.align_batch(qs, ts, **kw)
                for i in range(len(qs)):
                    assert ez[i].tolist() == ez2[i].tolist() and cg[i].tolist() == cg2[i].tolist(), (kw, len(qs[i]), len(ts[i]), ez[i], ez2[i])


def test_many_contigs_with_mismatches_allowed_in_the_catch_all_pass(hip, oracle):
    """min_overlap_pct = 1.0 with max_mismatch = 1: hardly any read merges, a region of 223 reads leaves 204 contigs and ends up in
    the HBM-arena pass through the general (mismatch-tolerant) insert path.  tools/stress_parity.py with randomised parameters
    found a memory fault there: a new contig's slot kept the ">= 3 supports" run of whatever the slot held before."""
    b, _ = synth.generate(n_regions=93, read_len=151, n_reads=(25, 251), err_rate=0.01, config_id=1102, dup_frac=0.2, seed=2144126653)
    kw = dict(min_ctg_len=73, min_overlap_pct=1.0, max_mismatch=1)
    exp = oracle.run_regions_mt(b, oracle.params(**kw), 16)
    assert int(exp.n_contigs_pre.max()) > 128
    for _ in range(3):
        assert_same(hip.run_regions(b, hip.params(**kw)), exp)
    one = b.slice(3, 4)
    assert_same(hip.run_regions(one, hip.params(**kw)), oracle.run_regions(one, oracle.params(**kw)))


def test_compaction_keeps_room_for_a_trimmed_contig_to_grow_in_place(hip, oracle):
    """A contig whose start a trim moved into the middle of a dword keeps, after a compaction of the packed area, the dwords
    its capacity needs from THAT start: one base added in place later crossed into the next contig's first dword otherwise
    (its first 16 bases read 'AAAA...' and a merge was missed; tools/thread_stress.py, one region in ~100 000)."""
    b, _ = synth.generate(n_regions=340, read_len=150, n_reads=(51, 53), err_rate=0.01, config_id=5618, dup_frac=0.0, seed=1031199230)
    b = b.with_trim_bounds()
    kw = dict(K=27, min_overlap_pct=1.0)
    assert_same(hip.run_regions(b, hip.params(**kw)), oracle.run_regions_mt(b, oracle.params(**kw), 16))
    one = b.slice(102, 103)
    assert_same(hip.run_regions(one, hip.params(**kw)), oracle.run_regions(one, oracle.params(**kw)))


def test_a_z_drop_on_the_last_diagonal_leaves_no_score(hip, oracle):
    """ksw2_extz2_sse.c:355-357: the z-drop test comes before `ez->score = H[tlen-1]`, so a sweep that drops on its very last
    diagonal reports no score.  Identical sequences but for the last base, z-drop below the mismatch penalty (found by
    stress_parity with randomised parameters: bw 20, z-drop 100, once in 3 000 configurations)."""
    rng = np.random.default_rng(12)
    qs, ts = [], []
    for n in (40, 97, 150, 203, 333):
        t = rng.choice(np.frombuffer(b"ACGT", np.uint8), n)
        q = t.copy()
        q[-1] = ord("A") if t[-1] != ord("A") else ord("C")
        qs.append(q)
        ts.append(t)
    for bw in (8, 20, 50, 62, -1):
        for flag in (0, A.KSW_EZ_RIGHT):
            kw = dict(match=1, mismatch=-5, gap_open=5, gap_ext=1, bw=bw, z=4, flag=flag)
            ez, cg = hip.align_batch(qs, ts, **kw)
            ez2, cg2 = oracle.align_batch(qs, ts, **kw)
            assert any(e["zdropped"] == 1 and e["score"] < -(1 << 29) for e in ez2), "the construction no longer drops on the last diagonal"
            for i in range(len(qs)):
                assert ez[i].tolist() == ez2[i].tolist() and cg[i].tolist() == cg2[i].tolist(), (kw, len(qs[i]), ez[i], ez2[i])


def test_compaction_never_moves_a_contig_up(hip, oracle):
    """The second half of the compaction fix: a trimmed contig keeps at most the capacity it has (old cap - trimmed bases, not a
    multiple of four); rounding its length up past that asked for a dword its slot never had, the contigs behind it moved up
    through each other and one lost its first dword (stress_parity seed 90001, configuration 2661: a read-rich region in the
    arena of a batch of smaller ones)."""
    b, _ = synth.generate(n_regions=108, read_len=200, n_reads=(35, 249), err_rate=0.001, config_id=3661, dup_frac=0.6, seed=2102651584)
    assert_same(hip.run_regions(b, hip.params(K=21)), oracle.run_regions_mt(b, oracle.params(K=21), 16))


def test_alignments_too_large_for_the_main_ksw_launch_take_the_roomy_one(hip, oracle):
    """A contig much longer than its reference window needs more LDS / traceback scratch than the main ksw2 launch gives a wave:
    such jobs go to a second, roomy launch of the same kernel instead of failing the batch (IHP_E_CAPACITY before).  Here the
    `ksw_p_cap` switch sends every job there; with the hint that the last batch needed none the launch is left out and the
    wait repeats the run."""
    b, _ = synth.generate(64, n_reads=(32, 48), err_rate=1e-3, config_id=73)
    exp = oracle.run_regions(b)
    clean, _ = synth.generate(16, n_reads=(20, 30), err_rate=0.0, config_id=5)
    try:
        for _ in range(3):                                     # (CLEAN_MIN batches in a row) leave "no job needs the roomy launch"
            h = hip.batch_upload(clean)
            hip.batch_run(h); hip.batch_sync(h); hip.batch_free(h)
        hip.debug_set(ksw_p_cap=4096)
        for _ in range(3):
            h = hip.batch_upload(b)
            hip.batch_run(h)
            hip.batch_sync(h)
            assert_same(hip.batch_fetch(h), exp)
            hip.batch_free(h)
    finally:
        hip.debug_set()
    # reads of 1 500 bases: contigs longer than window + 64 (refused until round 3)
    long_, _ = synth.generate(12, read_len=1500, n_reads=(20, 30), err_rate=1e-3, config_id=11, seed=5)
    assert_same(hip.run_regions(long_), oracle.run_regions(long_))
