"""Per-region path of the CPU oracle: golden fixtures, determinism, planted-event recovery."""
import numpy as np
import pytest

import golden_util
from indelope_amd import synth
from indelope_amd import _abi as A
from indelope_amd.host import BatchResult


@pytest.mark.parametrize("name", ["c1", "small", "long"])
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
