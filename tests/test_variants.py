"""Row f2: post-tally filters, Variant records and VCF lines (indelope.nim:375-428, :604-608, :104-113).

Host code on both sides: the product's C++ (indelope_amd/csrc/variants_host.h, no GPU needed) against the oracle's C
restatement (oracle/oracle_variants.c), over results produced by the oracle here and by the HIP path in the GPU test.
Parity unpinned: the reference has no test for this part and its `kmer.dists` distance is assumed (see the header)."""
import collections
import math

import numpy as np
import pytest

import indelope_amd
from indelope_amd import _abi as A
from indelope_amd import synth


def same_records(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        for k in x:
            u, v = x[k], y[k]
            if isinstance(u, float) and isinstance(v, float) and math.isnan(u) and math.isnan(v):
                continue
            assert u == v, (k, u, v, x["region"], x["event"])


def batches():
    yield synth.generate(250, n_reads=(12, 64), err_rate=1e-3, config_id=71, dup_frac=0.2)[0], dict(min_reads=3, min_ctg_len=73)
    b = synth.generate(200, n_reads=(6, 40), err_rate=3e-3, config_id=72, dup_frac=0.1)[0]
    rng = np.random.default_rng(4)
    b.mapq = rng.choice(np.array([0, 9, 10, 20, 37, 60], np.uint8), b.n_reads)
    yield b, dict()
    yield synth.generate(60, read_len=300, n_reads=(40, 64), err_rate=5e-4, n_events=2, window_len=1400, event_pos=500,
                         config_id=73, K=31)[0], dict(K=31)
    # reads that start at the event: small offsets (LO, CF = 0), short flanks
    yield synth.generate(150, read_len=100, n_reads=(10, 30), err_rate=1e-3, config_id=74, K=21)[0], dict(K=21, min_reads=2)


def test_product_equals_oracle_and_filters_fire(oracle):
    api = indelope_amd.api()
    seen = collections.Counter()
    lines = 0
    for b, kw in batches():
        res = oracle.run_regions(b, oracle.params(**kw))
        exp = oracle.call_variants(b, res, oracle.params(**kw))
        got = api.call_variants(b, res, api.params(**kw))
        same_records(got, exp)
        assert len(got) == int((res.events["status"] == A.IHP_EV_TALLIED).sum())
        seen.update(v["filter"] for v in got)
        for v in got:
            if v["filter"] != A.IHP_VF_EMITTED:
                assert (v["line"] is None) == (v["filter"] != A.IHP_VF_DUPLICATE)
                continue
            lines += 1
            f = v["line"].split("\t")
            assert len(f) == 10 and f[0] == "chr1" and int(f[1]) == v["start"] and f[3] == v["ref"] and f[4] == v["alt"]
            assert f[5] == "%.2f" % v["qual"] and f[6] == "PASS" and f[8] == "GT:GQ:GL"
            assert f[9].split(":")[0] == ["0/0", "0/1", "1/1", "./."][v["gt"]] and v["gt"] != A.IHP_GT_HOM_REF
            info = dict(kv.split("=") if "=" in kv else (kv, True) for kv in f[7].split(";"))
            assert info["AD"] == "%d,%d" % tuple(v["ad"]) and info["CC"] == v["cc"] and int(info["DP"]) == v["dp"]
            assert ("LO" in info) == bool(v["lo"]) and ("AL" in info) == bool(v["al"]) and ("BS" in info) == (v["bs"] > 0)
            # alleles: a deletion keeps the base before it, an insertion adds to it (indelope.nim:413-422)
            if v["event_type"] == 1:
                assert len(v["alt"]) == 1 and v["ref"][0] == v["alt"] and len(v["ref"]) >= 5
            else:
                assert len(v["ref"]) == 1 and len(v["alt"]) >= 5
    assert lines > 300
    # a stricter min_reads for the variant stage alone exercises :375
    b, kw = next(batches())
    res = oracle.run_regions(b, oracle.params(**kw))
    strict = dict(kw, min_reads=25)
    got = api.call_variants(b, res, api.params(**strict))
    same_records(got, oracle.call_variants(b, res, oracle.params(**strict)))
    seen.update(v["filter"] for v in got)
    for f in (A.IHP_VF_EMITTED, A.IHP_VF_LOW_ALT, A.IHP_VF_LOW_FRAC, A.IHP_VF_SMALL_FLANK):
        assert seen[f] > 0, (f, seen)


def test_duplicates_of_the_last_two_printed_are_dropped(oracle):
    """Two regions with the same reads and window give the same variant twice; the second is a duplicate (:604-608)."""
    api = indelope_amd.api()
    b, _ = synth.generate(1, n_reads=(48, 48), config_id=1)
    from indelope_amd.host import RegionBatch
    two = RegionBatch(np.array([0, 48, 96], np.int64), np.concatenate([b.read_off, b.read_off[1:] + b.read_off[-1]]),
                      np.tile(b.bases, 2), np.tile(b.quals, 2), np.tile(b.read_start, 2), np.tile(b.read_stop, 2),
                      np.tile(b.mapq, 2), np.tile(b.read_skip, 2), np.array([0, len(b.ref_bases), 2 * len(b.ref_bases)], np.int64),
                      np.tile(b.ref_bases, 2), np.tile(b.ref_origin, 2))
    res = oracle.run_regions(two)
    got = api.call_variants(two, res)
    same_records(got, oracle.call_variants(two, res))
    assert [v["filter"] for v in got] == [A.IHP_VF_EMITTED, A.IHP_VF_DUPLICATE]


@pytest.mark.gpu
def test_variants_from_device_results(hip, oracle):
    b, kw = next(batches())
    res = hip.run_regions(b, hip.params(**kw))
    got = hip.call_variants(b, res, hip.params(**kw))
    same_records(got, oracle.call_variants(b, oracle.run_regions(b, oracle.params(**kw)), oracle.params(**kw)))
    assert sum(v["filter"] == A.IHP_VF_EMITTED for v in got) > 150


def test_batching_adapter_is_batch_size_independent(oracle):
    """Row f3: the sweep-side adapter (indelope_amd/sweep.py) prints the same VCF lines whatever the flush size, and
    the last-two-printed window (indelope.nim:598-608) works across flushes."""
    from indelope_amd import sweep
    b, _ = synth.generate(90, n_reads=(12, 48), err_rate=1e-3, config_id=75, dup_frac=0.15)
    rois = sweep.rois_from_batch(b)
    # the same region three times in a row, and again two regions later: only the repeats inside the window are dropped
    rois = rois[:10] + [rois[10]] * 3 + rois[11:13] + [rois[10]] + rois[13:]
    kw = dict(min_reads=3, min_ctg_len=73)
    outs = {}
    for n in (1, 2, 7, 64, 10_000):
        for on_host in (True, False):
            c = sweep.BatchedCaller(oracle, oracle.params(**kw), batch_regions=n, trim_on_host=on_host)
            lines = []
            for roi in rois:
                lines += c.add(roi)
            lines += c.flush()
            outs[(n, on_host)] = lines
    ref = outs[(10_000, True)]
    assert len(ref) > 60
    for k, v in outs.items():
        assert v == ref, k
    # one flush == the library's own dedupe over the same regions
    one = sweep.BatchedCaller(oracle, oracle.params(**kw), batch_regions=10_000)
    for roi in rois:
        one.add(roi)
    batch = one._stage()
    res = oracle.run_regions(batch, oracle.params(**kw))
    direct = [v["line"] for v in oracle.call_variants(batch, res, oracle.params(**kw)) if v["filter"] == A.IHP_VF_EMITTED]
    assert direct == ref
    # the repeated region: printed once at its first occurrence, dropped while in the window
    pos = str(int(b.ref_origin[10]))[:4]
    assert sum(l.split("\t")[1].startswith(pos) for l in ref) <= 2


@pytest.mark.gpu
def test_batching_adapter_on_the_device(hip, oracle):
    from indelope_amd import sweep
    b, _ = synth.generate(120, n_reads=(12, 64), err_rate=1e-3, config_id=76, dup_frac=0.1)
    rois = sweep.rois_from_batch(b)
    kw = dict(min_reads=3, min_ctg_len=73)

    def run(api, n):
        c = sweep.BatchedCaller(api, api.params(**kw), batch_regions=n)
        lines = []
        for roi in rois:
            lines += c.add(roi)
        return lines + c.flush()
    exp = run(oracle, 10_000)
    assert len(exp) > 80
    assert run(hip, 10_000) == exp and run(hip, 17) == exp
