"""Load the committed golden fixtures (tests/golden/*.npz) and replay them through an Api."""
import os

import numpy as np

from indelope_amd.host import BatchResult, RegionBatch

HERE = os.path.dirname(os.path.abspath(__file__))
IN_FIELDS = ("region_read_off", "read_off", "bases", "quals", "read_start", "read_stop", "mapq", "read_skip",
             "ref_off", "ref_bases", "ref_origin")
EZ_FIELDS = ("max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "n_cigar")


class Expected:
    pass


def load_regions(name):
    z = np.load(os.path.join(HERE, "golden", "regions_golden.npz"))
    b = RegionBatch(*[z["%s.in.%s" % (name, f)] for f in IN_FIELDS])
    e = Expected()
    for f in BatchResult.FIELDS:
        setattr(e, f, z["%s.out.%s" % (name, f)])
    return b, int(z["%s.K" % name]), e


def check_regions(api, name):
    b, K, exp = load_regions(name)
    got = api.run_regions(b, api.params(K=K))
    assert BatchResult.first_difference(got, exp) is None, BatchResult.first_difference(got, exp)
    np.testing.assert_allclose(got.events["gl"], exp.events["gl"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(got.events["qual"], exp.events["qual"], rtol=1e-12, atol=1e-12)
    return got


def check_ksw2(api):
    """Every ksw_extz_t field and the full CIGAR against the compiled reference's outputs."""
    z = np.load(os.path.join(HERE, "golden", "ksw2_golden.npz"))
    par = z["params"]
    n = len(par)
    groups = {}
    for i in range(n):
        groups.setdefault(tuple(par[i].tolist()), []).append(i)
    for (gapo, gape, w, zd, flag), idx in groups.items():
        qs = [z["q"][z["q_off"][i]:z["q_off"][i + 1]] for i in idx]
        ts = [z["t"][z["t_off"][i]:z["t_off"][i + 1]] for i in idx]
        ez, cigs = api.align_batch(qs, ts, gap_open=gapo, gap_ext=gape, bw=w, z=zd, flag=flag, encoded=True)
        for k, i in enumerate(idx):
            got = [int(ez[k][f]) for f in EZ_FIELDS]
            assert got == z["ez"][i].tolist(), (i, (gapo, gape, w, zd, flag), got, z["ez"][i].tolist())
            assert cigs[k].tolist() == z["cigar"][z["cigar_off"][i]:z["cigar_off"][i + 1]].tolist(), i
    return n


def load_pair_golden():
    return np.load(os.path.join(HERE, "golden", "ksw2_pair_golden.npz"))


def _case(z, i):
    return (z["q"][z["q_off"][i]:z["q_off"][i + 1]], z["t"][z["t_off"][i]:z["t_off"][i + 1]],
            z["ez"][i].tolist(), z["cigar"][z["cigar_off"][i]:z["cigar_off"][i + 1]].tolist())


def check_ksw2_pair_groups(api, want_pairs=True):
    """tests/golden/ksw2_pair_golden.npz, the pair-sweep part: every group is one ihp_ksw_extz2_batch call whose jobs share a
    scoring scheme, band, z-drop and flag and come in sets of 2-9 equal contig lengths with windows the plan pairs; every
    ksw_extz_t field and CIGAR is the compiled reference's (ksw2_extz2_sse.c:113-388).  Returns (cases, pairs run)."""
    z = load_pair_golden()
    grp = z["group"]
    n_cases = n_pairs = 0
    for g in sorted(set(grp[grp >= 0].tolist())):
        idx = np.nonzero(grp == g)[0]
        ma, mi, go, ge, w, zd, flag = z["params"][idx[0]].tolist()
        assert all(z["params"][i].tolist() == [ma, mi, go, ge, w, zd, flag] for i in idx)
        cs = [_case(z, i) for i in idx]
        ez, cigs = api.align_batch([c[0] for c in cs], [c[1] for c in cs], match=ma, mismatch=mi, gap_open=go, gap_ext=ge,
                                   bw=w, z=zd, flag=flag, encoded=True)
        if want_pairs:
            np_ = api.b.debug_last_ksw_pairs()
            assert np_ > 0, ("no pair formed in group", g)
            n_pairs += np_
        for k, c in enumerate(cs):
            got = [int(ez[k][f]) for f in EZ_FIELDS]
            assert got == c[2], (g, int(idx[k]), (ma, mi, go, ge, w, zd, flag), got, c[2])
            assert cigs[k].tolist() == c[3], (g, int(idx[k]))
        n_cases += len(cs)
    return n_cases, n_pairs


def duo_cases():
    """The fallback part of the file: consecutive cases (2 i, 2 i + 1) share the read; -> (read, t0, t1, expected 0, expected 1)."""
    z = load_pair_golden()
    idx = np.nonzero(z["group"] < 0)[0]
    out = []
    for a, b in zip(idx[0::2], idx[1::2]):
        ca, cb = _case(z, a), _case(z, b)
        assert ca[0].tolist() == cb[0].tolist()
        out.append((ca[0], ca[1], cb[1], (ca[2], ca[3]), (cb[2], cb[3]), z["params"][a].tolist()))
    return out


def check_duo(api):
    """The two-target sweep (ksw_duo.h) against the compiled reference: max, max_q, max_t and the CIGAR of both alignments of every
    item -- what the alignment fallback reads (indelope.nim:343-344).  Returns (items, items the sweep took)."""
    cs = duo_cases()
    par = cs[0][5]
    assert all(c[5] == par for c in cs)
    ma, mi, go, ge, w, zd, flag = par
    ez, cigs = api.duo_batch([c[0] for c in cs], [c[1] for c in cs], [c[2] for c in cs], match=ma, mismatch=mi, gap_open=go, gap_ext=ge,
                             bw=w, z=zd, flag=flag)
    taken = 0
    for i, c in enumerate(cs):
        if len(c[0]) > 320:
            assert int(ez[i][0]["n_cigar"]) == -2
            continue
        for k in range(2):
            exp_ez, exp_cig = c[3 + k]
            e = dict(zip(EZ_FIELDS, exp_ez))
            got = ez[i][k]
            assert int(got["n_cigar"]) != -2, (i, len(c[0]), len(c[1]), len(c[2]))
            assert (int(got["max"]), int(got["max_q"]), int(got["max_t"]), int(got["n_cigar"])) == (e["max"], e["max_q"], e["max_t"], e["n_cigar"]), (i, k, got, e)
            assert cigs[i][k].tolist() == exp_cig, (i, k)
        taken += 1
    return len(cs), taken


def transcript_sets():
    """tests/golden/transcript_golden.npz: regions on which the two independent restatements of the path (the C oracle and
    the Python transcription of the Nim sources, oracle/nim_transcript.py) agreed field by field in the build container;
    -> [(key, RegionBatch, (K, min_reads, min_ctg_len), Expected)]."""
    z = np.load(os.path.join(HERE, "golden", "transcript_golden.npz"))
    keys = sorted({k.split(".")[0] for k in z.files})
    out = []
    for key in keys:
        b = RegionBatch(*[z["%s.in.%s" % (key, f)] for f in IN_FIELDS])
        e = Expected()
        for f in BatchResult.FIELDS:
            setattr(e, f, z["%s.out.%s" % (key, f)])
        out.append((key, b, tuple(int(x) for x in z["%s.params" % key]), e))
    return out


def deep_sets():
    """tests/golden/deep_golden.npz (make_deep_golden.py): 300 regions of 257-600 reads on which both restatements agreed.  The
    inputs come from the repository's own generator (indelope_amd/csrc/synth.cpp) and are checked against the SHA-256 the fixture
    holds; -> [(key, RegionBatch, Expected)]."""
    import hashlib
    from indelope_amd import synth
    z = np.load(os.path.join(HERE, "golden", "deep_golden.npz"))
    out = []
    for key in sorted({k.split(".")[0] for k in z.files}):
        name, first, n = key.rsplit("_", 2)
        b, _ = synth.config(name, n_regions=int(n), first_region=int(first))
        h = hashlib.sha256()
        for f in IN_FIELDS:
            h.update(np.ascontiguousarray(getattr(b, f)).tobytes())
        assert h.hexdigest() == z[key + ".sha256"].tobytes().decode(), "the generator no longer reproduces the inputs of " + key
        e = Expected()
        for f in BatchResult.FIELDS:
            setattr(e, f, z["%s.out.%s" % (key, f)])
        out.append((key, b, e))
    return out


def check_deep(api, trim_bounds=False, threads=None):
    n = 0
    for key, b, exp in deep_sets():
        bt = b.with_trim_bounds() if trim_bounds else b
        got = api.run_regions_mt(bt, api.params(K=27), threads) if threads else api.run_regions(bt, api.params(K=27))
        d = BatchResult.first_difference(got, exp)
        assert d is None, (key, d)
        np.testing.assert_allclose(got.events["gl"], exp.events["gl"], rtol=1e-12, atol=0)
        n += b.n_regions
    return n


def check_transcript(api):
    n = 0
    for key, b, (K, min_reads, min_ctg_len), exp in transcript_sets():
        got = api.run_regions(b, api.params(K=K, min_reads=min_reads, min_ctg_len=min_ctg_len))
        d = BatchResult.first_difference(got, exp)
        assert d is None, (key, d)
        np.testing.assert_allclose(got.events["gl"], exp.events["gl"], rtol=1e-12, atol=0)
        n += b.n_regions
    return n
