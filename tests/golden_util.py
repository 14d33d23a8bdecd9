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
