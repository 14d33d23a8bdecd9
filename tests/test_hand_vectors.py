"""Hand-derived vectors (tests/hand_vectors.py) against the CPU oracle: they narrow the part of the oracle that no
reference test pins (multi-read assemble, both combine passes, trim, Ez.cigar truncation, k-mer choice)."""
import pytest

import hand_vectors


@pytest.mark.parametrize("kat", hand_vectors.TRIM_KATS, ids=lambda f: f.__name__)
def test_trim_vectors(oracle, kat):
    kat(oracle)


@pytest.mark.parametrize("vec", hand_vectors.REGION_VECTORS, ids=lambda f: f.__name__)
def test_region_vectors(oracle, vec):
    vec(oracle)


def test_helper_batches_are_what_they_claim(oracle):
    """The inputs built for the GPU overflow tests do what their names say (checked on the oracle)."""
    import numpy as np
    b = hand_vectors.batch_with_long_cigars(4)
    res = oracle.run_regions(b)
    assert (res.aln_ez["n_cigar"][res.aln_flags & 1 != 0] > 32).all() and (res.aln_flags & 1).sum() == 4
    b = hand_vectors.batch_with_many_events(3)
    res = oracle.run_regions(b)
    per_region = [(res.events[res.event_off[res.contig_off[r]]:res.event_off[res.contig_off[r + 1]]]["status"] == 0).sum()
                  for r in range(res.n_regions)]
    assert min(per_region) >= 5, per_region            # more than the four fixed hit slots of a region
    b = hand_vectors.batch_with_one_base_reads()
    assert (np.diff(b.read_off) == 1).sum() == 4
    oracle.run_regions(b)
