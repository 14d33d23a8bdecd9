"""Pin the CPU oracle against every known-answer test the reference holds for the path."""
import numpy as np
import pytest

import kats


@pytest.mark.parametrize("kat", kats.CONTIG_KATS, ids=lambda f: f.__name__)
def test_contig_kats(oracle, kat):
    kat(oracle)


@pytest.mark.parametrize("variant", [0, 1])
def test_ksw2_kat(oracle, variant):
    oracle.set_variant(variant)
    try:
        kats.kat_ksw2(oracle)
    finally:
        oracle.set_variant(0)


def test_genotype_kat(oracle):
    kats.kat_genotype(oracle)


def test_match_sort_order(oracle):
    """contig.nim:345-354: higher matches first, then fewer mismatches; ties keep the first contig."""
    from indelope_amd import Contig
    # two identical targets: the tie must resolve to the lower contig index via best_match; exercised
    # through run_regions in test_oracle_regions.py.  Here: direct slide_align tie -> first offset wins.
    sa = oracle.slide_align("ACGTACGT", "ACGTACGTACGTACGT", min_overlap=5)
    assert sa.offset == 0 and sa.matches == 8


def test_min_overlap_minus_one_is_accepted(oracle):
    """contig.nim:81-82,107: best_ma starts at min_overlap-1 and best_mm at max_mismatch+1, so an
    offset with exactly min_overlap-1 matches and mm <= max_mismatch IS accepted."""
    sa = oracle.slide_align("ACGTT", "GGGGACGTTA", min_overlap=6)
    assert sa.offset == 4 and sa.matches == 5
    sa = oracle.slide_align("ACGTT", "GGGGACGTTAA", min_overlap=7)
    assert not sa.aligned


def test_read_trim(oracle):
    """indelope.nim:23-38."""
    q = np.full(10, 30, np.uint8)
    assert oracle.read_trim(q) == (0, 0, 10)
    q[:3] = 2
    q[8:] = 2
    assert oracle.read_trim(q) == (3, 3, 8)
    q[:] = 2                                   # never reaches a good base: a == high -> emptied
    assert oracle.read_trim(q) == (9, 0, 0)
    q[9] = 30                                  # quirk: only the last base is good -> still emptied (:28)
    assert oracle.read_trim(q) == (9, 0, 0)
    assert oracle.read_trim(np.zeros(0, np.uint8)) == (0, 0, 0)


def test_contig_trim(oracle):
    """contig.nim:49-68."""
    from indelope_amd import Contig
    c = Contig("ACGTACGTAC", 100, 1)
    c._sup[:10] = [1, 1, 3, 3, 3, 3, 3, 1, 1, 1]
    oracle.trim(c, 3)
    assert (c.sequence, c.start, c.support) == ("GTACG", 102, [3] * 5)
    c = Contig("ACGT", 5, 1)
    oracle.trim(c, 3)
    assert (len(c), c.nreads, c.start) == (0, 0, 8)
