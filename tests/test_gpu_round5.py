"""Round 5: the sweeps that carry the load against fixtures produced by the compiled reference C."""
import numpy as np
import pytest

import golden_util

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
