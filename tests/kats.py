"""Known-answer tests restated from the reference's own in-file tests.

Each function takes an `api` (indelope_amd.host.Api) so the identical vectors
run against the CPU oracle (tests/test_oracle_*.py, no GPU) and against the HIP
library through its C ABI (tests/test_gpu_kats.py, -m gpu).
"""
import numpy as np

from indelope_amd import Contig, unaligned
from indelope_amd import _abi as A

ALLOW_TEST = A.IHP_ALLOW_SUPPORT     # contig.nim:287-290


# ---- contig.nim:292-430 ------------------------------------------------------
def kat_slide_align_offsets(api):
    sa = api.slide_align("ACTGGGTACGGT", "TTAACTGGGTACGGT", min_overlap=5)            # :293-297
    assert sa.offset == 3 and sa.matches == 12
    assert api.slide_align("ACTGGGTACGGTGGG", "TTAACTGGGTACGGT", min_overlap=5).offset == 3   # :299-301
    assert api.slide_align("ACTGGGTACG", "TTAACTGGGTACGGT", min_overlap=5).offset == 3        # :302-304
    assert api.slide_align("TTAACTGGGTACGGT", "TTAACTGGGTACGGT", min_overlap=5).offset == 0   # :306-308
    assert api.slide_align("ATTAACTGGGTACGGT", "TTAACTGGGTACGGT", min_overlap=5).offset == -1  # :309-311
    assert api.slide_align("ATTAACTGGGTACGGT", "TTAACTGGGTACGGTTTT", min_overlap=5).offset == -1  # :313-314
    assert api.slide_align("ATTAACTGGGTACGGTTTGGGG", "TTAACTGGGTACGGTTTG", min_overlap=5).offset == -1  # :315-317
    assert api.slide_align("ATTAACTGGGTACGGTTTGGGG", "TTAACTGGGTACGGTTTG", min_overlap=50).offset == unaligned  # :319-321


def kat_corrections(api):                                                             # :323-343
    t = Contig("ATTAACTGGGTACGGTTTGGGG", 0, 2)
    q = Contig("TTAACTGGGXACGGTTTGG", 0, 6)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.corrections == []
    q = Contig("TTAACTGGGXACGGTTTGG", 0, 7)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert len(ma.corrections) == 1
    qoff, toff, qbest = ma.corrections[0]
    assert q.sequence[qoff] == "X" and t.sequence[toff] == "T" and qbest
    t = Contig("ATTAACTGGGAACGGTTTGGGG", 0, 7)
    q = Contig("GGAGATTAACTGGGXACGGTTTGG", 0, 2)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert len(ma.corrections) == 1
    qoff, toff, qbest = ma.corrections[0]
    assert q.sequence[qoff] == "X" and t.sequence[toff] == "A" and not qbest


def kat_insert_left_overhang(api):                                                    # :356-389
    t = Contig("ATTAACTGGGTACGGTTTGGGG", 3, 7)
    q = Contig("GGAGATTAACTGGGXACGGTTTGG", 1, 2)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.aligned
    assert (ma.offset, ma.matches, ma.corrections) == (-4, 19, [(14, 10, False)])
    api.insert(t, q, ma)
    assert t.sequence == "GGAGATTAACTGGGTACGGTTTGGGG"
    assert len(t) == 26 and t.start == 1
    assert t.support == [2, 2, 2, 2, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7]

    t = Contig("ATTAACTGGGTACGGTTTGGGG", 5, 2)
    q = Contig("GGAGATTAACTGGGXACGGTTTGG", 0, 7)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    api.insert(t, q, ma)
    assert t.start == 0 and ma.aligned
    assert t.sequence == "GGAGATTAACTGGGXACGGTTTGGGG"
    assert t.support == [7, 7, 7, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 2, 2]

    t = Contig("ATTAACTGGGTAC", 3, 7)
    q = Contig("GGAGATTAACTGGGXACGGTTTGG", 0, 2)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.aligned
    api.insert(t, q, ma)
    assert t.sequence == "GGAGATTAACTGGGTACGGTTTGG"
    assert t.support == [2, 2, 2, 2, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 9, 9, 2, 2, 2, 2, 2, 2, 2]
    assert t.start == 0


def kat_insert_right_overhang(api):                                                   # :391-422
    t = Contig("GGAGATTAACTGGGXACGGTTTGG", 1, 2)
    q = Contig("ATTAACTGGGTACGGTTTGGGG", 3, 7)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.aligned
    api.insert(t, q, ma)
    assert t.start == 1
    assert t.support == [2, 2, 2, 2, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7]
    assert t.sequence == "GGAGATTAACTGGGTACGGTTTGGGG"

    t = Contig("GGAGATTAACTGGGXACGGTTTGG", 90, 7)
    q = Contig("GGAGATTAACTGGGTACGGTTTGGGG", 90, 2)
    assert len(t) == 24
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.offset == 0 and ma.aligned
    api.insert(t, q, ma)
    assert t.start == 90 and len(t) == 26
    assert t.sequence == "GGAGATTAACTGGGXACGGTTTGGGG"

    t = Contig("GGAGATTAACTGGGXACGGTTTGG", 0, 2)
    q = Contig("AAAGGAGATTAACTGGGTACGGTTTGGGG", 3, 7)
    ma = api.slide_align(q, t, min_overlap=5, allowed=ALLOW_TEST)
    assert ma.offset == -3
    api.insert(t, q, ma)
    assert len(t) == len(q)
    assert t.sequence == "AAAGGAGATTAACTGGGTACGGTTTGGGG"
    assert t.start == 3
    assert t.support == [7, 7, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 9, 9, 9, 9, 9, 9, 9, 9, 9, 7, 7]


def kat_insert_contained(api):                                                        # :424-430
    from indelope_amd import Match
    tt = Contig("CCGGGCTGGGCTT", 1, 2)
    qq = Contig("GGCTGGGCT", 1, 2)
    api.insert(tt, qq, Match(matches=19, offset=3, mismatches=0, corrections=[], contig_i=1))
    assert tt.support == [2, 2, 2, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2]


CONTIG_KATS = [kat_slide_align_offsets, kat_corrections, kat_insert_left_overhang,
               kat_insert_right_overhang, kat_insert_contained]

# ---- ksw2.nim:166-216 --------------------------------------------------------
KSW_TGT = ("CGAAACTGGGCTACTCCATGACCAGGGGCAAAATAGGCTTTTAGCCGCTGCGTTCTGGGAGCTCCTCCCCCTTCTGGGAGCTCCTCCCCCTCCCCAGAAGG"
           "CCAAGGGATGTGGGGGCTGGGGGACTGGGAGGCCTGGCAGTCTT")                               # ksw2.nim:171
KSW_QRY = "CGAAACTGGGCTACTCCATGACCAGGGGCAAAATAGGCTTTTAGCCGCTGCGTTCTGGGAGCTCCTCCCCCTCCCCAGAAGGCCAAGGGATGTTGGGG"  # :172


def truncated_cigar(ez, cig):
    """Ez.cigar, ksw2.nim:22-33."""
    out, off, max_off = [], 0, int(ez["max_q"]) & 0xFFFFFFFF
    for w in cig.tolist():
        if off >= max_off:
            break
        op, ln = w & 0xf, w >> 4
        if op != 2:
            off += ln
        out.append((op, ln))
    return out


def kat_ksw2(api):
    assert api.encode(KSW_TGT)[0] == 1 and api.encode(KSW_QRY)[0] == 1                  # :185-189
    assert api.matrix().tolist() == [1, -2, -2, -2, 0, -2, 1, -2, -2, 0, -2, -2, 1, -2, 0,
                                     -2, -2, -2, 1, 0, 0, 0, 0, 0, 0]                   # :191-192
    ez, cigs = api.align_batch([KSW_QRY], [KSW_TGT], gap_open=3, gap_ext=1,
                               flag=A.KSW_EZ_EXTZ_ONLY | A.KSW_EZ_RIGHT)                # :178-180
    cig = truncated_cigar(ez[0], cigs[0])
    assert cig == [(0, 72), (2, 19), (0, 26)]                                          # :194-204
    assert ez[0]["max_q"] + 1 == 98 and ez[0]["max_t"] + 1 == 117                      # :206-208
    assert ez[0]["mqe_t"] == 116                                                       # :210-211
    assert max(ln for op, ln in cig if op != 0) == 19                                  # :213-214
    assert ez[0]["max"] == 73                                                          # SURVEY §8c probe
    # production settings (indelope.nim:221, :576) -- SURVEY §8c probe of the compiled reference
    ez, cigs = api.align_batch([KSW_QRY], [KSW_TGT], gap_open=4, gap_ext=1, bw=50, z=400, flag=0)
    full = "".join("%d%s" % (w >> 4, "MID"[w & 0xf]) for w in cigs[0].tolist())
    assert full == "52M19D46M28D"
    e = ez[0]
    assert (e["max"], e["max_q"], e["max_t"], e["mqe"], e["mqe_t"], e["score"], e["zdropped"]) == \
        (72, 71, 71, 72, 116, 40, 0)
    assert truncated_cigar(e, cigs[0]) == [(0, 52), (2, 19), (0, 46)]


# ---- genotyper.nim:49-67 -----------------------------------------------------
def kat_genotype(api):
    r = api.genotype(10, 10, 1e-4)
    assert r.gt == A.IHP_GT_HET and r.gl[1] > r.gl[0]
    assert api.genotype(20, 0, 1e-4).gt == A.IHP_GT_HOM_REF
    assert api.genotype(1, 19, 1e-2).gt == A.IHP_GT_HOM_ALT
    assert api.genotype(1, 19, 1e-8).gt == A.IHP_GT_HET
    assert api.genotype(0, 0, 1e-8).gt == A.IHP_GT_UNKNOWN
    from indelope_amd.host import GT_STR
    assert GT_STR[api.genotype(1, 19, 1e-8).gt] == "0/1"


def rand_dna(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
