"""Hand-derived vectors for the rows no reference test pins (multi-read `assemble` order effects, both
`combine` passes, `trim`, `Ez.cigar` truncation, the k-mer branches of indelope.nim:244-266).

Every expectation below is a literal derived by hand from the reference source (the derivation is in the comments, with
reference file:line); none is oracle output.  Each function takes an `api` (indelope_amd.host.Api), so the same vectors
run against the CPU oracle (tests/test_hand_vectors.py, no GPU) and against the HIP library (tests/test_gpu_round2.py).
The alignments inside the region vectors are ksw2's, which the compiled reference C pins (tests/test_oracle_ksw2.py): where
a vector depends on one, the full CIGAR and max_q it assumes are asserted first and the truncation / events / k-mers are
then derived by hand from those.
"""
import numpy as np

from indelope_amd import Contig
from indelope_amd import _abi as A
from indelope_amd.host import RegionBatch


def make_batch(regions, qual=30, mapq=60):
    """regions: list of dicts(reads=[(sequence, start)], ref=str, origin=int[, quals=[...], mapqs=[...]])."""
    rro, ro, bases, quals, rs, re_, mq, fo, ref, org = [0], [0], [], [], [], [], [], [0], [], []
    for g in regions:
        for k, (s, st) in enumerate(g["reads"]):
            bases.append(np.frombuffer(s.encode(), np.uint8))
            q = g.get("quals", {}).get(k)
            quals.append(np.full(len(s), qual, np.uint8) if q is None else np.asarray(q, np.uint8))
            ro.append(ro[-1] + len(s))
            rs.append(st)
            re_.append(st + len(s))
            mq.append(g.get("mapqs", {}).get(k, mapq))
        rro.append(len(rs))
        ref.append(np.frombuffer(g["ref"].encode(), np.uint8))
        fo.append(fo[-1] + len(g["ref"]))
        org.append(g["origin"])
    cat = lambda xs: np.concatenate(xs) if xs else np.zeros(0, np.uint8)
    return RegionBatch(np.array(rro, np.int64), np.array(ro, np.int64), cat(bases), cat(quals), np.array(rs, np.int64),
                       np.array(re_, np.int64), np.array(mq, np.uint8), np.zeros(len(rs), np.uint8),
                       np.array(fo, np.int64), cat(ref), np.array(org, np.int64))


def contigs_of(res, r):
    """[(sequence, start, nreads, support list)] of region r."""
    return [(res.contig_sequence(c), int(res.ctg_start[c]), int(res.ctg_nreads[c]), res.contig_support(c).tolist())
            for c in range(res.contig_off[r], res.contig_off[r + 1])]


# ================================================================================================ trim (contig.nim:49-68)
def trim_inner(api):
    # support 1 1 3 4 3 1, min_support 3: a stops at index 2 (first support >= 3, :52-53), start += 2 (:54);
    # b walks down from 5 to 4 (:62-64); kept [2..4] (:66-68)
    c = Contig("ACGTAC", 10)
    c._sup[:6] = [1, 1, 3, 4, 3, 1]
    c.nreads = 4
    api.trim(c, 3)
    assert (c.sequence, c.support, c.start, c.nreads) == ("GTA", [3, 4, 3], 12, 4)


def trim_nothing_left(api):
    # no base reaches min_support: a runs to len-1 = 3 (the loop tests a < len-1, :52), start += 3, a >= len-1 ->
    # sequence and support emptied, nreads 0 (:56-60)
    c = Contig("ACGT", 7)
    c._sup[:4] = [1, 2, 2, 1]
    c.nreads = 3
    api.trim(c, 3)
    assert (c.sequence, c.support, c.start, c.nreads) == ("", [], 10, 0)


def trim_only_last_base_qualifies(api):
    # the loop of :52 never looks at the last base: a = len-1 = 3 although support[3] >= min_support -> emptied all the same
    c = Contig("ACGT", 0)
    c._sup[:4] = [1, 1, 1, 9]
    c.nreads = 9
    api.trim(c, 2)
    assert (c.sequence, c.support, c.start, c.nreads) == ("", [], 3, 0)


def trim_keeps_all(api):
    # every support >= min_support: a = 0, b = len-1: unchanged (:66 is always true; slicing [0..len-1] is the identity)
    c = Contig("ACGTT", 5)
    c._sup[:5] = [2, 2, 5, 2, 2]
    c.nreads = 5
    api.trim(c, 2)
    assert (c.sequence, c.support, c.start, c.nreads) == ("ACGTT", [2, 2, 5, 2, 2], 5, 5)


def trim_single_base(api):
    # len 1: a < len-1 is false at once, a = 0 >= len-1 = 0 -> emptied, start += 0 (:56-60)
    c = Contig("A", 3, 5)
    api.trim(c, 1)
    assert (c.sequence, c.support, c.start, c.nreads) == ("", [], 3, 0)


def trim_min_support_zero(api):
    # min_support 0: support[a] < 0 never holds (uint32): a = 0, b = len-1: unchanged
    c = Contig("ACG", 1)
    c._sup[:3] = [0, 1, 0]
    api.trim(c, 0)
    assert (c.sequence, c.support, c.start) == ("ACG", [0, 1, 0], 1)


TRIM_KATS = [trim_inner, trim_nothing_left, trim_only_last_base_qualifies, trim_keeps_all, trim_single_base, trim_min_support_zero]


# ============================================================ assemble: BAM order decides the contig set (contig.nim:243-248)
S_, X_, Y_, Z_ = "GATTACAGGCTC", "TTGACCTA", "CAGTGGAT", "AAGCTTGC"
RA, RB, RC = X_ + S_, Y_ + S_, S_ + Z_            # 20 bases each; A and B share the suffix S, C starts with S


def order_decides_contigs(api):
    """min_overlap_pct 0.5 -> min_overlap = int(0.5 * 20) = 10 for every read (indelope.nim:169).
    A then B: slide_align(B, A) finds no offset whose whole overlap matches (offset 0 compares X with Y; every other
    offset in 0..10 / -1..-10 mismatches within its first three bases, checked base by base) -> B is a new contig (:248).
    C against [A, B]: offset 8 on either contig overlaps exactly S: 12 matches, 0 mismatches, 12 > best_ma = 9 (:81,:107);
    no other offset matches.  Both matches tie on (matches, mismatches); `matches.sort(match_sort)` is a stable merge sort
    (:239) so the lower contig index wins: C is merged into whichever of A / B came FIRST in BAM order.
    insert (:210-222), offset 8: contig grows to 8 + 20 = 28 bases, support[8..27] += 1 -> 1x8, 2x12, 1x8, nreads 2.
    combine: best_match's default min_overlap 65 (:224) exceeds every length here, so nothing merges in either pass.
    Pass 2 trims each contig with min(nreads, 3) (:267): the 2-read contig with 2 -> bases [8..19] = S, start += 8,
    support 2x12; the 1-read contig with 1 -> unchanged.  n_contigs (pre-combine, indelope.nim:171) = 2."""
    p = api.params(min_overlap_pct=0.5)
    ref = "A" * 40
    b = make_batch([dict(reads=[(RA, 100), (RB, 100), (RC, 108)], ref=ref, origin=90),
                    dict(reads=[(RB, 100), (RA, 100), (RC, 108)], ref=ref, origin=90)])
    res = api.run_regions(b, p)
    assert res.status.tolist() == [0, 0] and res.n_contigs_pre.tolist() == [2, 2]
    assert contigs_of(res, 0) == [(S_, 108, 2, [2] * 12), (RB, 100, 1, [1] * 20)]
    assert contigs_of(res, 1) == [(S_, 108, 2, [2] * 12), (RA, 100, 1, [1] * 20)]
    # nreads < min_reads = 4 (indelope.nim:211): nothing is aligned
    assert (res.aln_flags == 0).all() and res.n_events == 0


# ===================================================== combine: pass 1 merges, pass 2 trims a contig away (contig.nim:254-281)
G40 = "TGCATCGGATACCTGAAGTCCGATTGCAACGTTAGCCAGT"      # 40 bases, no repeat of length >= 4 that could create another overlap
U20 = "CCATGGTTAACGCGTATAGC"


def combine_merges_then_trim_empties(api):
    """Reads of 20 bases, min_overlap = int(0.88 * 20) = 17: R1 = G[0:20], R2 = G[10:30], R3 = G[20:40] overlap their
    neighbours by 10 < 16 matches, so the read phase leaves three contigs; R4 = R5 = R6 = U match each other on 20
    bases at offset 0 and pile up into a fourth contig (support 3x20, nreads 3).  n_contigs = 4.
    combine pass 1 (min_support 0: no trim, :259-260) with combine_min_overlap 8: result = [c0]; c1 against c0 matches at
    offset 10 on 10 bases (10 > best_ma = 7) -> merged: 30 bases, support 1x10 2x10 1x10, nreads 2; c2 against it at
    offset 20 on 10 bases -> 40 bases = G, support 1x10 2x10 2x10 1x10, nreads 3; U matches nothing -> appended.
    Pass 2: trim(min(3,3) = 3) on G: no base has support 3, a runs to len-1 = 39 -> emptied, nreads 0, start += 39
    (:56-60); U (support 3 everywhere) is untouched.  result starts with the first contig that still has reads = U
    (:268-270); the emptied contig aligns to nothing (an empty query has 0 matches at every offset) and has nreads 0, so it
    is dropped (:280).  Final: [U]."""
    p = api.params(combine_min_overlap=8)
    reads = [(G40[0:20], 200), (G40[10:30], 210), (G40[20:40], 220), (U20, 230), (U20, 230), (U20, 230)]
    b = make_batch([dict(reads=reads, ref="A" * 80, origin=190)])
    res = api.run_regions(b, p)
    assert res.status.tolist() == [0] and res.n_contigs_pre.tolist() == [4]
    assert contigs_of(res, 0) == [(U20, 230, 3, [3] * 20)]


def combine_pass1_merge_survives_trim(api):
    """The same chain without R3: pass 1 merges R2 into R1 (offset 10, 10 matches): 30 bases, support 1x10 2x10 1x10,
    nreads 2.  Pass 2 trims with min(2, 3) = 2: a = 10, b = 19 -> G[10:20], start 200 + 10, support 2x10."""
    p = api.params(combine_min_overlap=8)
    b = make_batch([dict(reads=[(G40[0:20], 200), (G40[10:30], 210)], ref="A" * 80, origin=190)])
    res = api.run_regions(b, p)
    assert res.n_contigs_pre.tolist() == [2]
    assert contigs_of(res, 0) == [(G40[10:20], 210, 2, [2] * 10)]


# ========================================================================== votes in combine, both directions (contig.nim:44-47)
GV = "ACGGTCAATGCCTAGGATCCGTTAACGATGCTTGAGCATCGGTAC"      # 45 bases


def votes_fire_both_ways(api):
    """Reads of 25 bases: min_overlap = int(0.88 * 25) = 22, i.e. at least 21 matching bases (best_ma starts at 21, :81).
    Region 0 (the vote changes the TARGET, qbest = true):
      four reads GV[0:25] pile up (support 4, nreads 4); q5 = GV[3:25] + three new bases, the middle one WRONG
      (GV[26] = 'G' replaced by 'A'), matches the contig at offset 3 on 22 bases -> contig Q = 28 bases, support 4x3 5x22
      1x3, nreads 5, Q[26] = 'A'.  Four reads GV[20:45] overlap Q on 8 bases only -> their own contig T (support 4,
      nreads 4).  n_contigs = 2.
      combine pass 1, combine_min_overlap 6: slide_align(q = T, t = Q), offset 20: T[0:8] against Q[20:28] differs at
      T[6] / Q[26]; allowable_mismatch(qsup = 4, tsup = 1, qreads = 4, treads = 5): second clause tsup < 3, qsup > 3*tsup,
      treads > 3*tsup -> allowed (:46-47): correction (6, 26, qbest = 4 > 1), 7 matches > 5, 0 mismatches.
      insert (:161-173): Q[26] := T[6] = 'G' with support 4, position 26 is not incremented afterwards (:217);
      Q grows to 20 + 25 = 45 bases = GV; support 4x3, 5x17, then [20..24] 5+4 = 9, [25] 1+4 = 5, [26] 4, [27] 5, 4x17; nreads 9.
      Pass 2 trims with 3: every support >= 4, unchanged.
    Region 1 (the vote changes the QUERY, qbest = false):
      four reads GV[0:25] -> T2 (support 4, nreads 4).  Four reads GV[20:45] -> contig (support 4); q5' = GV[17:42] with
      position 18 WRONG (GV[18] = 'C' replaced by 'G') matches it at offset -3 on 22 bases (:114-135: q[3:25] on t[0:22]) ->
      Q2 = 28 bases starting at 17, support 1x3 5x22 4x3, nreads 5, start = q5'.start.
      combine pass 1: slide_align(q = Q2, t = T2), offset 17: Q2[1] (support 1) against T2[18] (support 4) differs;
      allowable_mismatch(qsup = 1, tsup = 4, qreads = 5, treads = 4): first clause qsup < 3, tsup > 3*qsup, qreads > 3*qsup
      -> allowed: correction (1, 18, qbest = false).  insert: Q2[1] := T2[18] with support 4 (:168-169), target position
      18 is not incremented; T2 grows to 17 + 28 = 45 bases = GV: support 4x17, [17] 4+1 = 5, [18] 4, [19] 5, [20..24] 9,
      5x17 (new bases 25..41 carry the query's 5), [42..44] 4; nreads 9; start stays T2's."""
    p = api.params(combine_min_overlap=6)
    wrong26 = GV[3:26] + "A" + GV[27]                         # 25 bases: GV[3:28] with position 26 wrong ('G' -> 'A')
    assert GV[26] == "G" and len(wrong26) == 25
    r0 = [(GV[0:25], 500)] * 4 + [(wrong26, 503)] + [(GV[20:45], 520)] * 4
    assert GV[18] == "C"
    wrong18 = GV[17] + "G" + GV[19:42]                        # 25 bases: GV[17:42] with position 18 wrong ('C' -> 'G')
    r1 = [(GV[0:25], 500)] * 4 + [(GV[20:45], 520)] * 4 + [(wrong18, 517)]
    b = make_batch([dict(reads=r0, ref="A" * 100, origin=480), dict(reads=r1, ref="A" * 100, origin=480)])
    res = api.run_regions(b, p)
    assert res.status.tolist() == [0, 0] and res.n_contigs_pre.tolist() == [2, 2]
    sup0 = [4] * 3 + [5] * 17 + [9] * 5 + [5, 4, 5] + [4] * 17
    assert contigs_of(res, 0) == [(GV, 500, 9, sup0)]
    sup1 = [4] * 17 + [5, 4, 5] + [9] * 5 + [5] * 17 + [4] * 3
    assert contigs_of(res, 1) == [(GV, 500, 9, sup1)]


# ============================================================================= Ez.cigar truncation + events (ksw2.nim:22-33, :71-91)
KAT_Q = None        # filled from tests/kats.py (the reference's own ksw2 pair, ksw2.nim:171-172)


def _kat_pair():
    import kats
    return kats.KSW_QRY, kats.KSW_TGT


def cigar_truncation_drops_trailing_deletion(api):
    """The reference's own ksw2 pair (ksw2.nim:171-172) as a contig and its reference window, at production settings
    (gap open 4, ext 1, bw 50, zdrop 400, flag 0): the compiled reference C gives the full CIGAR 52M19D46M28D with
    max_q = 71 -- asserted below.  Ez.cigar (ksw2.nim:22-33): max_off = 71; 52M -> off 52 < 71, yielded;
    19D yielded, off stays 52 (D does not advance, :31); 46M yielded, off = 98; next op: off >= max_off -> break: the
    trailing 28D is dropped.  target_locations(ctg.start) (:71-80): off = start; 52M -> off += 52; 19D -> event
    (start+52, start+52+19, 19, Deletion); query_locations (:82-91): 52M -> off 52; 19D -> (52, 53, 19, Deletion).
    One event of length 19 >= min_event_len 4: k-mers (indelope.nim:236-262), K = 27, width = int((27+1)/2 - 1) = 13:
    tstart = max(0, 52 - 13) = 39 -> ref_kmer = reference[39:66]; qstart = max(52 - 13, 0) = 39 -> ctg[39:66] -- but the
    contig and the window share their first 72 bases (the window holds a 19-base tandem duplication), so alt_kmer ==
    ref_kmer and the retry of :255-262 fires: qstart = max(52 - 3, 0) = 49, 49 + 27 <= 98 -> alt_kmer = ctg[49:76], which
    differs; ref_kmer has four distinct bases (:266).  offset = min(52, 98 - 53 - 1) = 44 (:243).
    Tally (:285-311): each of the 5 reads IS the contig: it holds alt_kmer and, at [39:66], ref_kmer too -> k-mer counts
    ref 5, alt 5, both 5 -> both_found > 0 sends the event to the alignment fallback (:312-372): every read aligns to the
    contig as 98M (count_flanked_cigar = 1) and to the window as 52M 19D 46M (= 3), so :355-356 votes alt: ref 0, alt 5."""
    q, t = _kat_pair()
    assert len(q) == 98 and len(t) == 145
    # five identical reads = the contig (nreads 5 >= min_reads 4, 98 >= min_ctg_len 74); the window request
    # fai.get(ctg.start, max_stop + 13 + 50) (indelope.nim:220) is clamped to the 145-base slice we hand over
    start = 1000
    b = make_batch([dict(reads=[(q, start)] * 5, ref=t, origin=start)])
    res = api.run_regions(b)
    assert res.status.tolist() == [0] and res.n_contigs_pre.tolist() == [1]
    assert contigs_of(res, 0) == [(q, start, 5, [5] * 98)]
    assert res.aln_flags[0] & A.IHP_ALN_DONE and res.aln_ref_len[0] == 145
    assert res.cigar_string(0) == "52M19D46M28D" and res.aln_ez["max_q"][0] == 71          # pinned by the compiled reference
    ev = res.events[res.event_off[0]:res.event_off[1]]
    assert len(ev) == 1
    e = ev[0]
    assert (e["tstart"], e["tstop"], e["qstart"], e["qstop"], e["len"], e["type"]) == (start + 52, start + 71, 52, 53, 19, 1)
    assert e["cf_offset"] == 44
    assert e["ref_kmer"].decode() == t[39:66] and e["alt_kmer"].decode() == q[49:76] and q[39:66] == t[39:66]
    assert e["status"] == A.IHP_EV_TALLIED
    assert (e["kmer_ref_support"], e["kmer_alt_support"], e["kmer_both_found"]) == (5, 5, 5)
    assert (e["fallback_needed"], e["aligned"]) == (1, 1)
    assert (e["ref_support"], e["alt_support"], e["both_found"]) == (0, 5, 0)


# ========================================================================== k-mer choice (indelope.nim:236-281)
R200 = ("TAAACAATCTAGGGCGTTACAGTGATTGTGCGGGTACCCTAAGTCACAATATAAATCGGGCGACCAGCAGCAGTTCAATTCGGACTGGTCAGAAGCACAGCTGGGGACGTATTAGACA"
        "CCGCGGGATGATTGACCTCCCCTAGCTAACTTGCAACAACACTGTCTGAAGCCAAGCTGCTGACATGGGGTGTACTCTTCCT")


def kmer_clamped_at_contig_end(api):
    """Contig = R[0:80] + R[84:97]: a 4-base deletion 13 bases before its end (93 bases; five identical reads).
    ksw2 (pinned): 80M4D13M, max_q = 92 (score 80 - (4 + 4) + 13 = 85 beats the 80 before the gap).  Ez.cigar keeps all
    three ops (off reaches 93 >= 92 only after the last M).  target_locations: D at (start+80, start+84, 4);
    query_locations: (80, 81, 4).  K = 27, width 13: tstart = 80 - 13 = 67 -> ref_kmer = R[67:94];
    qstart = max(80 - 13, 0) = 67, but 67 + 27 = 94 > ctg.len = 93 -> qstart = 93 - 27 = 66 (:248-249):
    alt_kmer = ctg[66:93] = R[66:80] + R[84:97].  They differ; offset = min(80, 93 - 81 - 1) = 11 (:243).
    Tally: the five reads are the contig: alt_kmer yes; ref_kmer R[67:94] spans the deleted bases: no -> (0, 5, 0)."""
    assert len(R200) == 200
    ctg = R200[0:80] + R200[84:97]
    start = 5000
    b = make_batch([dict(reads=[(ctg, start)] * 5, ref=R200, origin=start)])
    res = api.run_regions(b)
    assert contigs_of(res, 0) == [(ctg, start, 5, [5] * 93)]
    assert res.cigar_string(0) == "80M4D13M" and res.aln_ez["max_q"][0] == 92               # pinned by the compiled reference
    ev = res.events[res.event_off[0]:res.event_off[1]]
    assert len(ev) == 1
    e = ev[0]
    assert (e["tstart"], e["tstop"], e["qstart"], e["qstop"], e["len"], e["type"]) == (start + 80, start + 84, 80, 81, 4, 1)
    assert e["ref_kmer"].decode() == R200[67:94] and e["alt_kmer"].decode() == ctg[66:93] and e["cf_offset"] == 11
    assert e["status"] == A.IHP_EV_TALLIED
    assert (e["kmer_ref_support"], e["kmer_alt_support"], e["kmer_both_found"], e["fallback_needed"]) == (0, 5, 0, 0)
    assert (e["ref_support"], e["alt_support"], e["both_found"]) == (0, 5, 0)


F70 = "CATCATTGAAGACTTTACCCAATGTATCCCTGGACGGCTAAATCGGGCGGGTCCACCTGGACTGCTTGGT"      # 70 bases, ends in ...GGT -> 'G' next
TAIL60 = "CGGCTAGAAGCACACCGGGCGAGACCGATCATGACTGATGGACCTAAGGCTGTCTCCCTC"


def kmer_retry_takes_the_end_branch(api):
    """Window = F (70 bases, last base changed to 'G') + ACGT x 4 + TAIL[0:3] (89 bases: the region's reference slice ends
    there, so the window request is clamped to it); contig = F + ACGT x 5 + TAIL[0:3] = 93 bases: one more copy of the unit,
    and the contig ends 23 bases after the repeat starts.
    ksw2 (pinned): 70M4I19M -- the sweep reaches the end (|tlen - qlen| = 4 <= band), so the traceback is the global one;
    the insertion is left-aligned to the start of the repeat (flag 0; it cannot move further left because F ends in 'G',
    not 'T').  max_q = 85: the best score is the UNGAPPED 86 matches (F + 16 repeat bases) at (85, 85), above the gapped
    70 + 19 - 8 = 81 at the end.
    Ez.cigar (max_off = 85): 70M (off 70), 4I (off 74 < 85), 19M (off 93): all three ops are yielded.
    query_locations: I at (70, 74, 4); target_locations: (start+70, start+71, 4).  K = 27, width 13:
    ref_kmer = window[57:84] = F[57:70] + ACGTACGTACGTAC; qstart = 57: ctg[57:84] is the same string (an extra unit inside a
    tandem repeat) -> alt_kmer == ref_kmer -> :255: qstart = 70 - 3 = 67, and 67 + 27 = 94 > ctg.len = 93 -> the `qend`
    branch (:258-260): qend = min(qloc.stop + 4, 93) = 78 -> alt_kmer = ctg[51:78].  Differs from ref_kmer, which has four
    distinct bases.  offset = min(70, 93 - 74 - 1) = 18.
    Tally: a read (= the contig) holds alt_kmer and also ref_kmer (at [57:84]) -> (5, 5, 5) -> alignment fallback: every
    read is 93M on the contig (count_flanked_cigar 1) and 70M 4I 19M on the window (3) -> alt votes: (0, 5)."""
    f = F70[:-1] + "G"
    win = f + "ACGT" * 4 + TAIL60[0:3]
    ctg = f + "ACGT" * 5 + TAIL60[0:3]
    assert len(ctg) == 93 and len(win) == 89 and win[57:84] == ctg[57:84]
    start = 7000
    b = make_batch([dict(reads=[(ctg, start)] * 5, ref=win, origin=start)])
    res = api.run_regions(b)
    assert contigs_of(res, 0) == [(ctg, start, 5, [5] * 93)]
    assert res.cigar_string(0) == "70M4I19M" and res.aln_ez["max_q"][0] == 85               # pinned by the compiled reference
    ev = res.events[res.event_off[0]:res.event_off[1]]
    assert len(ev) == 1
    e = ev[0]
    assert (e["tstart"], e["tstop"], e["qstart"], e["qstop"], e["len"], e["type"]) == (start + 70, start + 71, 70, 74, 4, 0)
    assert e["ref_kmer"].decode() == win[57:84] and e["alt_kmer"].decode() == ctg[51:78] and e["cf_offset"] == 18
    assert e["status"] == A.IHP_EV_TALLIED
    assert (e["kmer_ref_support"], e["kmer_alt_support"], e["kmer_both_found"], e["fallback_needed"], e["aligned"]) == (5, 5, 5, 1, 1)
    assert (e["ref_support"], e["alt_support"], e["both_found"]) == (0, 5, 0)


def kmer_low_complexity_is_skipped(api):
    """Window = F[0:56] + C x 14 + AC x 20 + TAIL[0:20] (130 bases); contig = F[0:56] + C x 14 + AC x 22 + TAIL[0:20] = 134
    bases (two more units).  ksw2 (pinned): 69M4I61M, max_q = 133 (the gapped 130 - 8 = 122 at the end beats the ungapped
    110 before the repeat runs out) -- the left-aligned position is one base before the AC run (the run is preceded by 'C', so
    'ACAC' before the run = 'CACA' one base earlier; the base before that is 'C', not 'A', so it stops there).
    query_locations: I at (69, 73, 4).  ref_kmer = window[56:83] = C x 14 + ACACACACACACA: two distinct bases.
    alt_kmer: ctg[56:83] is the same string -> retry at qstart = 66: ctg[66:93] = CCCC + AC... differs from ref_kmer, so
    :264 does not fire; :266 `ref_kmer.toSet.len < 3` does: the event is skipped (IHP_EV_LOW_CPLX), nothing is tallied."""
    head = F70[0:56]
    win = head + "C" * 14 + "AC" * 20 + TAIL60[0:20]
    ctg = head + "C" * 14 + "AC" * 22 + TAIL60[0:20]
    assert len(ctg) == 134 and len(win) == 130
    start = 9000
    b = make_batch([dict(reads=[(ctg, start)] * 5, ref=win, origin=start)])
    res = api.run_regions(b)
    assert contigs_of(res, 0) == [(ctg, start, 5, [5] * 134)]
    assert res.cigar_string(0) == "69M4I61M" and res.aln_ez["max_q"][0] == 133               # pinned by the compiled reference
    ev = res.events[res.event_off[0]:res.event_off[1]]
    assert len(ev) == 1
    e = ev[0]
    assert (e["qstart"], e["qstop"], e["len"], e["type"]) == (69, 73, 4, 0)
    assert e["status"] == A.IHP_EV_LOW_CPLX
    assert e["ref_kmer"].decode() == "C" * 14 + "ACACACACACACA"


REGION_VECTORS = [order_decides_contigs, combine_merges_then_trim_empties, combine_pass1_merge_survives_trim,
                  votes_fire_both_ways, cigar_truncation_drops_trailing_deletion, kmer_clamped_at_contig_end,
                  kmer_retry_takes_the_end_branch, kmer_low_complexity_is_skipped]


# ================================================================================================ inputs for other tests
def batch_with_one_base_reads():
    """A small synthetic batch in which some reads are cut to a single base (good quality, mapq 60)."""
    from indelope_amd import synth
    b, _ = synth.generate(12, n_reads=(10, 16), err_rate=0.0, config_id=81)
    ln = np.diff(b.read_off)
    cut = np.zeros(b.n_reads, bool)
    cut[[0, 5, 17, b.n_reads - 1]] = True
    newlen = np.where(cut, 1, ln)
    ro = np.concatenate([[0], np.cumsum(newlen)]).astype(np.int64)
    keep = np.concatenate([np.arange(b.read_off[i], b.read_off[i] + newlen[i]) for i in range(b.n_reads)])
    return RegionBatch(b.region_read_off, ro, b.bases[keep], b.quals[keep], b.read_start, b.read_start + newlen, b.mapq,
                       b.read_skip, b.ref_off, b.ref_bases, b.ref_origin)


def _rand_seq(rng, n):
    return "".join(rng.choice(list("ACGT"), n))


def batch_with_long_cigars(n_regions=24):
    """Contigs that differ from their reference window by many short deletions: CIGARs of more than 32 words, which do not
    fit an alignment's fixed slot and go to the shared bump pool."""
    rng = np.random.default_rng(77)
    regions = []
    for _ in range(n_regions):
        ref = _rand_seq(rng, 900)
        pieces, pos = [], 20
        for _ in range(20):                                # 20 x (30 matching bases, 2 deleted)
            pieces.append(ref[pos:pos + 30])
            pos += 32
        ctg = "".join(pieces)
        regions.append(dict(reads=[(ctg, 1020)] * 5, ref=ref, origin=1000))
    return make_batch(regions)


def batch_with_many_events(n_regions=8):
    """Regions with two contigs of three tallied events each (deletions of 8-12 bases): six tallied events, more than a
    region's four fixed hit slots, so the last ones need the shared bump region of the hit pool."""
    rng = np.random.default_rng(78)
    regions = []
    for _ in range(n_regions):
        ref = _rand_seq(rng, 700)
        a = ref[20:120] + ref[128:230] + ref[240:340] + ref[350:450]       # deletions of 8, 10 and 10 bases
        b = ref[20:100] + ref[108:200] + ref[210:300] + ref[312:450]       # deletions of 8, 10 and 12 bases elsewhere
        regions.append(dict(reads=[(a, 1020)] * 5 + [(b, 1020)] * 5, ref=ref, origin=1000))
    return make_batch(regions)
