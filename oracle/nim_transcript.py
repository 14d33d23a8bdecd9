"""TEST INFRASTRUCTURE ONLY -- a SECOND, independent restatement of the per-region path, in plain Python.

Written from the Nim sources alone (src/contig.nim, src/indelope.nim:23-38,157-372, src/ksw2/ksw2.nim:17-91,127-164,
src/genotyper.nim:36-47), line by line, WITHOUT looking at oracle/*.c: the rows of SURVEY.md 8a that no reference test
pins (assemble, combine, Ez.cigar, the event iterators, the k-mer choice, the tally, the alignment fallback's read loop)
then rest on two restatements by way of two languages and two readings of the source instead of one.  It keeps Nim's
semantics where they differ from Python's: inclusive `a..b` ranges, `0..<n`, `setLen` zero fill, the stable merge sort of
`algorithm.sort`, ref-object identity in `c == q`, uint32 supports, `int(float)` truncation, IndexDefect on a slice that
leaves its string (reported as status "oob": the reference would raise).

ksw2 itself is NOT restated here: alignments go through the reference's own C file compiled into oracle/_ref
(oracle/Makefile), which only exists in the build container -- so this module runs there only
(tests/golden/make_transcript_golden.py diffs it against the C oracle and writes fixtures for the GPU suite).
Only tests/ and that script import it; nothing in indelope_amd/ does.
"""
import ctypes as C
import math

UNALIGNED = -(1 << 63)          # low(int), contig.nim:27


# ------------------------------------------------------------------------------------------------ contig.nim
class Contig:                                                    # contig.nim:7-15
    __slots__ = ("sequence", "support", "nreads", "start")

    def __init__(self, sequence, support, nreads, start):
        self.sequence, self.support, self.nreads, self.start = sequence, support, nreads, start

    def __len__(self):                                           # :38
        return len(self.sequence)


class Match:                                                     # :21
    __slots__ = ("matches", "offset", "mismatches", "corrections", "contig_i")

    def __init__(self):
        self.matches, self.offset, self.mismatches, self.corrections, self.contig_i = 0, 0, 0, [], 0

    @property
    def aligned(self):                                           # :29-30
        return self.offset != UNALIGNED


def u32(x):
    return x & 0xffffffff


def allowable_mismatch(qsup, tsup, qreads, treads):              # :44-47
    return ((qsup < 3 and tsup > u32(3 * qsup) and qreads > 3 * qsup) or
            (tsup < 3 and qsup > u32(3 * tsup) and treads > 3 * tsup))


def trim_contig(c, min_support=2):                               # :49-68
    a = 0
    ms = u32(min_support)
    while a < len(c) - 1 and c.support[a] < ms:
        a += 1
    c.start += a
    if a >= len(c) - 1:
        c.sequence = []
        c.support = []
        c.nreads = 0
        return
    b = len(c) - 1
    while c.support[b] < ms and b > a:
        b -= 1
    if a > 0 or b <= len(c) - 1:
        c.support = c.support[a:b + 1]
        c.sequence = c.sequence[a:b + 1]


def slide_align(q, t, min_overlap=50, max_mismatch=0, allowed=allowable_mismatch):   # :70-141
    omin = -(len(q) - min_overlap)
    omax = len(t) - min_overlap
    obest = UNALIGNED
    best_ma = min_overlap - 1
    best_mm = max_mismatch + 1
    best_correction = []
    qs, ts, qsup, tsup = q.sequence, t.sequence, q.support, t.support
    lq, lt = len(qs), len(ts)
    for o in range(0, omax + 1):                                 # for o in 0..omax
        correction = []
        qo, to, mm, ma = 0, o, 0, 0
        while qo < lq and to < lt:
            if qs[qo] != ts[to]:
                if not allowed(qsup[qo], tsup[to], q.nreads, t.nreads):
                    mm += 1
                    if mm > max_mismatch:
                        break
                else:
                    correction.append((qo, to, qsup[qo] > tsup[to]))
            else:
                ma += 1
            qo += 1
            to += 1
        if mm <= max_mismatch and (ma > best_ma or ma == best_ma and mm < best_mm):
            best_ma, best_mm, obest, best_correction = ma, mm, o, correction
    for o in range(1, abs(omin) + 1):                            # for o in 1..abs(omin)
        correction = []
        qo, to, mm, ma = o, 0, 0, 0
        while qo < lq and to < lt:
            if qs[qo] != ts[to]:
                if not allowed(qsup[qo], tsup[to], q.nreads, t.nreads):
                    mm += 1
                    if mm > max_mismatch:
                        break
                else:
                    correction.append((qo, to, qsup[qo] > tsup[to]))
            else:
                ma += 1
            qo += 1
            to += 1
        if mm <= max_mismatch and (ma > best_ma or ma == best_ma and mm < best_mm):
            best_ma, best_mm, obest, best_correction = ma, mm, -o, correction
    m = Match()
    m.matches, m.offset, m.mismatches, m.corrections, m.contig_i = best_ma, obest, best_mm, best_correction, -1
    return m


def make_contig(dna, start, support=1):                          # :143-150
    return Contig(list(dna), [support] * len(dna), int(support), start)


def insert_into(t, q, m):                                        # insert(t, q, m), :156-222
    if not m.aligned:
        return
    dont_overwrite = set()
    for (qoff, toff, qbest) in m.corrections:
        if qbest:
            t.sequence[toff] = q.sequence[qoff]
            t.support[toff] = q.support[qoff]
        else:
            q.sequence[qoff] = t.sequence[toff]
            q.support[qoff] = t.support[toff]
        if m.offset < 0:
            dont_overwrite.add(qoff)
        else:
            dont_overwrite.add(toff)
    if m.offset < 0:
        ao = abs(m.offset)
        tseq = list(q.sequence[0:ao])
        tsup = list(q.support[0:ao])
        tseq.extend(t.sequence)
        tsup.extend(t.support)
        if len(q) > len(tseq):
            d = len(q) - len(tseq)
            tseq.extend(q.sequence[len(q) - d:len(q)])
            tsup.extend([0] * (len(tseq) - len(tsup)))           # tsup.set_len(tseq.len): zero fill
        for i in range(ao, len(q)):
            if i in dont_overwrite:
                continue
            tsup[i] = u32(tsup[i] + q.support[i])
        t.sequence = tseq
        t.support = tsup
        t.nreads += q.nreads
        t.start = q.start
        return
    original_len = len(t)
    if (m.offset + len(q)) > len(t):
        n = m.offset + len(q)
        t.sequence.extend([0] * (n - len(t.sequence)))           # set_len: '\0' fill, overwritten below
        t.support.extend([0] * (n - len(t.support)))
    for i in range(m.offset, min(len(q) + m.offset, len(t))):
        if i in dont_overwrite:
            continue
        qoff = i - m.offset
        t.support[i] = u32(t.support[i] + q.support[qoff])
        if i >= original_len:
            t.sequence[i] = q.sequence[qoff]
    t.nreads += q.nreads


def match_sort_key(m):                                           # :32-36 (a total preorder: a key gives the same stable order)
    return (-m.matches, m.mismatches)


def best_match(contigs, q, min_overlap=65, max_mismatch=0):      # :224-240
    matches = []
    for i, c in enumerate(contigs):
        if c is q:
            continue
        ma = slide_align(q, contigs[i], min_overlap=min_overlap, max_mismatch=max_mismatch)
        if ma.aligned:
            ma.contig_i = i
            matches.append(ma)
    if len(matches) == 0:
        ma = Match()
        ma.offset = UNALIGNED
        return ma
    matches.sort(key=match_sort_key)                             # algorithm.sort: stable merge sort; list.sort is stable
    return matches[0]


def insert_contig(contigs, q, min_overlap=50, max_mismatch=0):   # :243-248
    ma = best_match(contigs, q, min_overlap=min_overlap, max_mismatch=max_mismatch)
    if ma.aligned:
        insert_into(contigs[ma.contig_i], q, ma)
    else:
        contigs.append(q)


def insert_read(contigs, q, start, min_overlap=50, max_mismatch=0):   # :250-252
    insert_contig(contigs, make_contig(q, start), min_overlap=min_overlap, max_mismatch=max_mismatch)


def combine(contigs, max_mismatch=0, min_support=3, again=True):  # :254-281
    if again:
        contigs = combine(contigs, max_mismatch, min_support=0, again=False)
    result = []
    usedi = 0
    for i, c in enumerate(contigs):
        if min_support > 0:
            trim_contig(c, min_support=min(c.nreads, min_support))
        if c.nreads > 0 and len(result) == 0:
            result.append(c)
            usedi = i
    if len(result) == 0:
        return result
    for i in range(0, len(contigs)):
        if i == usedi:
            continue
        ma = best_match(result, contigs[i], max_mismatch=max_mismatch)
        if ma.aligned:
            insert_into(result[ma.contig_i], contigs[i], ma)
        elif contigs[i].nreads > 0:
            result.append(contigs[i])
    return result


# ----------------------------------------------------------------------------------------------- indelope.nim
def trim_read(sequence, base_qualities, min_quality=15):         # :23-38 -> (a, sequence)
    high = len(base_qualities) - 1
    a = 0
    while a < high and base_qualities[a] < min_quality:
        a += 1
    if a == high:
        return a, sequence[0:0]
    b = high
    while b > a and base_qualities[b] < min_quality:
        b -= 1
    if a != 0 or b != high:
        sequence = sequence[a:b + 1]
    return a, sequence


class Read:
    __slots__ = ("seq", "quals", "start", "stop", "mapq", "skip")

    def __init__(self, seq, quals, start, stop, mapq, skip):
        self.seq, self.quals, self.start, self.stop, self.mapq, self.skip = seq, quals, start, stop, mapq, skip


def assemble(reads, min_qual=20, min_overlap_pct=0.88):          # :157-183 -> (contigs, n_contigs)
    contigs = []
    for read in reads:
        if read.mapq < min_qual:
            continue
        if read.skip:
            continue
        o, read_seq = trim_read(read.seq, read.quals)
        insert_read(contigs, read_seq, read.start + o, min_overlap=int(min_overlap_pct * float(len(read_seq))))
    n_contigs = len(contigs)
    contigs = combine(contigs, min_support=3)
    return contigs, n_contigs


# ------------------------------------------------------------------------------------------------- ksw2.nim
LOOKUP = [4] * 256
for _c, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3)):
    LOOKUP[ord(_c)] = _v
    LOOKUP[ord(_c.lower())] = _v                                 # :127 (the table maps both cases)


def encode(dna):                                                 # :129-132
    return bytes(LOOKUP[b] for b in dna)


def matrix(match=1, mismatch=-2):                                # :135-140
    m, x = match, mismatch
    return [m, x, x, x, 0, x, m, x, x, 0, x, x, m, x, 0, x, x, x, m, 0, 0, 0, 0, 0, 0]


class KswExtz(C.Structure):                                      # ksw2.h:22-30
    _fields_ = [("max", C.c_uint32, 31), ("zdropped", C.c_uint32, 1), ("max_q", C.c_int), ("max_t", C.c_int), ("mqe", C.c_int),
                ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int), ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


class Ez:                                                        # ksw2.nim:5-12, new_ez :142-149
    def __init__(self, lib, match=1, mismatch=-2, gap_open=4, gap_ext=1):
        self.lib = lib
        self.c = KswExtz()
        self.mat = (C.c_int8 * 25)(*matrix(match, mismatch))
        self.gap_open, self.gap_ext = abs(gap_open), abs(gap_ext)

    def align_to(self, query, target, flag=0, bw=-1, z=-1):      # :151-164
        q, t = encode(query), encode(target)
        self.c.n_cigar = 0
        qb = (C.c_uint8 * max(1, len(q))).from_buffer_copy(q if len(q) else b"\0")
        tb = (C.c_uint8 * max(1, len(t))).from_buffer_copy(t if len(t) else b"\0")
        self.lib.ksw_extz2_sse(None, len(q), qb, len(t), tb, 5, self.mat, self.gap_open, self.gap_ext, bw, z, flag, C.byref(self.c))

    def full_cigar(self):                                        # :17-20
        return [(self.c.cigar[i] & 0xf, self.c.cigar[i] >> 4) for i in range(self.c.n_cigar)]

    def cigar(self):                                             # :22-33
        out = []
        max_off = self.c.max_q & 0xffffffff                      # uint32(e.c.max_q)
        off = 0
        for i in range(self.c.n_cigar):
            if off >= max_off:
                break
            op, length = self.c.cigar[i] & 0xf, self.c.cigar[i] >> 4
            if op != 2:
                off = u32(off + length)
            out.append((op, length))
        return out

    def target_locations(self, start):                           # :71-80
        off = start
        for op, length in self.cigar():
            if op == 1:
                yield (off, off + 1, length, 0)                  # Insertion
            elif op == 2:
                yield (off, off + length, length, 1)             # Deletion
            if op != 1:
                off += length

    def query_locations(self, start=0):                          # :82-91
        off = start
        for op, length in self.cigar():
            if op == 2:
                yield (off, off + 1, length, 1)
            elif op == 1:
                yield (off, off + length, length, 0)
            if op != 2:
                off += length

    def record(self):
        c = self.c
        return dict(max=c.max, zdropped=c.zdropped, max_q=c.max_q, max_t=c.max_t, mqe=c.mqe, mqe_t=c.mqe_t, mte=c.mte, mte_q=c.mte_q,
                    score=c.score, n_cigar=c.n_cigar)


def load_reference_ksw2(path):
    lib = C.CDLL(path)
    lib.ksw_extz2_sse.restype = None
    lib.ksw_extz2_sse.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int8, C.c_void_p, C.c_int8, C.c_int8,
                                  C.c_int, C.c_int, C.c_int, C.POINTER(KswExtz)]
    return lib


def count_flanked_cigar(ez):                                     # indelope.nim:185-199
    matched = False
    n = 0
    last_op = 0
    for op, _ in ez.cigar():
        if not matched:
            if op == 0:
                n += 1
                matched = True
        else:
            n += 1
        last_op = op
    if last_op != 0:
        n -= 1
    return n


# ------------------------------------------------------------------------------------------------ genotyper.nim
def genotype(r, a, error):                                       # :36-47 -> (GT, GL)
    total = float(r + a)
    gls = [0.0, 0.0, 0.0]
    if total == 0:
        return 3, gls
    gt = 0
    log2 = math.log(2.0)
    for G in range(3):
        gls[G] = -total * log2 + float(r) * math.log(float(G) * error + float(2 - G) * (1 - error)) + float(a) * math.log(float(G) * (1 - error) + float(2 - G) * error)
        if gls[G] > gls[gt]:
            gt = G
    return gt, gls


# ---------------------------------------------------------------------------------------- callsemble, :201-372
COMP = {65: 84, 67: 71, 71: 67, 84: 65}


def revcomp(k):
    return bytes(COMP[b] for b in reversed(k))


def first_hit(seq, K, kmer, rc):
    """Position of the first k-mer of `seq` whose canonical code equals that of `kmer` (indelope.nim:300-309: on upper-case
    ACGT input `e == refe` <=> the read's k-mer is the k-mer or its reverse complement, whatever the package's encoding)."""
    for i in range(0, len(seq) - K + 1):
        w = seq[i:i + K]
        if w == kmer or w == rc:
            return i
    return -1


class Fai:
    """fai.get(chrom, start, stop): 0-based, end-inclusive, clamped to the sequence the way faidx does (the sequence here is
    the region's reference slice at `origin`)."""

    def __init__(self, seq, origin):
        self.seq, self.origin = seq, origin

    def get(self, start, stop):
        b, e = start - self.origin, stop - self.origin
        clamped = b < 0 or e >= len(self.seq)
        b = max(b, 0)
        e = min(e, len(self.seq) - 1)
        return (self.seq[b:e + 1] if e >= b else b""), clamped


def callsemble(reads, fai, ksw_lib, min_ctg_len=74, min_reads=4, min_event_len=4, K=27, fallback=True):
    """indelope.nim:201-372 up to the tallied / voted supports of every event (the filters behind them are row f2).
    Returns the region's record: n_contigs (pre-combine), the final contigs, per contig the alignment and its events."""
    contigs, n_contigs = assemble(reads)
    ez = Ez(ksw_lib)
    out = {"n_pre": n_contigs, "contigs": []}
    for ctg in contigs:
        rec = {"start": ctg.start, "nreads": ctg.nreads, "seq": bytes(ctg.sequence), "support": list(ctg.support), "aligned": False}
        out["contigs"].append(rec)
        if n_contigs > 20:
            continue
        if ctg.nreads < min_reads or len(ctg) < min_ctg_len:
            continue
        max_stop = ctg.start
        for read in reads:
            if read.mapq <= 5:
                continue
            max_stop = max(max_stop, read.stop)
        width = int((K + 1) / 2 - 1)
        reference, clamped = fai.get(ctg.start, max_stop + width + 50)
        ctg_seq = bytes(ctg.sequence)
        ez.align_to(ctg_seq, reference, bw=50, z=400)
        rec.update(aligned=True, clamped=clamped, ref_len=len(reference), ez=ez.record(), full_cigar=ez.full_cigar(), cigar=ez.cigar(), events=[])
        qlocs = list(ez.query_locations())
        if len(qlocs) == 0 or len(qlocs) > 4:
            rec["n_qlocs"] = len(qlocs)
            continue
        tlocs = list(ez.target_locations(ctg.start))
        ii = -1
        for tloc in tlocs:
            ii += 1
            qloc = qlocs[ii]
            ev = {"tstart": tloc[0], "tstop": tloc[1], "qstart": qloc[0], "qstop": qloc[1], "len": tloc[2], "type": tloc[3], "where": None}
            rec["events"].append(ev)
            if tloc[2] < min_event_len:
                ev["where"] = 234
                continue
            tstart = max(0, tloc[0] - ctg.start - width)
            if tstart + K > len(reference):
                tstart = len(reference) - K
            if tstart < 0:
                ev["where"] = "oob"                              # reference[tstart..<tstart+K] raises IndexDefect
                continue
            ref_kmer = reference[tstart:tstart + K]
            offset = min(qloc[0], len(ctg) - qloc[1] - 1)
            qstart = max(qloc[0] - width, 0)
            if qstart + K > len(ctg):
                qstart = len(ctg) - K
            if qstart < 0:
                ev["where"] = "oob"
                continue
            alt_kmer = ctg_seq[qstart:qstart + K]
            if alt_kmer == ref_kmer:
                ev["retried"] = True                             # (for the diff script's coverage count; not a result)
                qstart = max(qloc[0] - 3, 0)
                if qstart + K > len(ctg_seq):
                    qend = min(qloc[1] + 4, len(ctg))
                    if qend - K < 0:
                        ev["where"] = "oob"
                        continue
                    alt_kmer = ctg_seq[qend - K:qend]
                else:
                    alt_kmer = ctg_seq[qstart:qstart + K]
            ev.update(cf_offset=offset, ref_kmer=ref_kmer, alt_kmer=alt_kmer)
            if ref_kmer == alt_kmer and (qloc[0] == 0 or len(set(alt_kmer)) == 1):
                ev["where"] = 264
                continue
            if len(set(ref_kmer)) < 3:
                ev["where"] = 266
                continue
            if ref_kmer == alt_kmer:
                ev["where"] = 268
                continue
            if any(b not in COMP for b in ref_kmer + alt_kmer):
                ev["where"] = "non-acgt"                         # `kmer` package behaviour unknown: nothing claimed
                continue
            rrc, arc = revcomp(ref_kmer), revcomp(alt_kmer)
            alt_support = ref_support = both_found = 0
            ref_hit, alt_hit = [], []
            for read in reads:
                if read.mapq < 10:
                    ref_hit.append(-1)
                    alt_hit.append(-1)
                    continue
                rp = first_hit(read.seq, K, ref_kmer, rrc)
                ap = first_hit(read.seq, K, alt_kmer, arc)
                ref_hit.append(rp)
                alt_hit.append(ap)
                if rp >= 0:
                    ref_support += 1
                if ap >= 0:
                    alt_support += 1
                if rp >= 0 and ap >= 0:
                    both_found += 1
            ev.update(where="tallied", kmer_ref_support=ref_support, kmer_alt_support=alt_support, kmer_both_found=both_found,
                      ref_hit=ref_hit, alt_hit=alt_hit, fallback_needed=both_found > 0, aligned=False)
            if both_found > 0 and fallback:
                both_found = 0
                ez_ref = Ez(ksw_lib, mismatch=-2, gap_open=5, gap_ext=1)
                ez_alt = Ez(ksw_lib, mismatch=-2, gap_open=5, gap_ext=1)
                ref_support = alt_support = 0
                for read in reads:
                    if read.mapq < 10:
                        continue
                    o, read_seq = trim_read(read.seq, read.quals)
                    rs = read.start + o
                    if rs > tloc[1]:
                        continue
                    L = tloc[2] if tloc[3] == 0 else 0
                    if rs + len(read_seq) + L < tloc[0]:
                        continue
                    start = max(rs, ctg.start) - ctg.start
                    ref_sub = reference[start:len(reference)]
                    ctg_sub = ctg_seq[start:len(ctg)]
                    if len(read_seq) == 0 or len(ref_sub) == 0 or len(ctg_sub) == 0:
                        # (encode() of an empty string leaves an empty seq and `query[0].addr` raises: nothing the reference
                        # can vote with; stated, not claimed)
                        ev["degenerate_fallback_read"] = True
                        continue
                    ez_ref.align_to(read_seq, ref_sub)
                    ez_alt.align_to(read_seq, ctg_sub)
                    rn = count_flanked_cigar(ez_ref)
                    an = count_flanked_cigar(ez_alt)
                    if rn == 1 and an > 1:
                        ref_support += 1
                    elif an == 1 and rn > 1:
                        alt_support += 1
                ev["aligned"] = True
            ev.update(ref_support=ref_support, alt_support=alt_support, both_found=both_found)
            gt, gl = genotype(ref_support, alt_support, 1e-3)
            ev.update(gt=gt, gl=gl)
    return out
