"""Host-side mirror of the reference's interface for the hot path.

Names, argument meaning and defaults follow the Nim procs they replace
(src/contig.nim, src/ksw2/ksw2.nim, src/genotyper.nim, src/indelope.nim) so the
parity tests read like the reference's own `when isMainModule` tests.  Every
call goes through the C ABI of include/indelope_hip.h; there is no Python or CPU
implementation of the arithmetic in this package.

`Api(bound)` is generic over a bound library: the package default binds the HIP
library (`indelope_amd.api()`); tests bind the CPU oracle to the same class to
compare results call by call.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _abi as A

unaligned = A.IHP_UNALIGNED                      # contig.nim:27


class IhpError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("indelope_hip error %d %s" % (code, what))


class Contig:
    """contig.nim:7-15.  `sequence`/`support` are views of capacity-backed buffers."""

    def __init__(self, dna, start=0, support=1, cap=None):   # make_contig, contig.nim:143-150
        if isinstance(dna, str):
            dna = dna.encode()
        n = len(dna)
        self.cap = int(cap if cap is not None else max(64, 4 * n + 64))
        self._seq = np.zeros(self.cap, np.uint8)
        self._sup = np.zeros(self.cap, np.uint32)
        self._seq[:n] = np.frombuffer(dna, np.uint8)
        self._sup[:n] = support
        self.len, self.nreads, self.start = n, int(support), int(start)

    @property
    def sequence(self):
        return self._seq[:self.len].tobytes().decode("latin1")

    @property
    def support(self):
        return self._sup[:self.len].tolist()

    def __len__(self):
        return self.len

    def _c(self):
        return A.Contig(A.ptr(self._seq, A.u8p), A.ptr(self._sup, A.u32p), self.len, self.cap,
                        self.nreads, self.start)

    def _back(self, c):
        self.len, self.nreads, self.start = c.len, c.nreads, c.start


@dataclass
class Match:                                      # contig.nim:21
    matches: int = 0
    offset: int = unaligned
    mismatches: int = 0
    corrections: list = field(default_factory=list)   # (qoff, toff, qbest)
    contig_i: int = -1

    @property
    def aligned(self):                            # contig.nim:29
        return self.offset != unaligned


@dataclass
class RegionBatch:
    """Flat arrays of include/indelope_hip.h `ihp_batch_in`."""
    region_read_off: np.ndarray
    read_off: np.ndarray
    bases: np.ndarray
    quals: np.ndarray
    read_start: np.ndarray
    read_stop: np.ndarray
    mapq: np.ndarray
    read_skip: np.ndarray
    ref_off: np.ndarray
    ref_bases: np.ndarray
    ref_origin: np.ndarray
    trim_lo: np.ndarray = None        # optional: trim() done by the stager (then `quals` is not needed)
    trim_hi: np.ndarray = None

    @property
    def n_regions(self):
        return len(self.region_read_off) - 1

    @property
    def n_reads(self):
        return len(self.read_off) - 1

    def as_c(self):
        g = lambda a, t, dt: A.ptr(None if a is None else np.ascontiguousarray(a, dt), t)
        keep = []

        def p(a, t, dt):
            if a is None:
                return A.ptr(None, t)
            b = np.ascontiguousarray(a, dt)
            keep.append(b)
            return A.ptr(b, t)
        c = A.BatchIn(self.n_regions, self.n_reads,
                      p(self.region_read_off, A.i64p, np.int64), p(self.read_off, A.i64p, np.int64),
                      p(self.bases, A.u8p, np.uint8), p(self.quals, A.u8p, np.uint8),
                      p(self.read_start, A.i64p, np.int64), p(self.read_stop, A.i64p, np.int64),
                      p(self.mapq, A.u8p, np.uint8), p(self.read_skip, A.u8p, np.uint8),
                      p(self.ref_off, A.i64p, np.int64), p(self.ref_bases, A.u8p, np.uint8),
                      p(self.ref_origin, A.i64p, np.int64),
                      p(self.trim_lo, A.i32p, np.int32), p(self.trim_hi, A.i32p, np.int32))
        c._keep = keep
        return c

    def slice(self, lo, hi):
        """Regions [lo, hi) as an independent batch (used to shard across ranks)."""
        r0, r1 = int(self.region_read_off[lo]), int(self.region_read_off[hi])
        b0, b1 = int(self.read_off[r0]), int(self.read_off[r1])
        f0, f1 = int(self.ref_off[lo]), int(self.ref_off[hi])
        return RegionBatch(self.region_read_off[lo:hi + 1] - r0, self.read_off[r0:r1 + 1] - b0,
                           self.bases[b0:b1], None if self.quals is None else self.quals[b0:b1],
                           self.read_start[r0:r1], self.read_stop[r0:r1], self.mapq[r0:r1],
                           None if self.read_skip is None else self.read_skip[r0:r1],
                           self.ref_off[lo:hi + 1] - f0, self.ref_bases[f0:f1], self.ref_origin[lo:hi],
                           None if self.trim_lo is None else self.trim_lo[r0:r1],
                           None if self.trim_hi is None else self.trim_hi[r0:r1])

    def with_trim_bounds(self, min_quality=15):
        """The same batch with trim(sequence, base_qualities, min_quality) (indelope.nim:23-38) done here, the way a
        stager would: per read the kept range [lo, hi) instead of the qualities."""
        n = self.n_reads
        ln = np.diff(self.read_off).astype(np.int64)
        lo, hi = np.zeros(n, np.int32), ln.astype(np.int32)
        hi[ln == 1] = 0                        # a == high == 0: trim() empties a 1-base read whatever its quality (:28-30)
        if self.quals is not None and n and len(self.quals):
            allgood = int(np.asarray(self.quals).min()) >= min_quality      # the usual case: nothing to trim, no temporaries
            good = None if allgood else np.asarray(self.quals) >= min_quality
            if not allgood:
                idx = np.arange(len(good), dtype=np.int64)
                off = np.asarray(self.read_off[:-1], np.int64)
                nz = ln > 0
                first = np.full(n, np.iinfo(np.int64).max)
                last = np.full(n, -1, np.int64)
                # reduceat over the non-empty reads only (their offsets are strictly increasing)
                o = off[nz]
                first[nz] = np.minimum.reduceat(np.where(good, idx, np.iinfo(np.int64).max), o)
                last[nz] = np.maximum.reduceat(np.where(good, idx, -1), o)
                high = ln - 1
                a = np.where(first - off < high, first - off, high)          # :25 scans i < high only
                emptied = a == high                                         # :28-30 returns a, sequence emptied
                b = np.maximum(last - off, a)                               # :33 last i > a with q >= min, else a
                lo = np.where(nz, a, 0).astype(np.int32)
                hi = np.where(nz, np.where(emptied, a, b + 1), 0).astype(np.int32)
        return RegionBatch(self.region_read_off, self.read_off, self.bases, None, self.read_start, self.read_stop,
                           self.mapq, self.read_skip, self.ref_off, self.ref_bases, self.ref_origin, lo, hi)

    def algorithmic_input_bytes(self):
        """sum_reads(len + 4 start + 1 mapq + 4 trim) + len_refwindow."""
        return int(len(self.bases) + 9 * self.n_reads + len(self.ref_bases))


def _np(ptr_, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr_, C.POINTER(C.c_uint8)), (n * np.dtype(dtype).itemsize,)) \
        .view(dtype).copy()


class BatchResult:
    """numpy copy of `ihp_batch_out` (the C buffers are freed right after)."""
    FIELDS = ("status", "n_contigs_pre", "contig_off", "ctg_start", "ctg_nreads", "ctg_seq_off",
              "ctg_seq", "ctg_support", "aln_flags", "aln_ref_start", "aln_ref_len", "aln_ez",
              "cigar_off", "cigar", "event_off", "events", "hit_off", "ref_hit", "alt_hit")

    def __init__(self, o, expand_with=None):
        R, Cn, E, W, B = o.n_regions, o.n_contigs, o.n_events, o.n_cigar_words, o.n_bases
        self.compact = bool(o.ctg_seq4)                     # IHP_FETCH_COMPACT: 4-bit bases + byte supports arrived
        self.n_regions, self.n_contigs, self.n_events = R, Cn, E
        self.status = _np(o.status, R, np.int32)
        self.n_contigs_pre = _np(o.n_contigs_pre, R, np.int32)
        self.contig_off = _np(o.contig_off, R + 1, np.int64)
        self.ctg_start = _np(o.ctg_start, Cn, np.int64)
        self.ctg_nreads = _np(o.ctg_nreads, Cn, np.int64)
        self.ctg_seq_off = _np(o.ctg_seq_off, Cn + 1, np.int64)
        if self.compact:
            self.ctg_seq4 = _np(o.ctg_seq4, (B + 1) // 2 + Cn + 1, np.uint8)
            self.ctg_sup8 = _np(o.ctg_sup8, B, np.uint8)
            self.sup_escape_idx = _np(o.sup_escape_idx, o.n_sup_escapes, np.int64)
            self.sup_escape_val = _np(o.sup_escape_val, o.n_sup_escapes, np.uint32)
            self.ctg_seq, self.ctg_support = np.zeros(B, np.uint8), np.zeros(B, np.uint32)
            self.expanded = expand_with is not None
            if expand_with is not None:                     # contig by contig through ihp_out_contig (what a host does for the contigs it reads)
                for c in range(Cn):
                    o0, o1 = int(self.ctg_seq_off[c]), int(self.ctg_seq_off[c + 1])
                    if o1 > o0:
                        rc = expand_with(C.byref(o), c, self.ctg_seq[o0:o1].ctypes.data_as(A.u8p), self.ctg_support[o0:o1].ctypes.data_as(A.u32p))
                        if rc != 0:
                            raise IhpError(rc, "ihp_out_contig")
        else:
            self.ctg_seq = _np(o.ctg_seq, B, np.uint8)
            self.ctg_support = _np(o.ctg_support, B, np.uint32)
        self.aln_flags = _np(o.aln_flags, Cn, np.int32)
        self.aln_ref_start = _np(o.aln_ref_start, Cn, np.int64)
        self.aln_ref_len = _np(o.aln_ref_len, Cn, np.int32)
        self.aln_ez = _np(o.aln_ez, Cn, A.EZ_DTYPE)
        self.cigar_off = _np(o.cigar_off, Cn + 1, np.int64)
        self.cigar = _np(o.cigar, W, np.uint32)
        self.event_off = _np(o.event_off, Cn + 1, np.int64)
        self.events = _np(o.events, E, A.EVENT_DTYPE)
        self.n_hits = o.n_hits
        self.hit_off = _np(o.hit_off, E + 1, np.int64)
        self.ref_hit = _np(o.ref_hit, o.n_hits, np.int32)
        self.alt_hit = _np(o.alt_hit, o.n_hits, np.int32)

    def as_c(self):
        """An `ihp_batch_out` view of these arrays (for the host-side entry points that take results as input)."""
        o = A.BatchOut()
        o.n_regions, o.n_contigs, o.n_events = self.n_regions, self.n_contigs, self.n_events
        o.n_cigar_words, o.n_bases, o.n_hits = len(self.cigar), len(self.ctg_seq), len(self.ref_hit)
        keep = []

        def p(a, t):
            b = np.ascontiguousarray(a if len(a) else np.zeros(1, a.dtype))
            keep.append(b)
            return C.cast(b.ctypes.data, t)
        o.status, o.n_contigs_pre, o.contig_off = p(self.status, A.i32p), p(self.n_contigs_pre, A.i32p), p(self.contig_off, A.i64p)
        o.ctg_start, o.ctg_nreads, o.ctg_seq_off = p(self.ctg_start, A.i64p), p(self.ctg_nreads, A.i64p), p(self.ctg_seq_off, A.i64p)
        o.ctg_seq, o.ctg_support = p(self.ctg_seq, A.u8p), p(self.ctg_support, A.u32p)
        if getattr(self, "compact", False) and not getattr(self, "expanded", True):   # the compact form as it arrived (ihp_call_variants expands what it reads)
            o.ctg_seq, o.ctg_support = C.cast(None, A.u8p), C.cast(None, A.u32p)
            o.ctg_seq4, o.ctg_sup8 = p(self.ctg_seq4, A.u8p), p(self.ctg_sup8, A.u8p)
            o.n_sup_escapes = len(self.sup_escape_idx)
            o.sup_escape_idx, o.sup_escape_val = p(self.sup_escape_idx, A.i64p), p(self.sup_escape_val, A.u32p)
        o.aln_flags, o.aln_ref_start, o.aln_ref_len = p(self.aln_flags, A.i32p), p(self.aln_ref_start, A.i64p), p(self.aln_ref_len, A.i32p)
        o.aln_ez, o.cigar_off, o.cigar = p(self.aln_ez, C.POINTER(A.Ez)), p(self.cigar_off, A.i64p), p(self.cigar, A.u32p)
        o.event_off, o.events = p(self.event_off, A.i64p), p(self.events, C.POINTER(A.Event))
        o.hit_off, o.ref_hit, o.alt_hit = p(self.hit_off, A.i64p), p(self.ref_hit, A.i32p), p(self.alt_hit, A.i32p)
        o._keep = keep
        return o

    def contig_sequence(self, c):
        return self.ctg_seq[self.ctg_seq_off[c]:self.ctg_seq_off[c + 1]].tobytes().decode("latin1")

    def contig_support(self, c):
        return self.ctg_support[self.ctg_seq_off[c]:self.ctg_seq_off[c + 1]]

    def cigar_string(self, c):
        w = self.cigar[self.cigar_off[c]:self.cigar_off[c + 1]]
        return "".join("%d%s" % (x >> 4, "MID"[x & 0xf]) for x in w.tolist())

    def algorithmic_output_bytes(self, K):
        """sum_contigs(5 len + 16) + sum_aln(44 + 4 n_cigar) + sum_events(2K + 12)."""
        naln = int((self.aln_flags & A.IHP_ALN_DONE != 0).sum())
        return int(5 * len(self.ctg_seq) + 16 * self.n_contigs + 44 * naln + 4 * len(self.cigar)
                   + (2 * K + 12) * self.n_events)

    def algorithmic_bytes_by_kernel(self, batch, K):
        """The same SURVEY.md §8d terms split by the kernel that moves them: assemble = reads in
        + final contigs out; ksw2 = contig + window in, result record + CIGAR out; tally = the region's read bases
        once per tallied event + the event record out."""
        done = self.aln_flags & A.IHP_ALN_DONE != 0
        clen = np.diff(self.ctg_seq_off)
        asm = int(len(batch.bases) + 9 * batch.n_reads + 5 * len(self.ctg_seq) + 16 * self.n_contigs)
        ksw = int(clen[done].sum() + self.aln_ref_len[done].sum() + 44 * int(done.sum()) + 4 * len(self.cigar))
        region_bases = np.diff(batch.read_off[batch.region_read_off])
        ctg_region = np.repeat(np.arange(self.n_regions), np.diff(self.contig_off))
        ev_ctg = np.repeat(np.arange(self.n_contigs), np.diff(self.event_off))
        tallied = self.events["status"] == A.IHP_EV_TALLIED
        tally = int(region_bases[ctg_region[ev_ctg[tallied]]].sum() + (2 * K + 12) * self.n_events)
        return {"k_assemble": asm, "k_ksw": ksw, "k_tally": tally}

    def first_difference(self, other):
        """None if bit-identical to `other`, else a description (integer fields only;
        GL/qual are floating point and compared with a tolerance by the caller)."""
        for f in self.FIELDS:
            a, b = getattr(self, f), getattr(other, f)
            if f == "events":
                names = [n for n in a.dtype.names if n not in ("gl", "qual")]
                if a.shape != b.shape:
                    return "events: count %d != %d" % (len(a), len(b))
                for n in names:
                    if not np.array_equal(a[n], b[n]):
                        i = int(np.flatnonzero(a[n] != b[n])[0])
                        return "events.%s differs at %d: %r != %r" % (n, i, a[n][i], b[n][i])
                continue
            if a.shape != b.shape:
                return "%s: shape %s != %s" % (f, a.shape, b.shape)
            if not np.array_equal(a, b):
                i = int(np.flatnonzero((a != b) if a.dtype.names is None else
                                       np.array([x != y for x, y in zip(a, b)]))[0])
                return "%s differs at %d: %r != %r" % (f, i, a[i], b[i])
        return None


class Api:
    def __init__(self, bound):
        self.b = bound

    # ---- params ------------------------------------------------------------
    def params(self, **kw):
        p = A.Params()
        self.b.params_default(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
        return p

    @staticmethod
    def _chk(rc, what=""):
        if rc != 0:
            raise IhpError(rc, what)

    # ---- contig.nim --------------------------------------------------------
    def slide_align(self, q, t, min_overlap=50, max_mismatch=0, allowed=A.IHP_ALLOW_DEFAULT,
                    qstart=0, tstart=0):
        """contig.nim:70 (and the string overload :152)."""
        if isinstance(q, (str, bytes)):
            q = Contig(q, qstart)
        if isinstance(t, (str, bytes)):
            t = Contig(t, tstart)
        cap = 16
        while True:
            corr = (A.Correction * cap)()
            m = A.Match(corrections=corr, corr_cap=cap)
            qc, tc = q._c(), t._c()
            rc = self.b.slide_align(C.byref(qc), C.byref(tc), min_overlap, max_mismatch, allowed, C.byref(m))
            if rc == A.IHP_E_CAPACITY:
                cap = int(m.n_corrections)
                continue
            self._chk(rc, "slide_align")
            break
        return Match(m.matches, m.offset, m.mismatches,
                     [(corr[i].qoff, corr[i].toff, bool(corr[i].qbest)) for i in range(m.n_corrections)],
                     m.contig_i)

    def insert(self, t, q, m):
        """insert(t, q, m), contig.nim:156."""
        n = len(m.corrections)
        corr = (A.Correction * max(1, n))()
        for i, (qo, to, qb) in enumerate(m.corrections):
            corr[i].qoff, corr[i].toff, corr[i].qbest = qo, to, int(qb)
        cm = A.Match(m.matches, m.offset, m.mismatches, n, m.contig_i, corr, max(1, n))
        need = abs(m.offset) + t.len + q.len if m.aligned else 0
        if need > t.cap:
            t._seq = np.concatenate([t._seq, np.zeros(need, np.uint8)])
            t._sup = np.concatenate([t._sup, np.zeros(need, np.uint32)])
            t.cap = len(t._seq)
        tc, qc = t._c(), q._c()
        self._chk(self.b.contig_insert(C.byref(tc), C.byref(qc), C.byref(cm)), "contig_insert")
        t._back(tc)
        q._back(qc)

    def trim(self, c, min_support=2):
        """trim(c, min_support), contig.nim:49."""
        cc = c._c()
        self._chk(self.b.contig_trim(C.byref(cc), min_support), "contig_trim")
        c._back(cc)

    # ---- ksw2.nim ----------------------------------------------------------
    def encode(self, dna):
        """encode, ksw2.nim:129."""
        if isinstance(dna, str):
            dna = dna.encode()
        a = np.frombuffer(dna, np.uint8).copy()
        out = np.empty(len(a), np.uint8)
        self.b.encode(A.ptr(a, A.u8p), len(a), A.ptr(out, A.u8p))
        return out

    def matrix(self, match=1, mismatch=-2):
        """matrix, ksw2.nim:135."""
        out = np.zeros(25, np.int8)
        self.b.matrix(match, mismatch, A.ptr(out, A.i8p))
        return out

    def align_batch(self, queries, targets, match=1, mismatch=-2, gap_open=4, gap_ext=1,
                    bw=-1, z=-1, flag=0, encoded=False):
        """n x align_to (ksw2.nim:151-164).  Returns (ez records, list of CIGAR word arrays)."""
        enc = (lambda s: np.asarray(s, np.uint8)) if encoded else self.encode
        qs, ts = [enc(s) for s in queries], [enc(s) for s in targets]
        n = len(qs)
        q_off = np.zeros(n + 1, np.int64)
        t_off = np.zeros(n + 1, np.int64)
        q_off[1:] = np.cumsum([len(x) for x in qs])
        t_off[1:] = np.cumsum([len(x) for x in ts])
        qcat = np.concatenate(qs) if n else np.zeros(0, np.uint8)
        tcat = np.concatenate(ts) if n else np.zeros(0, np.uint8)
        qcat = np.ascontiguousarray(qcat if len(qcat) else np.zeros(1, np.uint8))
        tcat = np.ascontiguousarray(tcat if len(tcat) else np.zeros(1, np.uint8))
        mat = self.matrix(match, mismatch)
        ez = np.zeros(n, A.EZ_DTYPE)
        cap = max(64, 16 * n)
        while True:
            cig = np.zeros(cap, np.uint32)
            coff = np.zeros(n + 1, np.int64)
            rc = self.b.ksw_extz2_batch(n, A.ptr(qcat, A.u8p), A.ptr(q_off, A.i64p), A.ptr(tcat, A.u8p),
                                        A.ptr(t_off, A.i64p), 5, A.ptr(mat, A.i8p), abs(gap_open), abs(gap_ext),
                                        bw, z, flag, ez.ctypes.data_as(C.POINTER(A.Ez)),
                                        A.ptr(cig, A.u32p), cap, A.ptr(coff, A.i64p))
            if rc == A.IHP_E_CAPACITY:
                cap = int(coff[n]) + 16
                continue
            self._chk(rc, "ksw_extz2_batch")
            break
        return ez, [cig[coff[i]:coff[i + 1]].copy() for i in range(n)]

    def duo_batch(self, reads, t0s, t1s, match=1, mismatch=-2, gap_open=5, gap_ext=1, bw=-1, z=-1, flag=0):
        """Diagnostics: (read, target 0, target 1) items through the alignment fallback's two-target sweep (ksw_duo.h; the two
        align_to calls of indelope.nim:340-341 in one pass).  Strings are base codes.  Returns (ez records [n, 2], CIGARs [n][2])."""
        n = len(reads)

        def cat(xs):
            off = np.zeros(n + 1, np.int64)
            off[1:] = np.cumsum([len(x) for x in xs])
            c = np.concatenate([np.asarray(x, np.uint8) for x in xs]) if n else np.zeros(0, np.uint8)
            return np.ascontiguousarray(c if len(c) else np.zeros(1, np.uint8)), off
        (q, qo), (a, ao), (b, bo) = cat(reads), cat(t0s), cat(t1s)
        slot = max([len(r) + max(len(x), len(y)) for r, x, y in zip(reads, t0s, t1s)] + [8]) + 16
        ez = np.zeros(2 * n, A.EZ_DTYPE)
        cig = np.zeros((2 * n, slot), np.uint32)
        mat = self.matrix(match, mismatch)
        self._chk(self.b.debug_ksw_duo_batch(n, A.ptr(q, A.u8p), A.ptr(qo, A.i64p), A.ptr(a, A.u8p), A.ptr(ao, A.i64p), A.ptr(b, A.u8p),
                                             A.ptr(bo, A.i64p), 5, A.ptr(mat, A.i8p), abs(gap_open), abs(gap_ext), bw, z, flag,
                                             ez.ctypes.data_as(C.c_void_p), A.ptr(cig, A.u32p), slot), "debug_ksw_duo_batch")
        cigs = [[cig[2 * i + k, :max(0, int(ez[2 * i + k]["n_cigar"]))].copy() for k in range(2)] for i in range(n)]
        return ez.reshape(n, 2), cigs

    # ---- tally / genotype ---------------------------------------------------
    def kmer_tally(self, reads, ref_kmer, alt_kmer, K=27, mapq=None, min_mapq=10):
        """The tally loop of indelope.nim:285-311 for one event."""
        rs = [np.frombuffer(r.encode() if isinstance(r, str) else r, np.uint8) for r in reads]
        off = np.zeros(len(rs) + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in rs])
        cat = np.ascontiguousarray(np.concatenate(rs) if rs else np.zeros(1, np.uint8))
        mq = np.full(len(rs), 60, np.uint8) if mapq is None else np.asarray(mapq, np.uint8)
        counts = np.zeros(3, np.int32)
        self._chk(self.b.kmer_tally(len(rs), A.ptr(cat, A.u8p), A.ptr(off, A.i64p), A.ptr(mq, A.u8p), min_mapq, K,
                                    ref_kmer.encode(), alt_kmer.encode(), A.ptr(counts, A.i32p)), "kmer_tally")
        return tuple(int(x) for x in counts)

    def genotype(self, r, a, error):
        """genotype(r, a, error), genotyper.nim:36."""
        g = A.Genotype()
        self._chk(self.b.genotype(r, a, error, C.byref(g)), "genotype")
        return g

    def qual(self, g):
        """qual(g), genotyper.nim:22."""
        return float(self.b.genotype_qual(C.byref(g)))

    # ---- the batched per-region path ----------------------------------------
    def run_regions(self, batch, params=None):
        p = params if params is not None else self.params()
        cin = batch.as_c()
        out = A.BatchOut()
        self._chk(self.b.run_regions(C.byref(p), C.byref(cin), C.byref(out)), "run_regions")
        try:
            return BatchResult(out)
        finally:
            self.b.free_out(C.byref(out))


GT_STR = ["0/0", "0/1", "1/1", "./."]             # genotyper.nim:17


VARIANT_FIELDS = ("region", "contig", "event", "filter", "gt", "start", "qual", "gq", "ake", "rke", "dp", "bs", "mf", "cf",
                  "nc", "amq", "rmq", "lo", "al", "event_type")


def _variants(self, batch, result, params=None, chrom="chr1"):
    """Api.call_variants: indelope.nim:375-428 + :604-608 over one batch -> list of dicts (one per tallied event), each
    with the fields of `ihp_variant`, `ref`/`alt`/`cc` strings and, for variants that pass every test of :375-428
    (printed, or dropped only as a duplicate of the last two), the VCF `line`.  `chrom`: a name, or region -> name."""
    chrom_of = chrom if callable(chrom) else (lambda r: chrom)
    p = params if params is not None else self.params()
    cin, cout = batch.as_c(), result.as_c()
    vs = A.Variants()
    self._chk(self.b.call_variants(C.byref(p), C.byref(cin), C.byref(cout), C.byref(vs)), "call_variants")
    try:
        chars = C.string_at(vs.chars, vs.n_chars) if vs.n_chars else b""
        res = []
        for i in range(vs.n):
            v = vs.v[i]
            d = {f: getattr(v, f) for f in VARIANT_FIELDS}
            d["gl"], d["ad"] = list(v.gl), list(v.ad)
            d["ref_kmer"], d["alt_kmer"] = v.ref_kmer.decode(), v.alt_kmer.decode()
            d["ref"] = chars[v.ref_off:v.ref_off + v.ref_len].decode()
            d["alt"] = chars[v.alt_off:v.alt_off + v.alt_len].decode()
            d["cc"] = chars[v.cc_off:v.cc_off + v.cc_len].decode()
            d["line"] = None
            if v.filter in (A.IHP_VF_EMITTED, A.IHP_VF_DUPLICATE):
                name = chrom_of(v.region).encode()
                need = self.b.format_variant(C.byref(v), vs.chars, name, None, 0)
                buf = C.create_string_buffer(need + 1)
                self.b.format_variant(C.byref(v), vs.chars, name, buf, need + 1)
                d["line"] = buf.value.decode()
            res.append(d)
        return res
    finally:
        self.b.free_variants(C.byref(vs))


Api.call_variants = _variants

CIGAR_OPS = "MIDNSHP=X"


def _gen_roi(self, read_start, read_stop, cigars, read_skip=None, origin=0, span=None, min_event_support=4,
             min_read_coverage=4, max_read_coverage=600):
    """gen_roi (indelope.nim:515-545) over reads in BAM order.  `cigars`: one uint32 array (BAM encoding
    len << 4 | op) per read.  Returns [(roi_start, roi_stop, [read indices])]."""
    n = len(read_start)
    st = np.ascontiguousarray(read_start, np.int64)
    en = np.ascontiguousarray(read_stop, np.int64)
    off = np.zeros(n + 1, np.int64)
    off[1:] = np.cumsum([len(c) for c in cigars])
    cig = np.ascontiguousarray(np.concatenate([np.asarray(c, np.uint32) for c in cigars]) if n and off[-1] else np.zeros(1, np.uint32))
    sk = None if read_skip is None else np.ascontiguousarray(read_skip, np.uint8)
    if span is None:
        span = int(en.max() - origin + 1) if n else 0
    rin = A.RoiIn(n, A.ptr(st if n else np.zeros(1, np.int64), A.i64p), A.ptr(en if n else np.zeros(1, np.int64), A.i64p),
                  A.ptr(sk, A.u8p), A.ptr(off, A.i64p), A.ptr(cig, A.u32p), origin, span,
                  min_event_support, min_read_coverage, max_read_coverage)
    out = A.RoiOut()
    self._chk(self.b.gen_roi(C.byref(rin), C.byref(out)), "gen_roi")
    try:
        rs, re = _np(out.roi_start, out.n_roi, np.int64), _np(out.roi_stop, out.n_roi, np.int64)
        ro, rd = _np(out.read_off, out.n_roi + 1, np.int64), _np(out.reads, out.n_read_idx, np.int64)
        return [(int(rs[k]), int(re[k]), rd[ro[k]:ro[k + 1]].tolist()) for k in range(out.n_roi)]
    finally:
        self.b.free_roi(C.byref(out))


Api.gen_roi = _gen_roi


def _pack_out(self, result):
    """(slab bytes as a uint8 array, counts[6]) of a BatchResult: the layout ihp_batch_pack_dev produces on the device."""
    o = result.as_c()
    nbytes, counts = C.c_int64(), np.zeros(6, np.int64)
    self._chk(self.b.pack_out(C.byref(o), None, 0, C.byref(nbytes), A.ptr(counts, A.i64p)), "pack_out")
    slab = np.zeros(max(1, nbytes.value), np.uint8)
    self._chk(self.b.pack_out(C.byref(o), slab.ctypes.data_as(C.c_void_p), len(slab), C.byref(nbytes), A.ptr(counts, A.i64p)),
              "pack_out")
    return slab[:nbytes.value], counts


def _unpack_slab(self, slab, counts, error=1e-3):
    """BatchResult from a packed slab (uint8 array; e.g. another rank's device slab copied to the host)."""
    slab = np.ascontiguousarray(slab, np.uint8)
    counts = np.ascontiguousarray(counts, np.int64)
    out = A.BatchOut()
    buf = slab if len(slab) else np.zeros(1, np.uint8)
    self._chk(self.b.unpack_slab(buf.ctypes.data_as(C.c_void_p), len(slab), A.ptr(counts, A.i64p), error, C.byref(out)),
              "unpack_slab")
    return BatchResult(out)             # numpy copies; `slab` stays the owner of the memory the views pointed into


Api.pack_out = _pack_out
Api.unpack_slab = _unpack_slab


def concat_batches(parts):
    """RegionBatches one behind the other as one batch (regions keep their order; trim bounds only if every part has them)."""
    cat = np.concatenate

    def offs(name, cnt):
        out, base = [np.zeros(1, np.int64)], 0
        for p in parts:
            a = np.asarray(getattr(p, name), np.int64)
            out.append(a[1:] + base)
            base += int(a[-1])
        return cat(out)

    def opt(name, fill, lens):
        cols = [getattr(p, name) for p in parts]
        if all(c is None for c in cols):
            return None
        return cat([np.full(n, fill, np.uint8) if c is None else c for c, n in zip(cols, lens)])
    n_reads = [p.n_reads for p in parts]
    n_bases = [len(p.bases) for p in parts]
    trims = all(p.trim_lo is not None for p in parts)
    return RegionBatch(offs("region_read_off", 0), offs("read_off", 0), cat([p.bases for p in parts]), opt("quals", 30, n_bases),
                       cat([p.read_start for p in parts]), cat([p.read_stop for p in parts]), cat([p.mapq for p in parts]),
                       opt("read_skip", 0, n_reads), offs("ref_off", 0), cat([p.ref_bases for p in parts]), cat([p.ref_origin for p in parts]),
                       cat([p.trim_lo for p in parts]) if trims else None, cat([p.trim_hi for p in parts]) if trims else None)


def concat_results(parts):
    """BatchResults of consecutive region ranges (the ranks' shards, in rank order) as one."""
    import copy
    if len(parts) == 1:
        return parts[0]
    r = copy.copy(parts[0])
    cat = np.concatenate
    for f in ("status", "n_contigs_pre", "ctg_start", "ctg_nreads", "ctg_seq", "ctg_support", "aln_flags", "aln_ref_start",
              "aln_ref_len", "aln_ez", "cigar", "events", "ref_hit", "alt_hit"):
        setattr(r, f, cat([getattr(p, f) for p in parts]))

    def offs(name, base_len):
        out, base = [], 0
        for i, p in enumerate(parts):
            a = getattr(p, name)
            out.append(a[:-1] + base)
            base += int(a[-1])
        return cat(out + [np.array([base], np.int64)])
    r.contig_off = offs("contig_off", None)
    r.ctg_seq_off = offs("ctg_seq_off", None)
    r.cigar_off = offs("cigar_off", None)
    r.event_off = offs("event_off", None)
    r.hit_off = offs("hit_off", None)
    r.n_regions = sum(p.n_regions for p in parts)
    r.n_contigs = sum(p.n_contigs for p in parts)
    r.n_events = sum(p.n_events for p in parts)
    r.n_hits = sum(p.n_hits for p in parts)
    return r
