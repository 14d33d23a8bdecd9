"""The batching adapter of SURVEY.md §8f row f3, in the host language available here.

indelope's main loop (src/indelope.nim:601-608) calls `callsemble` once per `roi` and prints what it yields,
skipping a variant that equals one of the last two printed.  Behind the GPU library the sweep instead accumulates
`roi`s, flushes them to `ihp_run_regions` every N regions and walks the results in region order.  This module is
that adapter over the C ABI (`nim/indelope_hip.nim` is the same logic written for the reference's own host
language, which cannot be compiled in this image):

    caller = BatchedCaller(api, api.params(min_reads=3, min_ctg_len=73), batch_regions=4096)
    for roi in rois:                       # Roi: what `gen_roi` yields, decoded (indelope.nim:21)
        for line in caller.add(roi):       # VCF lines, in region order, as batches complete
            print(line)
    for line in caller.flush():
        print(line)

The output does not depend on `batch_regions`: regions are independent, results come back in region order, and the
last-two-printed window (:598-608) is carried across flushes here rather than inside one `ihp_call_variants` call.
"""
from dataclasses import dataclass, field

import numpy as np

from . import _abi as A
from .host import RegionBatch


@dataclass
class Read:
    """The fields of an hts-nim `Record` the path reads (indelope.nim:163-169, :213-216, :293-300)."""
    sequence: bytes
    qualities: np.ndarray            # phred, len(sequence); None = all good
    start: int
    stop: int
    mapq: int
    skippable: bool = False          # indelope.nim:40-47


@dataclass
class Roi:
    """`roi = tuple[start, stop, reads]` (indelope.nim:21) plus the reference slice the region's `fai.get`s fall in."""
    start: int
    stop: int
    reads: list
    ref_bases: bytes = b""
    ref_origin: int = 0
    chrom: str = "chr1"


def trim_bounds(qualities, n, min_quality=15):
    """trim(sequence, base_qualities, min_quality) of indelope.nim:23-38 -> (lo, hi): kept [lo, hi), start += lo."""
    if n == 1:
        return 0, 0                            # a == high == 0: the read is emptied whatever its quality (:28-30)
    if qualities is None or n == 0:
        return 0, n
    high = n - 1
    good = np.flatnonzero(np.asarray(qualities[:n]) >= min_quality)
    head = good[good < high]
    a = int(head[0]) if len(head) else high
    if a == high:
        return a, a
    tail = good[good > a]
    b = int(tail[-1]) if len(tail) else a
    return a, b + 1


@dataclass
class BatchedCaller:
    api: object
    params: object = None
    batch_regions: int = 4096
    trim_on_host: bool = True        # hand over trim bounds instead of qualities (half the bytes)
    _rois: list = field(default_factory=list)
    _printed: list = field(default_factory=list)     # last_var, last_var2 (:598-599): (chrom, start, ref, alt)

    def __post_init__(self):
        if self.params is None:
            self.params = self.api.params()

    # ---- staging (nim/indelope_hip.nim `Stager.add`) ---------------------------------------
    def _stage(self):
        rro, ro, rfo = [0], [0], [0]
        bases, quals, starts, stops, mapq, skip, lo, hi, refs, origins = [], [], [], [], [], [], [], [], [], []
        for roi in self._rois:
            for rd in roi.reads:
                n = len(rd.sequence)
                bases.append(np.frombuffer(rd.sequence, np.uint8))
                q = np.full(n, 255, np.uint8) if rd.qualities is None else np.asarray(rd.qualities, np.uint8)[:n]
                quals.append(q)
                a, b = trim_bounds(rd.qualities, n, self.params.trim_min_qual)
                lo.append(a)
                hi.append(b)
                ro.append(ro[-1] + n)
                starts.append(rd.start)
                stops.append(rd.stop)
                mapq.append(rd.mapq)
                skip.append(1 if rd.skippable else 0)
            rro.append(len(starts))
            refs.append(np.frombuffer(roi.ref_bases, np.uint8))
            rfo.append(rfo[-1] + len(roi.ref_bases))
            origins.append(roi.ref_origin)
        cat = lambda xs, dt: np.ascontiguousarray(np.concatenate(xs) if xs else np.zeros(0, dt), dt)  # noqa: E731
        return RegionBatch(np.array(rro, np.int64), np.array(ro, np.int64), cat(bases, np.uint8),
                           None if self.trim_on_host else cat(quals, np.uint8),
                           np.array(starts, np.int64), np.array(stops, np.int64), np.array(mapq, np.uint8),
                           np.array(skip, np.uint8), np.array(rfo, np.int64), cat(refs, np.uint8),
                           np.array(origins, np.int64),
                           np.array(lo, np.int32) if self.trim_on_host else None,
                           np.array(hi, np.int32) if self.trim_on_host else None)

    # ---- the sweep ------------------------------------------------------------------------------
    def add(self, roi):
        self._rois.append(roi)
        if len(self._rois) >= self.batch_regions:
            return self.flush()
        return []

    def flush(self):
        if not self._rois:
            return []
        batch = self._stage()
        res = self.api.run_regions(batch, self.params)
        lines = []
        rois = self._rois
        for v in self.api.call_variants(batch, res, self.params, chrom=lambda r: rois[r].chrom):
            # every test of indelope.nim:375-428 is the library's; the last-two window (:604-608) is kept here so that
            # it runs across flushes: a record the library marked DUPLICATE is a candidate like an EMITTED one
            if v["filter"] not in (A.IHP_VF_EMITTED, A.IHP_VF_DUPLICATE):
                continue
            key = (rois[v["region"]].chrom, v["start"], v["ref"], v["alt"])
            if key in self._printed:
                continue
            lines.append(v["line"])
            self._printed = [key] + self._printed[:1]
        self._rois = []
        return lines


def rois_from_batch(batch, chrom="chr1"):
    """Synthetic regions (a RegionBatch) as the `Roi`s a BAM sweep would hand over (tests, examples)."""
    out = []
    for r in range(batch.n_regions):
        r0, r1 = int(batch.region_read_off[r]), int(batch.region_read_off[r + 1])
        reads = []
        for i in range(r0, r1):
            lo, hi = int(batch.read_off[i]), int(batch.read_off[i + 1])
            reads.append(Read(batch.bases[lo:hi].tobytes(), None if batch.quals is None else batch.quals[lo:hi],
                              int(batch.read_start[i]), int(batch.read_stop[i]), int(batch.mapq[i]),
                              bool(batch.read_skip[i]) if batch.read_skip is not None else False))
        f0, f1 = int(batch.ref_off[r]), int(batch.ref_off[r + 1])
        s = min((rd.start for rd in reads), default=int(batch.ref_origin[r]))
        e = max((rd.stop for rd in reads), default=s)
        out.append(Roi(s, e, reads, batch.ref_bases[f0:f1].tobytes(), int(batch.ref_origin[r]), chrom))
    return out


def call_target(api, reads, cigars, fetch, params=None, chrom="chr1", batch_regions=4096, target_len=None,
                min_event_support=None, max_read_coverage=600):
    """The whole post-decode sweep of one target (indelope.nim:601-608): gen_roi over the decoded reads (`reads`: list
    of `Read` in BAM order, `cigars`: their BAM CIGAR words), every region staged with the reference slice its reads
    span (`fetch(start, stop)` -> bytes for [start, stop), clamped by the caller like faidx), batched through the
    path, VCF lines out in order."""
    p = params if params is not None else api.params()
    if min_event_support is None:
        min_event_support = max(3, p.min_reads - 2)            # indelope.nim:602
    st = np.array([r.start for r in reads], np.int64)
    en = np.array([r.stop for r in reads], np.int64)
    sk = np.array([1 if r.skippable else 0 for r in reads], np.uint8)
    span = (target_len if target_len is not None else int(en.max()) + 1) if len(reads) else 0
    rois = api.gen_roi(st, en, cigars, read_skip=sk, origin=0, span=span, min_event_support=min_event_support,
                       min_read_coverage=p.min_reads, max_read_coverage=max_read_coverage)
    caller = BatchedCaller(api, p, batch_regions=batch_regions)
    width = int((p.K + 1) / 2 - 1)                             # indelope.nim:218
    lines = []
    for rs, re, idx in rois:
        rr = [reads[i] for i in idx]
        lo = max(0, min(r.start for r in rr) - 1)              # REF alleles reach one base left of the event (:414, :421)
        hi = max(r.stop for r in rr) + width + p.ref_pad + 1   # fai.get(chrom, ctg.start, max_stop + width + 50), :220
        lines += caller.add(Roi(rs, re, rr, fetch(lo, hi), lo, chrom))
    return lines + caller.flush(), rois
