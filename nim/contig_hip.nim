## contig_hip.nim -- the `Contig` surface of src/contig.nim with the reference's proc signatures, delegating every
## step that touches bases or supports to the HIP library through include/indelope_hip.h (SOURCE ONLY: there is no
## Nim toolchain in the build image, so this file has never been compiled).
##
## These are the single-step entries the reference's own in-file tests exercise (contig.nim:292-430); one call is one
## kernel launch and two copies, so they are the compatibility path.  The BAM sweep uses the batched path of
## nim/indelope_hip.nim (Stager + flush), where assemble/combine run inside ihp_run_regions.
import algorithm
import indelope_hip

type
  Contig* = ref object of RootObj        # contig.nim:7-15
    sequence*: string
    support*: seq[uint32]
    nreads*: int
    start*: int

  correction_site* = tuple[qoff: int, toff: int, qbest: bool]                                        # contig.nim:17
  Match* = tuple[matches: int, offset: int, mismatches: int, corrections: seq[correction_site], contig_i: int]   # :21

const unaligned* = low(int)              # contig.nim:27

proc aligned*(ma: Match): bool {.inline.} = ma.offset != unaligned                                    # :29
proc len*(c: Contig): int {.inline.} = c.sequence.len                                                 # :38
proc `[]`*(c: Contig, i: int): char {.inline.} = c.sequence[i]                                        # :41

proc match_sort(a, b: Match): int =      # contig.nim:32-36
  if a.matches == b.matches: return a.mismatches - b.mismatches
  return b.matches - a.matches

proc make_contig*(dna: string, start: int, support: uint32 = 1): Contig =                             # :143-150
  var bc = new_seq[uint32](dna.len)
  for i in 0..bc.high: bc[i] = support
  return Contig(sequence: dna, support: bc, nreads: int(support), start: start)

# view of a Contig's buffers with `extra` elements of headroom for an insert that makes it longer
proc view(c: Contig, extra: int = 0): IhpContig =
  let n = c.sequence.len
  if extra > 0:
    c.sequence.set_len(n + extra); c.support.set_len(n + extra)
  result.sequence = if c.sequence.len > 0: cast[ptr uint8](addr c.sequence[0]) else: nil
  result.support = if c.support.len > 0: addr c.support[0] else: nil
  result.len = int64(n); result.cap = int64(c.sequence.len)
  result.nreads = int64(c.nreads); result.start = int64(c.start)

proc take(c: Contig, v: IhpContig) =
  c.sequence.set_len(int(v.len)); c.support.set_len(int(v.len))
  c.nreads = int(v.nreads); c.start = int(v.start)

proc check(rc: cint, what: string) =
  if rc != 0: raise newException(IOError, what & ": " & $ihp_strerror(rc) & " / " & $ihp_last_hip_error())

proc trim*(c: Contig, min_support: int = 2) =                                                         # contig.nim:49
  var v = c.view()
  check(ihp_contig_trim(addr v, int64(min_support)), "ihp_contig_trim")
  c.take(v)

proc slide_align*(q: Contig, t: var Contig, min_overlap: int = 50, max_mismatch: int = 0,
                  allow_rule: cint = IHP_ALLOW_DEFAULT): Match =                                       # contig.nim:70
  ## `allowed: allowable_mismatch_fn` of the reference is a closure and cannot cross to the device; the two rules the
  ## reference ever passes (contig.nim:44-47 and the test rule :287-290) are selected by `allow_rule`.
  var corr = new_seq[IhpCorrection](16)
  var m: IhpMatch
  while true:
    m.corrections = addr corr[0]; m.corr_cap = int64(corr.len)
    var qv = q.view()
    var tv = t.view()
    let rc = ihp_slide_align(addr qv, addr tv, int64(min_overlap), int64(max_mismatch), allow_rule, addr m)
    if rc == IHP_E_CAPACITY:              # n_corrections holds the count needed
      corr.set_len(int(m.n_corrections)); continue
    check(rc, "ihp_slide_align")
    break
  result = (matches: int(m.matches), offset: (if m.offset == IHP_UNALIGNED: unaligned else: int(m.offset)),
            mismatches: int(m.mismatches), corrections: new_seq[correction_site](int(m.n_corrections)), contig_i: -1)
  for i in 0..<int(m.n_corrections):
    result.corrections[i] = (int(corr[i].qoff), int(corr[i].toff), corr[i].qbest != 0)

proc insert*(t: var Contig, q: var Contig, m: var Match) =                                             # contig.nim:156
  if not m.aligned: return
  var corr = new_seq[IhpCorrection](max(1, m.corrections.len))
  for i, c in m.corrections:
    corr[i].qoff = int64(c.qoff); corr[i].toff = int64(c.toff); corr[i].qbest = int32(c.qbest)
  var cm: IhpMatch
  cm.matches = int64(m.matches); cm.offset = int64(m.offset); cm.mismatches = int64(m.mismatches)
  cm.n_corrections = int64(m.corrections.len); cm.contig_i = int64(m.contig_i)
  cm.corrections = addr corr[0]; cm.corr_cap = int64(corr.len)
  var tv = t.view(extra = abs(m.offset) + q.len)      # the merged contig is at most |offset| + len(t) + len(q) long
  var qv = q.view()
  check(ihp_contig_insert(addr tv, addr qv, addr cm), "ihp_contig_insert")
  t.take(tv); q.take(qv)                              # insert mutates q too (corrections, :167-169)

proc best_match(contigs: var seq[Contig], q: Contig, min_overlap: int = 65, max_mismatch: int = 0): Match =   # :224-240
  var matches = new_seq_of_cap[Match](2)
  for i, c in contigs:
    if c == q: continue
    var ma = slide_align(q, contigs[i], min_overlap = min_overlap, max_mismatch = max_mismatch)
    if ma.aligned:
      ma.contig_i = i
      matches.add(ma)
  if len(matches) == 0:
    var ma: Match
    ma.offset = unaligned
    return ma
  matches.sort(match_sort)               # stable merge sort: ties go to the lower contig index
  return matches[0]

proc insert*(contigs: var seq[Contig], q: var Contig, min_overlap: int = 50, max_mismatch: int = 0) =  # contig.nim:243
  var ma = contigs.best_match(q, min_overlap = min_overlap, max_mismatch = max_mismatch)
  if ma.aligned: contigs[ma.contig_i].insert(q, ma)
  else: contigs.add(q)

proc insert*(contigs: var seq[Contig], q: string, start: int, min_overlap: int = 50, max_mismatch: int = 0) =   # :250
  var qc = make_contig(q, start)
  contigs.insert(qc, min_overlap = min_overlap, max_mismatch = max_mismatch)

proc combine*(contigs: var seq[Contig], max_mismatch: int = 0, min_support: int = 3, again: bool = true): seq[Contig] =   # :254
  if again:
    contigs = contigs.combine(max_mismatch, min_support = 0, again = false)
  result = new_seq_of_cap[Contig](len(contigs))
  var usedi = 0
  for i, c in contigs:
    if min_support > 0:
      c.trim(min_support = min(c.nreads, min_support))
    if c.nreads > 0 and result.len == 0:
      result.add(c)
      usedi = i
  if result.len == 0: return
  for i in 0..contigs.high:
    if i == usedi: continue
    var ma = result.best_match(contigs[i], max_mismatch = max_mismatch)
    if ma.aligned:
      result[ma.contig_i].insert(contigs[i], ma)
    elif contigs[i].nreads > 0:
      result.add(contigs[i])
