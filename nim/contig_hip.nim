## contig_hip.nim -- a PATCH for src/contig.nim, not a module: `include` it right after contig.nim's type section
## (contig.nim:6-27); it adds three procs that hand the work of `trim` (contig.nim:49), `slide_align` (:70) and
## `insert(t, q, m)` (:156) to the HIP library through include/indelope_hip.h.  Everything else of contig.nim -- the
## types, `match_sort`, `make_contig`, `best_match`, the seq `insert`s, `combine` -- stays the reference's own code and
## reaches the device through these three (INTEGRATION.md section 3 shows the diff).
## SOURCE ONLY: there is no Nim toolchain in the build image, so this file has never been compiled.
##
## One call is one kernel launch and two copies: the compatibility path the reference's in-file tests exercise
## (contig.nim:292-430).  The BAM sweep uses the batched path of nim/indelope_hip.nim (Stager + flush).
import indelope_hip

proc hipView(c: Contig, extra: int = 0): IhpContig =
  ## the contig's own buffers, grown by `extra` elements when the call may make it longer
  let n = c.sequence.len
  if extra > 0:
    c.sequence.set_len(n + extra); c.support.set_len(n + extra)
  result.sequence = if c.sequence.len > 0: cast[ptr uint8](addr c.sequence[0]) else: nil
  result.support = if c.support.len > 0: addr c.support[0] else: nil
  result.len = int64(n); result.cap = int64(c.sequence.len)
  result.nreads = int64(c.nreads); result.start = int64(c.start)

proc hipTake(c: Contig, v: IhpContig) =
  c.sequence.set_len(int(v.len)); c.support.set_len(int(v.len))
  c.nreads = int(v.nreads); c.start = int(v.start)

proc hipCheck(rc: cint, what: string) =
  if rc != 0: raise newException(IOError, what & ": " & $ihp_strerror(rc) & " / " & $ihp_last_hip_error())

proc hip_trim(c: Contig, min_support: int) =
  var v = c.hipView()
  hipCheck(ihp_contig_trim(addr v, int64(min_support)), "ihp_contig_trim")
  c.hipTake(v)

proc hip_slide_align(q: Contig, t: Contig, min_overlap, max_mismatch: int, default_rule: bool): Match =
  ## `allowed` of the reference is a Nim closure and cannot cross to the device; the two rules it is ever given
  ## (contig.nim:44-47 and the tests' `allow_test`, :287-290) are IHP_ALLOW_DEFAULT / IHP_ALLOW_SUPPORT.
  var corr = new_seq[IhpCorrection](16)
  var m: IhpMatch
  while true:
    m.corrections = addr corr[0]; m.corr_cap = int64(corr.len)
    var qv = q.hipView()
    var tv = t.hipView()
    let rc = ihp_slide_align(addr qv, addr tv, int64(min_overlap), int64(max_mismatch),
                             (if default_rule: IHP_ALLOW_DEFAULT else: IHP_ALLOW_SUPPORT), addr m)
    if rc == IHP_E_CAPACITY:               # m.n_corrections holds the count needed
      corr.set_len(int(m.n_corrections)); continue
    hipCheck(rc, "ihp_slide_align")
    break
  result.matches = int(m.matches); result.mismatches = int(m.mismatches); result.contig_i = -1
  result.offset = if m.offset == IHP_UNALIGNED: unaligned else: int(m.offset)
  result.corrections = new_seq[correction_site](int(m.n_corrections))
  for i in 0..<int(m.n_corrections):
    result.corrections[i] = (int(corr[i].qoff), int(corr[i].toff), corr[i].qbest != 0)

proc hip_insert(t: Contig, q: Contig, m: Match) =
  var corr = new_seq[IhpCorrection](max(1, m.corrections.len))
  for i, c in m.corrections:
    corr[i].qoff = int64(c.qoff); corr[i].toff = int64(c.toff); corr[i].qbest = int32(c.qbest)
  var cm: IhpMatch
  cm.matches = int64(m.matches); cm.offset = int64(m.offset); cm.mismatches = int64(m.mismatches)
  cm.n_corrections = int64(m.corrections.len); cm.contig_i = int64(m.contig_i)
  cm.corrections = addr corr[0]; cm.corr_cap = int64(corr.len)
  var tv = t.hipView(extra = abs(m.offset) + q.sequence.len)   # the merged contig is at most |offset| + len(t) + len(q) long
  var qv = q.hipView()
  hipCheck(ihp_contig_insert(addr tv, addr qv, addr cm), "ihp_contig_insert")
  t.hipTake(tv); q.hipTake(qv)              # the library mutates q as the reference does (corrections)
