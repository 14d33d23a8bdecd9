## indelope_hip.nim -- Nim binding of include/indelope_hip.h (SOURCE ONLY: no Nim toolchain in the build image,
## so this file has never been compiled; it documents the binding a maintainer adds to brentp/indelope).
##
## 1. ksw2 seam: in src/ksw2/ksw2_c.nim replace
##        {.compile: "csrc/ksw2_extz2_sse.c".}
##    by
##        {.passL: "-lindelope_hip".}
##    `ksw_extz2_sse` (ksw2_c.nim:53-55) then resolves to the HIP library: same cdecl signature, same ksw_extz_t.
## 2. batched path: the sweep in indelope.nim:601-603 accumulates `roi`s, calls `run_regions` once per N regions
##    and feeds the per-event counts to genotype()/the filters of indelope.nim:375-428 unchanged.

{.passL: "-lindelope_hip".}

type
  IhpParams* {.importc: "ihp_params", header: "indelope_hip.h", bycopy.} = object
    struct_size*: int32
    min_overlap_pct*: float64
    min_mapq_assemble*, min_mapq_stop*, min_mapq_tally*, trim_min_qual*: int32
    combine_min_support*, combine_min_overlap*, max_mismatch*, max_pre_contigs*: int32
    min_ctg_len*, min_reads*, min_event_len*, K*, max_events*, ref_pad*: int32
    match*, mismatch*, gap_open*, gap_ext*: int8
    bw*, zdrop*, ksw_flag*: int32
    error*: float64
    fallback*: int32                      # run the alignment fallback of indelope.nim:312-372 on the GPU (default 1)
    fb_match*, fb_mismatch*, fb_gap_open*, fb_gap_ext*: int8   # new_ez(mismatch = -2, gap_open = 5, gap_ext = 1), :318-319
    fb_bw*, fb_zdrop*, fb_flag*: int32    # align_to defaults (-1, -1, 0), ksw2.nim:159

  IhpBatchIn* {.importc: "ihp_batch_in", header: "indelope_hip.h", bycopy.} = object
    n_regions*: int32
    n_reads*: int64
    region_read_off*, read_off*: ptr int64
    bases*, quals*: ptr uint8
    read_start*, read_stop*: ptr int64
    mapq*, read_skip*: ptr uint8
    ref_off*: ptr int64
    ref_bases*: ptr uint8
    ref_origin*: ptr int64
    trim_lo*, trim_hi*: ptr int32         # optional: trim() of indelope.nim:23-38 done by the stager; quals then unused

  IhpEz* {.importc: "ihp_ez", header: "indelope_hip.h", bycopy.} = object
    max*, zdropped*, max_q*, max_t*, mqe*, mqe_t*, mte*, mte_q*, score*, n_cigar*: int32

  IhpEvent* {.importc: "ihp_event", header: "indelope_hip.h", bycopy.} = object
    tstart*, tstop*, qstart*, qstop*: int64
    len*: uint32
    `type`*, status*, fallback_needed*, aligned*: uint8   # aligned = `aligned` of indelope.nim:372
    cf_offset*, ref_support*, alt_support*, both_found*: int32   # as they stand at indelope.nim:375
    ref_kmer*, alt_kmer*: array[32, char]
    gt*: int32
    kmer_ref_support*, kmer_alt_support*, kmer_both_found*: int32   # the k-mer tally itself (:285-311)
    gl*: array[3, float64]
    qual*: float64

  IhpBatchOut* {.importc: "ihp_batch_out", header: "indelope_hip.h", bycopy.} = object
    n_regions*: int32
    n_contigs*, n_events*, n_cigar_words*, n_bases*, n_hits*: int64
    status*, n_contigs_pre*: ptr int32
    contig_off*, ctg_start*, ctg_nreads*, ctg_seq_off*: ptr int64
    ctg_seq*: ptr uint8
    ctg_support*: ptr uint32
    aln_flags*: ptr int32
    aln_ref_start*: ptr int64
    aln_ref_len*: ptr int32
    aln_ez*: ptr IhpEz
    cigar_off*: ptr int64
    cigar*: ptr uint32
    event_off*: ptr int64
    events*: ptr IhpEvent
    hit_off*: ptr int64                   # [E+1]; per tallied event one entry per read of its region:
    ref_hit*, alt_hit*: ptr int32         # start of the first ref / alt k-mer window in the read, -1 = none (indelope.nim:301-309)

  IhpVariant* {.importc: "ihp_variant", header: "indelope_hip.h", bycopy.} = object   # one per tallied event (indelope.nim:375-428)
    region*, contig*: int32
    event*: int64
    filter*, gt*: int32                   # IHP_VF_* (0 = the reference prints it), IHP_GT_*
    start*: int64
    qual*, gq*: float64
    gl*: array[3, float64]
    ake*, rke*: float64
    ad*: array[2, int32]
    dp*, bs*, mf*, cf*, nc*, amq*, rmq*: int32
    lo*, al*, event_type*, pad: uint8
    ref_len*, alt_len*, cc_len*: int32
    ref_off*, alt_off*, cc_off*: int64
    ref_kmer*, alt_kmer*: array[32, char]

  IhpVariants* {.importc: "ihp_variants", header: "indelope_hip.h", bycopy.} = object
    n*: int64
    v*: ptr IhpVariant
    n_chars*: int64
    chars*: cstring

type
  IhpRoiIn* {.importc: "ihp_roi_in", header: "indelope_hip.h", bycopy.} = object   # gen_roi (indelope.nim:515-545) over decoded reads
    n_reads*: int64
    read_start*, read_stop*: ptr int64
    read_skip*: ptr uint8
    cigar_off*: ptr int64
    cigar*: ptr uint32                    # BAM encoding, as in bam1_t
    origin*, span*: int64
    min_event_support*, min_read_coverage*, max_read_coverage*: int32
  IhpRoiOut* {.importc: "ihp_roi_out", header: "indelope_hip.h", bycopy.} = object
    n_roi*, n_read_idx*: int64
    roi_start*, roi_stop*, read_off*, reads*: ptr int64

proc ihp_gen_roi*(inp: ptr IhpRoiIn, outp: ptr IhpRoiOut): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_free_roi*(outp: ptr IhpRoiOut) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_call_variants*(p: ptr IhpParams, inp: ptr IhpBatchIn, outp: ptr IhpBatchOut, vars: ptr IhpVariants): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_free_variants*(vars: ptr IhpVariants) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_format_variant*(v: ptr IhpVariant, chars: cstring, chrom: cstring, buf: cstring, cap: int64): int64 {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_host_alloc*(bytes: csize_t): pointer {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_host_free*(p: pointer) {.importc, cdecl, header: "indelope_hip.h".}

proc ihp_init*(device: cint): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_strerror*(code: cint): cstring {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_params_default*(p: ptr IhpParams) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_run_regions*(p: ptr IhpParams, inp: ptr IhpBatchIn, outp: ptr IhpBatchOut): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_free_out*(outp: ptr IhpBatchOut) {.importc, cdecl, header: "indelope_hip.h".}

## Stager for `roi = tuple[start, stop: int, reads: seq[Record]]` (indelope.nim:21): what assemble() reads from
## each Record (indelope.nim:163-169, :213-216, :293-300) flattened into the batch arrays.
type Stager* = object
  region_read_off*, read_off*, read_start*, read_stop*, ref_off*, ref_origin*: seq[int64]
  bases*, quals*, mapq*, read_skip*, ref_bases*: seq[uint8]
  trim_lo*, trim_hi*: seq[int32]        # what trim(read_seq, base_q) keeps: [lo, hi); start moves by lo

proc init*(s: var Stager) =
  s.region_read_off = @[0'i64]; s.read_off = @[0'i64]; s.ref_off = @[0'i64]

# proc add*(s: var Stager, r: roi, fai: Fai) =
#   var lo = high(int); var hi = 0
#   var read_seq = ""; var base_q = new_seq[uint8](300)
#   for read in r.reads:
#     discard read.sequence(read_seq); discard read.base_qualities(base_q)
#     for c in read_seq: s.bases.add(uint8(c))
#     # either hand over the qualities ...   for q in base_q[0..<read_seq.len]: s.quals.add(q)
#     # ... or trim here (half the bytes to upload): a = first i < high with q >= 15 else high; emptied read: lo = hi = a;
#     # else b = last i > a with q >= 15 else a; lo = a, hi = b + 1      (indelope.nim:23-38)
#     let (lo, hi) = trim_bounds(base_q, read_seq.len); s.trim_lo.add(lo); s.trim_hi.add(hi)
#     s.read_off.add(int64(s.bases.len))
#     s.read_start.add(read.start); s.read_stop.add(read.stop)
#     s.mapq.add(read.qual); s.read_skip.add(uint8(read.skippable(allow_unmapped=false)))
#     lo = min(lo, read.start); hi = max(hi, read.stop)
#   s.region_read_off.add(int64(s.read_start.len))
#   # one window per region covers every per-contig fai.get of indelope.nim:220 (width = 13 for K = 27)
#   let w = fai.get(r.reads[0].chrom, lo, hi + 13 + 50)
#   for c in w: s.ref_bases.add(uint8(c))
#   s.ref_off.add(int64(s.ref_bases.len)); s.ref_origin.add(int64(lo))

proc run*(s: var Stager, p: var IhpParams, outp: var IhpBatchOut): cint =
  var b: IhpBatchIn
  b.n_regions = int32(s.ref_origin.len); b.n_reads = int64(s.read_start.len)
  b.region_read_off = addr s.region_read_off[0]; b.read_off = addr s.read_off[0]
  b.bases = addr s.bases[0]; b.quals = addr s.quals[0]
  b.read_start = addr s.read_start[0]; b.read_stop = addr s.read_stop[0]
  b.mapq = addr s.mapq[0]; b.read_skip = addr s.read_skip[0]
  b.ref_off = addr s.ref_off[0]; b.ref_bases = addr s.ref_bases[0]; b.ref_origin = addr s.ref_origin[0]
  if s.trim_lo.len > 0: (b.trim_lo = addr s.trim_lo[0]; b.trim_hi = addr s.trim_hi[0]; b.quals = nil)
  result = ihp_run_regions(addr p, addr b, addr outp)
