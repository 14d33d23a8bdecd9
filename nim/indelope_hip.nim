## indelope_hip.nim -- Nim binding of include/indelope_hip.h (SOURCE ONLY: no Nim toolchain in the build image,
## so this file has never been compiled; it is the binding a maintainer adds to brentp/indelope).
## Companion files: nim/contig_hip.nim (a patch for contig.nim: `trim`, `slide_align`, `insert(t, q, m)` over the ABI) and
## nim/genotyper_hip.nim (a patch for genotyper.nim: `genotype`).  `roi`, `Fai`, `Record`, `skippable` are the
## reference's own (indelope.nim:21,:40; hts-nim): this file is meant to be `include`d into / imported by indelope.nim.
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
    # IHP_FETCH_COMPACT: ctg_seq / ctg_support nil; bases 4 bits each (contig c from byte (ctg_seq_off[c] shr 1) + c, first base in
    # the high nibble), supports a byte each, 255 = among the escapes; ihp_out_contig expands one contig
    ctg_seq4*, ctg_sup8*: ptr uint8
    n_sup_escapes*: int64
    sup_escape_idx*: ptr int64
    sup_escape_val*: ptr uint32

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
proc ihp_copy_to_host*(dev_ptr: pointer, bytes: int64, outp: pointer): cint {.importc, cdecl, header: "indelope_hip.h".}

# ---- Contig / genotyper / ksw2 single-step entries (called from the reference's own procs through the patches
# nim/contig_hip.nim and nim/genotyper_hip.nim)
type
  IhpContig* {.importc: "ihp_contig", header: "indelope_hip.h", bycopy.} = object   # contig.nim:7-15 over caller-owned buffers
    sequence*: ptr uint8
    support*: ptr uint32
    len*, cap*, nreads*, start*: int64
  IhpCorrection* {.importc: "ihp_correction", header: "indelope_hip.h", bycopy.} = object   # contig.nim:17
    qoff*, toff*: int64
    qbest*, pad: int32
  IhpMatch* {.importc: "ihp_match", header: "indelope_hip.h", bycopy.} = object          # contig.nim:21
    matches*, offset*, mismatches*, n_corrections*, contig_i*: int64
    corrections*: ptr IhpCorrection
    corr_cap*: int64
  IhpGenotype* {.importc: "ihp_genotype_t", header: "indelope_hip.h", bycopy.} = object   # genotyper.nim:14
    gt*, pad: int32
    gl*: array[3, float64]
  IhpBatch* {.importc: "ihp_batch", header: "indelope_hip.h", incompleteStruct.} = object   # opaque device-resident batch
  IhpRegionSummary* {.importc: "ihp_region_summary", header: "indelope_hip.h", bycopy.} = object
    status*, n_contigs_pre*, n_contigs*, n_aligned*, n_events*, n_tallied*, ref_support*, alt_support*: int32

const
  IHP_UNALIGNED* = low(int64)             # contig.nim:27
  IHP_ALLOW_DEFAULT* = 0.cint             # contig.nim:44-47
  IHP_ALLOW_SUPPORT* = 1.cint             # contig.nim:287-290 (the reference's test rule)
  IHP_E_ARG* = -3.cint
  IHP_E_CAPACITY* = -5.cint
  IHP_ALN_DONE* = 1'i32
  IHP_EV_TALLIED* = 0'u8
  IHP_VF_EMITTED* = 0'i32

proc ihp_slide_align*(q, t: ptr IhpContig, min_overlap, max_mismatch: int64, allow_rule: cint, m: ptr IhpMatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_contig_insert*(t, q: ptr IhpContig, m: ptr IhpMatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_contig_trim*(c: ptr IhpContig, min_support: int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_kmer_tally*(n_reads: int32, bases: ptr uint8, read_off: ptr int64, mapq: ptr uint8, min_mapq, K: int32,
                     ref_kmer, alt_kmer: cstring, counts: ptr int32): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_genotype*(r, a: int64, error: float64, g: ptr IhpGenotype): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_genotype_qual*(g: ptr IhpGenotype): float64 {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_ksw_extz2_batch*(n: int32, queries: ptr uint8, q_off: ptr int64, targets: ptr uint8, t_off: ptr int64, m: int8,
                          mat: ptr int8, q, e: int8, w, zdrop, flag: cint, ez: ptr IhpEz, cigar: ptr uint32, cigar_cap: int64,
                          cigar_off: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_ksw_last_status*(): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_encode*(dna: ptr uint8, n: int64, outp: ptr uint8) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_matrix*(match, mismatch: int8, out25: ptr int8) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_last_hip_error*(): cstring {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_shutdown*() {.importc, cdecl, header: "indelope_hip.h".}
# a batch kept resident in HBM: upload once, run (asynchronous, its own stream), sync, fetch
proc ihp_batch_upload*(p: ptr IhpParams, inp: ptr IhpBatchIn, b: ptr ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_run*(b: ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_sync*(b: ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_fetch*(b: ptr IhpBatch, outp: ptr IhpBatchOut): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_out_contig*(outp: ptr IhpBatchOut, c: int64, seq: ptr uint8, sup: ptr uint32): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_release_outputs*(b: ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_free*(b: ptr IhpBatch) {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_pack_dev*(b: ptr IhpBatch, dev_ptr: ptr pointer, bytes: ptr int64, counts: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_unpack_slab*(slab: pointer, bytes: int64, counts: ptr int64, error: float64, outp: ptr IhpBatchOut): cint {.importc, cdecl, header: "indelope_hip.h".}
# BLOCKING: waits for the batch's run and confirms it (the run may be repeated in full at this point); the records are final on return
proc ihp_batch_summary_dev*(b: ptr IhpBatch, dev_ptr: ptr pointer, n: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
# the same address without the wait (fixed from upload to free): for callers that order by stream / call ihp_batch_sync themselves
proc ihp_batch_summary_ptr*(b: ptr IhpBatch, dev_ptr: ptr pointer, n: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_summary_host*(b: ptr IhpBatch, outp: ptr IhpRegionSummary, cap: int64): cint {.importc, cdecl, header: "indelope_hip.h".}

# ---- multi-GPU: one process per GPU, ONE gather at the end (dist_host.h: librccl behind the C ABI; include/indelope_hip.h) ----------
type IhpDist* {.importc: "ihp_dist", header: "indelope_hip.h", incompleteStruct.} = object
const IHP_DIST_ID_BYTES* = 128
proc ihp_dist_unique_id*(id: pointer, cap: int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_init*(rank, world: int32, id: pointer, id_bytes: int64, outp: ptr ptr IhpDist): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_rank*(d: ptr IhpDist): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_world*(d: ptr IhpDist): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_gather_records*(d: ptr IhpDist, dev_records: pointer, n: int64, root: int32, counts_in: ptr int64,
                              outp: ptr IhpRegionSummary, cap: int64, n_total: ptr int64, counts_out: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_gather_summaries*(d: ptr IhpDist, b: ptr IhpBatch, root: int32, counts_in: ptr int64,
                                outp: ptr IhpRegionSummary, cap: int64, n_total: ptr int64, counts_out: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_gather_payload*(d: ptr IhpDist, b: ptr IhpBatch, root: int32, outs: ptr IhpBatchOut, bytes_out: ptr int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_dist_finalize*(d: ptr IhpDist): cint {.importc, cdecl, header: "indelope_hip.h".}
# one page-locked slab per batch (4-bit bases as BAM stores them, trim bounds): the sections' offsets, the upload, what a fetch returns
type IhpSlabLayout* {.importc: "ihp_slab_layout", header: "indelope_hip.h", bycopy.} = object
  region_read_off*, read_off*, read_start*, read_stop*, ref_off*, ref_origin*: int64
  trim_lo*, trim_hi*, mapq*, read_skip*, ref_bases*, bases4*, bytes*: int64
proc ihp_slab_layout_for*(n_regions: int32, n_reads, n_bases, n_ref: int64, outp: ptr IhpSlabLayout): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_upload_slab*(p: ptr IhpParams, n_regions: int32, n_reads: int64, slab: pointer, layout: ptr IhpSlabLayout,
                            flags: int32, b: ptr ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
# round 5: the compact slab (14 bytes per read, windows 2 or 4 bits per base; the arrays of ihp_batch_in are made on the device)
type IhpSlab2Layout* {.importc: "ihp_slab2_layout", header: "indelope_hip.h", bycopy.} = object
  region_read_off*, region_base_off*, ref_off*, ref_origin*, start_rel*, len*, span*, trim_lo*, trim_hi*: int64
  mapq*, rflags*, ref_packed*, bases4*, bytes*: int64
const IHP_SLAB2_REF_2BIT* = 2'i32
const IHP_SLAB2_BASES_2BIT* = 4'i32   ## the read bases 2 bits each in the library's packed form (every base of the batch is A C G T)
proc ihp_slab2_layout_for*(n_regions: int32, n_reads, n_bases, n_ref: int64, flags: int32, outp: ptr IhpSlab2Layout): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_upload_slab2*(p: ptr IhpParams, n_regions: int32, n_reads: int64, slab: pointer, layout: ptr IhpSlab2Layout,
                             flags: int32, b: ptr ptr IhpBatch): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_host_alloc*(bytes: csize_t): pointer {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_host_free*(p: pointer) {.importc, cdecl, header: "indelope_hip.h".}
const
  IHP_FETCH_NO_BASES* = 1'i32   ## ihp_batch_fetch brings everything but the contigs' bases and supports
  IHP_FETCH_EAGER* = 2'i32      ## every run also counts what its results will take: the fetch is one enqueue and one wait
  IHP_FETCH_COMPACT* = 4'i32    ## full results with the contigs' bases 4 bits each and their supports a byte each (a third of the copy)
proc ihp_batch_set_fetch*(b: ptr IhpBatch, flags: int32): cint {.importc, cdecl, header: "indelope_hip.h".}
# timing and diagnostics (what bench.py and the tests read; a caller needs none of them)
proc ihp_batch_set_timing*(b: ptr IhpBatch, on: cint): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_kernel_ms*(b: ptr IhpBatch, ms: ptr cfloat): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_kernel_ms_mean*(b: ptr IhpBatch, ms: ptr cfloat, n_runs: ptr int64, reset: cint): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_batch_profile_n*(b: ptr IhpBatch, outp: ptr int64, cap: int32): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_debug_set*(key: cstring, value: int64): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_debug_last_ksw_mode*(): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_debug_last_ksw_pairs*(): cint {.importc, cdecl, header: "indelope_hip.h".}
proc ihp_debug_ksw_duo_batch*(n: int32, reads: ptr uint8, q_off: ptr int64, t0: ptr uint8, t0_off: ptr int64, t1: ptr uint8, t1_off: ptr int64,
                              m: int8, mat: ptr int8, q: int8, e: int8, w: cint, zdrop: cint, flag: cint, ez: ptr IhpEz, cigar: ptr uint32,
                              cig_slot: int32): cint {.importc, cdecl, header: "indelope_hip.h".}

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

proc trim_bounds*(base_q: seq[uint8], n: int, min_quality: int = 15): tuple[lo: int32, hi: int32] =
  ## trim(sequence, base_qualities, min_quality) of indelope.nim:23-38 as bounds: the read keeps [lo, hi) and its start
  ## moves by lo; an emptied read (a == high, :28-30 -- every 1-base read among them) is lo == hi == a.
  let high = n - 1
  var a = 0
  while a < high and base_q[a] < uint8(min_quality): a += 1
  if a == high or n <= 0: return (int32(max(a, 0)), int32(max(a, 0)))
  var b = high
  while b > a and base_q[b] < uint8(min_quality): b -= 1
  return (int32(a), int32(b + 1))

proc add*(s: var Stager, r: roi, fai: Fai, trim_on_host: bool = true, K: int = 27) =
  ## One `roi` (indelope.nim:21) appended to the batch: every field assemble / callsemble read from a Record
  ## (indelope.nim:163-169 sequence + qualities + start, :213-216 stop + mapq, :293-300 sequence + mapq) and one reference
  ## slice that covers every per-contig fai.get of :220.
  var lo = high(int)
  var hi = 0
  var read_seq = ""
  var base_q = new_seq[uint8](300)
  for read in r.reads:
    discard read.sequence(read_seq)
    discard read.base_qualities(base_q)
    for c in read_seq: s.bases.add(uint8(c))
    if trim_on_host:                      # half the bytes to upload: the bounds instead of the qualities
      let (tlo, thi) = trim_bounds(base_q, read_seq.len)
      s.trim_lo.add(tlo); s.trim_hi.add(thi)
    else:
      for i in 0..<read_seq.len: s.quals.add(base_q[i])
    s.read_off.add(int64(s.bases.len))
    s.read_start.add(int64(read.start)); s.read_stop.add(int64(read.stop))
    s.mapq.add(read.qual)
    s.read_skip.add(uint8(read.skippable(allow_unmapped=false)))
    lo = min(lo, read.start); hi = max(hi, read.stop)
  s.region_read_off.add(int64(s.read_start.len))
  # contigs start at a read's (trimmed) start and windows end at max_stop + width + 50 (:218-220)
  let width = int((K + 1) / 2 - 1)
  let w = fai.get(r.reads[0].chrom, lo, hi + width + 50)
  for c in w: s.ref_bases.add(uint8(c))
  s.ref_off.add(int64(s.ref_bases.len)); s.ref_origin.add(int64(lo))

proc clear*(s: var Stager) =
  s.region_read_off.set_len(1); s.read_off.set_len(1); s.ref_off.set_len(1)
  s.read_start.set_len(0); s.read_stop.set_len(0); s.ref_origin.set_len(0)
  s.bases.set_len(0); s.quals.set_len(0); s.mapq.set_len(0); s.read_skip.set_len(0); s.ref_bases.set_len(0)
  s.trim_lo.set_len(0); s.trim_hi.set_len(0)

proc fill*(s: var Stager, b: var IhpBatchIn) =
  b.n_regions = int32(s.ref_origin.len); b.n_reads = int64(s.read_start.len)
  b.region_read_off = addr s.region_read_off[0]; b.read_off = addr s.read_off[0]
  b.bases = addr s.bases[0]; b.quals = if s.quals.len > 0: addr s.quals[0] else: nil
  b.read_start = addr s.read_start[0]; b.read_stop = addr s.read_stop[0]
  b.mapq = addr s.mapq[0]; b.read_skip = addr s.read_skip[0]
  b.ref_off = addr s.ref_off[0]; b.ref_bases = addr s.ref_bases[0]; b.ref_origin = addr s.ref_origin[0]
  if s.trim_lo.len > 0: (b.trim_lo = addr s.trim_lo[0]; b.trim_hi = addr s.trim_hi[0]; b.quals = nil)

proc fill_slab2*(s: var Stager, L: var IhpSlab2Layout, flags: var int32): pointer =
  ## The staged batch as ONE page-locked compact slab (ihp_slab2_layout): what goes over PCIe in a single copy.  Needs the trim
  ## bounds (trim_on_host).  nil when a read does not fit the 16 / 32-bit fields or a base has no 4-bit code -- a lower-case
  ## (soft-masked) window base, any byte outside "=ACMGRSVTWYHKDBN" -- (the caller then uses `fill` + the arrays, whose kernels
  ## take lower case as ksw2.nim:127-132 does).  Every base is checked BEFORE anything is written: find() == -1 never reaches uint8().
  ## A stager that holds hts Records writes `bases4` with a copy of bam_get_seq() per read instead of re-encoding the ASCII.
  let nreg = int32(s.ref_origin.len)
  let nr = int64(s.read_start.len)
  var two_bit = true
  for c in s.ref_bases:
    if c != uint8('A') and c != uint8('C') and c != uint8('G') and c != uint8('T'): two_bit = false
  flags = if two_bit: IHP_SLAB2_REF_2BIT else: 0'i32
  const codes4 = "=ACMGRSVTWYHKDBN"
  var reads_2bit = s.bases.len > 0
  for c in s.bases:
    if codes4.find(char(c)) < 0: return nil
    if c != uint8('A') and c != uint8('C') and c != uint8('G') and c != uint8('T'): reads_2bit = false
  if reads_2bit: flags = flags or IHP_SLAB2_BASES_2BIT     # half the bytes again: one N anywhere in the batch and the 4-bit form travels
  if not two_bit:
    for c in s.ref_bases:
      if codes4.find(char(c)) < 0: return nil      # e.g. hg19 / hg38 soft-masked lower case: the arrays carry it as it is
  if ihp_slab2_layout_for(nreg, nr, int64(s.bases.len), int64(s.ref_bases.len), flags, addr L) != 0: return nil
  result = ihp_host_alloc(csize_t(L.bytes))
  if result == nil: return nil
  let m = cast[ptr UncheckedArray[uint8]](result)
  template at(T: typedesc, off: int64): untyped = cast[ptr UncheckedArray[T]](addr m[off])
  for r in 0..nreg:
    at(int64, L.region_read_off)[r] = s.region_read_off[r]
    at(int64, L.region_base_off)[r] = s.read_off[int(s.region_read_off[r])]
    at(int64, L.ref_off)[r] = s.ref_off[r]
  var r = 0
  for i in 0..<int(nr):
    while int64(i) >= s.region_read_off[r + 1]: r += 1
    let rel = s.read_start[i] - s.ref_origin[r]
    let ln = s.read_off[i + 1] - s.read_off[i]
    let sp = s.read_stop[i] - s.read_start[i]
    if rel < int64(low(int32)) or rel > int64(high(int32)) or ln > 65535 or sp < 0 or sp > 65535: (ihp_host_free(result); return nil)
    at(int32, L.start_rel)[i] = int32(rel)
    at(uint16, L.len)[i] = uint16(ln); at(uint16, L.span)[i] = uint16(sp)
    at(uint16, L.trim_lo)[i] = uint16(s.trim_lo[i]); at(uint16, L.trim_hi)[i] = uint16(s.trim_hi[i])
    at(uint8, L.mapq)[i] = s.mapq[i]; at(uint8, L.rflags)[i] = s.read_skip[i] and 1
    if reads_2bit:
      # the library's packed form: read i from 32-bit word (read_off[i] >> 4) + i, base j in bits 2 (j and 15) of word j shr 4,
      # code (ASCII shr 1) and 3 (A 0, C 1, T 2, G 3)
      let w0 = (s.read_off[i] shr 4) + int64(i)
      for j in 0..<int(ln):
        let code = (uint32(s.bases[int(s.read_off[i]) + j]) shr 1) and 3'u32
        if (j and 15) == 0: at(uint32, L.bases4)[w0 + int64(j shr 4)] = 0
        at(uint32, L.bases4)[w0 + int64(j shr 4)] = at(uint32, L.bases4)[w0 + int64(j shr 4)] or (code shl (2 * (j and 15)))
    else:
      # 4-bit bases: read i from byte (read_off[i] >> 1) + i, first base in the high nibble (BAM's own packing)
      let dst = L.bases4 + (s.read_off[i] shr 1) + int64(i)
      for j in 0..<int(ln):
        let code = uint8("=ACMGRSVTWYHKDBN".find(char(s.bases[int(s.read_off[i]) + j])))
        if (j and 1) == 0: m[dst + int64(j shr 1)] = code shl 4
        else: m[dst + int64(j shr 1)] = m[dst + int64(j shr 1)] or code
  for r in 0..<int(nreg):
    at(int64, L.ref_origin)[r] = s.ref_origin[r]
    let f0 = s.ref_off[r]
    for j in 0..<int(s.ref_off[r + 1] - f0):
      let c = char(s.ref_bases[int(f0) + j])
      if two_bit:
        let dst = L.ref_packed + (f0 shr 2) + int64(r) + int64(j shr 2)
        if (j and 3) == 0: m[dst] = 0
        m[dst] = m[dst] or (uint8("ACGT".find(c)) shl (2 * (j and 3)))
      else:
        let dst = L.ref_packed + (f0 shr 1) + int64(r) + int64(j shr 1)
        let code = uint8("=ACMGRSVTWYHKDBN".find(c))
        if (j and 1) == 0: m[dst] = code shl 4
        else: m[dst] = m[dst] or code

proc run*(s: var Stager, p: var IhpParams, outp: var IhpBatchOut): cint =
  var b: IhpBatchIn
  s.fill(b)
  result = ihp_run_regions(addr p, addr b, addr outp)


# ---- ONE host thread, several batches in flight (what bench.py measures as `sustained_one_thread`) -------------------------------
# ihp_batch_upload_slab2 and ihp_batch_run only enqueue (the slab's copy, the kernels), so the thread that walks gen_roi can hand
# over batch k+2 while batch k+1 runs and batch k's results travel back: three in flight keep the copy engine and the compute
# units busy from one thread (5.0-5.7 M regions/s on C2 against 2.9 M one batch at a time).  The stager's arrays of a batch must
# outlive it (ihp_call_variants reads them when the batch is collected), so a caller keeps one Stager per batch in flight.
type
  InFlight* = object
    b*: ptr IhpBatch
    slab*: pointer                       ## from ihp_host_alloc: untouched until the batch's first wait has returned
  Pipeline* = object
    q*: seq[InFlight]                    ## oldest first
    depth*: int                          ## 3: one batch uploading, one running, one on its way back

proc submit*(pl: var Pipeline, s: var Stager, p: var IhpParams): cint =
  ## The staged batch enqueued: compact slab, upload, run.  Returns at once.  (A batch that does not fit the slab's 16 / 32-bit
  ## fields: collect everything in flight, then `s.run`.)
  var L: IhpSlab2Layout
  var flags: int32
  let slab = s.fill_slab2(L, flags)
  if slab == nil: return IHP_E_ARG
  var b: ptr IhpBatch
  result = ihp_batch_upload_slab2(addr p, int32(s.ref_origin.len), int64(s.read_start.len), slab, addr L, flags, addr b)
  if result == 0:
    discard ihp_batch_set_fetch(b, IHP_FETCH_NO_BASES or IHP_FETCH_EAGER)
    result = ihp_batch_run(b)
    if result != 0: ihp_batch_free(b)
  if result != 0:
    ihp_host_free(slab)
    return
  pl.q.add InFlight(b: b, slab: slab)

proc collect*(pl: var Pipeline, outp: var IhpBatchOut): cint =
  ## Waits for the OLDEST batch and fetches its results (ihp_free_out when done with them); its slab is free again.
  let f = pl.q[0]
  pl.q.delete(0)
  result = ihp_batch_fetch(f.b, addr outp)
  ihp_batch_free(f.b)
  ihp_host_free(f.slab)

# the sweep:   for every full stager:  if pl.q.len == pl.depth: (collect the oldest, print its variants);  pl.submit(stager, p)
#              at the end:             while pl.q.len > 0: collect, print

# ---- the main loop of indelope.nim:601-608 over batches ---------------------------------------------------------
# `flush` runs one staged batch and prints what the reference's loop would have printed for those regions, in region
# order, through the same last-two-variants window (:604-608), which has to survive from one flush to the next.
type Printed* = tuple[chrom: string, start: int64, refa: string, alta: string]

proc flush*(s: var Stager, pending: var seq[roi], p: var IhpParams, last_var, last_var2: var Printed): cint =
  if pending.len == 0: return 0
  var outp: IhpBatchOut
  result = s.run(p, outp)
  if result != 0:
    stderr.write_line("indelope_hip: " & $ihp_strerror(result) & " / " & $ihp_last_hip_error())
    return
  # filters, qual scalings, INFO and REF/ALT of indelope.nim:375-428 for every tallied event of the batch
  var b: IhpBatchIn
  s.fill(b)
  var vars: IhpVariants
  result = ihp_call_variants(addr p, addr b, addr outp, addr vars)
  if result == 0:
    let vs = cast[ptr UncheckedArray[IhpVariant]](vars.v)
    var buf = new_string(4096)
    for i in 0..<int(vars.n):
      if vs[i].filter != IHP_VF_EMITTED: continue        # ihp_call_variants dedupes inside a batch; across batches here
      let chrom = pending[int(vs[i].region)].reads[0].chrom
      let refa = ($vars.chars)[int(vs[i].ref_off)..<int(vs[i].ref_off + vs[i].ref_len)]
      let alta = ($vars.chars)[int(vs[i].alt_off)..<int(vs[i].alt_off + vs[i].alt_len)]
      let cur: Printed = (chrom, vs[i].start, refa, alta)
      if cur == last_var or cur == last_var2: continue   # v.same(last_var) / v.same(last_var2), :604-605
      let n = ihp_format_variant(addr vs[i], vars.chars, chrom.cstring, buf.cstring, int64(buf.len))
      echo buf[0..<int(n)]                               # :606
      last_var2 = last_var; last_var = cur               # :607-608
    ihp_free_variants(addr vars)
  ihp_free_out(addr outp)
  s.clear(); pending.set_len(0)


# ---- eight processes, one chromosome each ... or one target list cut in eight: the sweep on a node of MI355X ----------------------
# Regions are independent past gen_roi (indelope.nim:601-603), so every rank walks ITS contiguous range of the targets with the
# Stager / Pipeline above and nothing crosses ranks until the end.  What the reference's main loop does with a region's result
# is sequential -- the last-two-variants window (:604-608) -- so the results have to meet on one rank IN REGION ORDER: that is
# the one collective of the path, and `MultiGpuSweep` is all of it.  A launcher starts the ranks (one per GPU: mpirun, srun, a
# shell loop) and gives each its rank, the world size and a path every rank can read; rank 0 leaves the communicator's id there.
import os                                  # file_exists, move_file, sleep (indelope.nim imports os already, :1-18)
type MultiGpuSweep* = object
  d*: ptr IhpDist
  rank*, world*: int32

proc open_sweep*(rank, world: int32, id_path: string): MultiGpuSweep =
  ## ihp_init(rank) binds GPU `rank` of the node; the 128-byte id travels through `id_path`.
  doAssert ihp_init(cint(rank)) == 0
  var id: array[IHP_DIST_ID_BYTES, uint8]
  if rank == 0:
    doAssert ihp_dist_unique_id(addr id[0], int64(id.len)) == 0
    var f = open(id_path & ".tmp", fmWrite)
    discard f.write_buffer(addr id[0], id.len); f.close()
    move_file(id_path & ".tmp", id_path)                 # (os) the others never see half a file
  else:
    while not file_exists(id_path): sleep(10)
    var f = open(id_path, fmRead)
    doAssert f.read_buffer(addr id[0], id.len) == id.len; f.close()
  result.rank = rank; result.world = world
  doAssert ihp_dist_init(rank, world, addr id[0], int64(id.len), addr result.d) == 0

proc gather_results*(m: var MultiGpuSweep, b: ptr IhpBatch): seq[IhpBatchOut] =
  ## Every rank calls this with its (last) batch; rank 0 gets one IhpBatchOut per rank in rank = region order (ihp_free_out each),
  ## the others an empty seq.  Rank 0 then walks them exactly as `flush` walks one: ihp_call_variants per rank's batch needs that
  ## rank's inputs, so in practice every rank calls ihp_call_variants on its OWN results first and only the printable records
  ## travel -- or, as here, the packed results travel and rank 0 holds every rank's stager (the targets are known to all).
  if m.rank == 0: result = new_seq[IhpBatchOut](int(m.world))
  let rc = ihp_dist_gather_payload(m.d, b, 0, if m.rank == 0: addr result[0] else: nil, nil)
  doAssert rc == 0, $ihp_strerror(rc) & " / " & $ihp_last_hip_error()

proc gather_records*(m: var MultiGpuSweep, b: ptr IhpBatch, n_regions_total: int): seq[IhpRegionSummary] =
  ## The fixed-size half: 32 bytes per region (status, contigs, events, the first tallied event's ref / alt support) to rank 0.
  if m.rank == 0: result = new_seq[IhpRegionSummary](n_regions_total)
  var n_total: int64
  let rc = ihp_dist_gather_summaries(m.d, b, 0, nil, if m.rank == 0: addr result[0] else: nil, int64(n_regions_total), addr n_total, nil)
  doAssert rc == 0, $ihp_strerror(rc) & " / " & $ihp_last_hip_error()
  if m.rank == 0: result.set_len(int(n_total))

proc close*(m: var MultiGpuSweep) =
  discard ihp_dist_finalize(m.d); m.d = nil
