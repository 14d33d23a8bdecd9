/*
 * indelope_hip.h -- C ABI of the MI355X (gfx950) implementation of indelope's
 * per-region hot path: slide-assembly (contig.nim), ksw2 contig->reference
 * alignment (ksw2_extz2_sse.c) and the ref/alt k-mer tally (indelope.nim:283-311).
 *
 * Every entry point is plain C: pointers, sizes, POD structs.  No C++ or torch
 * types cross this boundary.  All functions return 0 (IHP_OK) or a negative
 * IHP_E_* code; nothing aborts or throws across the ABI.  `ksw_extz2_sse`
 * keeps the reference's void signature (ksw2.h:54).
 *
 * Citations "file:line" are into the reference tree (brentp/indelope).
 */
#ifndef INDELOPE_HIP_H_
#define INDELOPE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ errors */
#define IHP_OK            0
#define IHP_E_NODEVICE   (-1)  /* no gfx950 device / HIP runtime unusable       */
#define IHP_E_HIP        (-2)  /* a HIP call failed (see ihp_last_hip_error)    */
#define IHP_E_ARG        (-3)  /* bad argument (null pointer, negative size...) */
#define IHP_E_NOMEM      (-4)  /* host or device allocation failed              */
#define IHP_E_CAPACITY   (-5)  /* caller-provided buffer too small              */
#define IHP_E_UNSUPPORTED (-6) /* parameter combination the entry point does not take (e.g. KSW_EZ_SCORE_ONLY on the per-region path) */

const char *ihp_strerror(int code);
const char *ihp_last_hip_error(void);   /* text of the last failing HIP call    */
const char *ihp_version(void);

/* Bind the calling process to one GPU (one process per GPU).  Idempotent.
 * Environment (the caller's to set, before the process's first HIP call): GPU_MAX_HW_QUEUES=16.  Every batch in flight
 * drives two streams, and the HIP runtime folds all streams of a process onto that many hardware queues (default 4): with
 * more than two batches about, launch chains that should overlap share a queue every few runs and run one after the other
 * (-15 % measured).  The library does not set it itself (setenv is not thread safe and comes too late once HIP is up). */
int ihp_init(int device);
int ihp_device_info(int *cu_count, int *wave_size, int64_t *hbm_bytes);
void ihp_shutdown(void);

/* ------------------------------------------------- ksw2 (L2b) : the FFI seam */
/* ksw2.h:6-16 */
#define KSW_NEG_INF        (-0x40000000)
#define KSW_EZ_SCORE_ONLY  0x01
#define KSW_EZ_RIGHT       0x02
#define KSW_EZ_GENERIC_SC  0x04
#define KSW_EZ_APPROX_MAX  0x08
#define KSW_EZ_APPROX_DROP 0x10
#define KSW_EZ_SPLICE_FOR  0x100   /* ignored, as by ksw2_extz2_sse.c */
#define KSW_EZ_SPLICE_REV  0x200
#define KSW_EZ_EXTZ_ONLY   0x40
#define KSW_EZ_REV_CIGAR   0x80

/* ksw2.h:22-30 / ksw2_c.nim:18-30 -- identical layout (48 bytes on LP64). */
typedef struct {
	uint32_t max:31, zdropped:1;
	int max_q, max_t;
	int mqe, mqe_t;
	int mte, mte_q;
	int score;
	int m_cigar, n_cigar;
	uint32_t *cigar;
} ksw_extz_t;

/*
 * Symbol-compatible replacement of the reference's only native entry point
 * (ksw2.h:54, bound by ksw2_c.nim:53-55, sole caller ksw2.nim:154-157).
 * One alignment, computed by the HIP kernel.  `km` is ignored (always nil in
 * the reference).  ez->cigar is (re)allocated with realloc and retained by the
 * caller exactly as ksw2_extz2_sse.c:31-41 does.  The reference signature has
 * no return code, so a failure (HIP error, unsupported flag, out of memory or
 * capacity) is made loud instead of looking like "no alignment": ez is left
 * reset (n_cigar 0, score KSW_NEG_INF), the reason is printed on stderr and
 * kept for ihp_last_hip_error(), ihp_ksw_last_status() returns the IHP_E_* code
 * of the most recent call (0 = ok), and with IHP_KSW_STRICT=1 in the
 * environment the call abort()s the way the reference's own assert
 * (ksw2_extz2_sse.c:237) would.
 */
void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                   ksw_extz_t *ez);

int ihp_ksw_last_status(void);

/* Scalar result fields of one alignment (ksw_extz_t minus the pointer).       */
typedef struct {
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
} ihp_ez;

/*
 * Batched form of the same call: n independent (query,target) pairs, encoded
 * 0..m-1 (ksw2.nim:129-132), concatenated; q_off/t_off have n+1 entries.
 * Results: ez[n]; CIGARs (len<<4|op) concatenated in pair order into `cigar`
 * (capacity cigar_cap words) with cigar_off[n+1].  IHP_E_CAPACITY if the
 * CIGARs do not fit (cigar_off[n] then holds the required size).
 */
int ihp_ksw_extz2_batch(int32_t n, const uint8_t *queries, const int64_t *q_off,
                        const uint8_t *targets, const int64_t *t_off,
                        int8_t m, const int8_t *mat, int8_t q, int8_t e,
                        int w, int zdrop, int flag,
                        ihp_ez *ez, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_off);

/* ksw2.nim:129-132 (encode) and :135-140 (matrix): host helpers.              */
void ihp_encode(const uint8_t *dna, int64_t n, uint8_t *out);
void ihp_matrix(int8_t match, int8_t mismatch, int8_t out25[25]);

/* --------------------------------------------- contig.nim (L2a) : Contig API */
#define IHP_UNALIGNED INT64_MIN          /* contig.nim:27  `unaligned = low(int)` */

/* contig.nim:7-15.  Caller-owned buffers; `cap` = elements allocated in both. */
typedef struct {
	uint8_t  *sequence;
	uint32_t *support;
	int64_t   len;
	int64_t   cap;
	int64_t   nreads;
	int64_t   start;
} ihp_contig;

/* contig.nim:17 */
typedef struct { int64_t qoff, toff; int32_t qbest; int32_t _pad; } ihp_correction;

/* contig.nim:21.  `corrections` is caller-owned with capacity corr_cap.       */
typedef struct {
	int64_t matches, offset, mismatches;
	int64_t n_corrections;
	int64_t contig_i;
	ihp_correction *corrections;
	int64_t corr_cap;
} ihp_match;

/* allowable_mismatch_fn (contig.nim:25) cannot cross to the device as a
 * closure; the two rules the reference ever passes are selectable.            */
#define IHP_ALLOW_DEFAULT 0   /* contig.nim:44-47  */
#define IHP_ALLOW_SUPPORT 1   /* contig.nim:287-290 (test rule `allow_test`)    */

/* slide_align (contig.nim:70-141).  IHP_E_CAPACITY if corr_cap is too small
 * (n_corrections then holds the required count).                             */
int ihp_slide_align(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                    int64_t max_mismatch, int allow_rule, ihp_match *out);
/* insert(t,q,m) (contig.nim:156-222): mutates t AND q.  t->cap must hold the
 * merged contig (IHP_E_CAPACITY otherwise, nothing modified).                 */
int ihp_contig_insert(ihp_contig *t, ihp_contig *q, const ihp_match *m);
/* trim(c,min_support) (contig.nim:49-68).                                     */
int ihp_contig_trim(ihp_contig *c, int64_t min_support);

/* ------------------------------------------------- k-mer tally (L2c, G1-G2) */
/* One event: reads concatenated (ASCII), read_off[n+1], mapq[n].  Reads with
 * mapq < min_mapq are skipped (indelope.nim:294).  counts = {ref_support,
 * alt_support, both_found} (indelope.nim:301-311).  K <= 31.                  */
int ihp_kmer_tally(int32_t n_reads, const uint8_t *bases, const int64_t *read_off,
                   const uint8_t *mapq, int32_t min_mapq, int32_t K,
                   const char *ref_kmer, const char *alt_kmer, int32_t counts[3]);

/* ------------------------------------------------ genotyper.nim (G3, host fp64) */
#define IHP_GT_HOM_REF 0
#define IHP_GT_HET     1
#define IHP_GT_HOM_ALT 2
#define IHP_GT_UNKNOWN 3
typedef struct { int32_t gt; int32_t _pad; double gl[3]; } ihp_genotype_t;  /* genotyper.nim:14 */
int    ihp_genotype(int64_t r, int64_t a, double error, ihp_genotype_t *out); /* :36-47 */
double ihp_genotype_qual(const ihp_genotype_t *g);                            /* :22-29 */

/* ------------------------------------------- batched per-region path (L3 ★) */
/* Every default argument on the path (SURVEY.md §5 "Config / flags").         */
typedef struct {
	int32_t struct_size;          /* sizeof(ihp_params); checked                 */
	double  min_overlap_pct;      /* 0.88   indelope.nim:157                     */
	int32_t min_mapq_assemble;    /* 20     indelope.nim:157,164                 */
	int32_t min_mapq_stop;        /* 5      indelope.nim:215 (qual <= 5 skipped) */
	int32_t min_mapq_tally;       /* 10     indelope.nim:294                     */
	int32_t trim_min_qual;        /* 15     indelope.nim:23                      */
	int32_t combine_min_support;  /* 3      indelope.nim:176                     */
	int32_t combine_min_overlap;  /* 65     contig.nim:224 (best_match default)  */
	int32_t max_mismatch;         /* 0      contig.nim:243,254                   */
	int32_t max_pre_contigs;      /* 20     indelope.nim:209                     */
	int32_t min_ctg_len;          /* 74     indelope.nim:201 (CLI: 73)           */
	int32_t min_reads;            /* 4      indelope.nim:201 (CLI: 3)            */
	int32_t min_event_len;        /* 4      indelope.nim:201                     */
	int32_t K;                    /* 27     indelope.nim:201                     */
	int32_t max_events;           /* 4      indelope.nim:229                     */
	int32_t ref_pad;              /* 50     indelope.nim:220                     */
	int8_t  match, mismatch, gap_open, gap_ext; /* 1,-2,4,1  ksw2.nim:142        */
	int32_t bw, zdrop, ksw_flag;  /* 50,400,0  indelope.nim:221                  */
	double  error;                /* 1e-3   indelope.nim:379                     */
	/* alignment-fallback genotyper (indelope.nim:312-372): when an event's k-mer tally
	 * has both_found > 0 every read is aligned to the reference window and to the contig */
	int32_t fallback;             /* 1      run it (0: only flag fallback_needed)  */
	int8_t  fb_match, fb_mismatch, fb_gap_open, fb_gap_ext; /* 1,-2,5,1 indelope.nim:318-319 */
	int32_t fb_bw, fb_zdrop, fb_flag;  /* -1,-1,0  align_to defaults ksw2.nim:159 */
} ihp_params;

void ihp_params_default(ihp_params *p);

/*
 * One batch of candidate regions, flat caller-owned host buffers (what the
 * BAM sweep's `roi` tuples -- indelope.nim:21 -- decode to).  Reads of region
 * r are region_read_off[r] .. region_read_off[r+1]-1, in BAM order.
 */
typedef struct {
	int32_t n_regions;
	int64_t n_reads;
	const int64_t *region_read_off;   /* [n_regions+1]                           */
	const int64_t *read_off;          /* [n_reads+1] into bases / quals          */
	const uint8_t *bases;             /* ASCII, Record.sequence                  */
	const uint8_t *quals;             /* phred, Record.base_qualities; NULL = 255 */
	const int64_t *read_start;        /* Record.start (0-based)                  */
	const int64_t *read_stop;         /* Record.stop  (bam_endpos)               */
	const uint8_t *mapq;              /* Record.qual                             */
	const uint8_t *read_skip;         /* skippable(r) (indelope.nim:40-47); NULL = 0 */
	const int64_t *ref_off;           /* [n_regions+1] into ref_bases            */
	const uint8_t *ref_bases;         /* per-region slice of the chromosome      */
	const int64_t *ref_origin;        /* [n_regions] genomic pos of slice[0]; window
	                                     requests (indelope.nim:220) are clamped to
	                                     the slice the way faidx clamps to a
	                                     chromosome, and flagged IHP_ALN_REF_CLAMPED */
	/* Optional: the result of trim(sequence, base_qualities) (indelope.nim:23-38) computed by
	 * the stager -- read i keeps bases [trim_lo[i], trim_hi[i]) and its start moves by
	 * trim_lo[i] (an emptied read: lo == hi == the returned `a`).  When given, `quals` is not
	 * read and need not be uploaded: half the bytes of a batch.  NULL: trim from `quals`.    */
	const int32_t *trim_lo, *trim_hi; /* [n_reads]                               */
} ihp_batch_in;

/*
 * The same batch as ONE caller-filled slab -- what a stager that wants the copy engines to itself hands over: allocate
 * `bytes` with ihp_host_alloc (page-locked), fill the sections at the byte offsets below, and ihp_batch_upload_slab sends
 * the whole slab with a single asynchronous copy.  The read bases are 4 bits each, exactly as a BAM record stores them
 * (Record.sequence decodes these, src/indelope.nim:166): read i starts at byte (read_off[i] >> 1) + i of `bases4`, first
 * base in the high nibble, codes "=ACMGRSVTWYHKDBN" -- a memcpy of bam_get_seq(), half the bytes of the
 * ASCII form.  The device writes the ASCII bases it needs itself.  The read trim (indelope.nim:23-38) comes as bounds
 * (trim_lo / trim_hi, as in ihp_batch_in); base qualities are not part of a slab.
 */
typedef struct {
	int64_t region_read_off, read_off, read_start, read_stop, ref_off, ref_origin;   /* int64 sections, as in ihp_batch_in */
	int64_t trim_lo, trim_hi;                                                        /* int32 [n_reads]                    */
	int64_t mapq, read_skip;                                                         /* uint8 [n_reads]; read_skip only with IHP_SLAB_HAS_SKIP */
	int64_t ref_bases;                                                               /* ASCII                              */
	int64_t bases4;                                                                  /* (n_bases >> 1) + n_reads bytes      */
	int64_t bytes;                                                                   /* size of the slab                   */
} ihp_slab_layout;
#define IHP_SLAB_HAS_SKIP 1
int  ihp_slab_layout_for(int32_t n_regions, int64_t n_reads, int64_t n_bases, int64_t n_ref, ihp_slab_layout *out);

/*
 * Round 5: the COMPACT slab -- the same batch in about three quarters of the bytes (C2: 59 MB instead of 75.6), because the
 * slab's copy over PCIe is what bounds a sweep that keeps the GPU fed (the copy of the form above takes longer than the
 * kernels need for the batch).  What a `roi` of src/indelope.nim:21 holds per read, in the narrowest types that hold it:
 *   start_rel  int32   Record.start - ref_origin[region]           (a region's reads lie within +-2^31 of its window)
 *   len        uint16  length of Record.sequence                   (read_off becomes a prefix sum, made on the device)
 *   span       uint16  Record.stop - Record.start                  (read_stop = start + span)
 *   trim_lo/hi uint16  the kept range of trim(sequence, base_qualities), src/indelope.nim:23-38, as in ihp_batch_in
 *   mapq       uint8   Record.qual
 *   rflags     uint8   bit 0: skippable(r), src/indelope.nim:40-47
 * = 14 bytes per read instead of 34.  Per region: region_read_off, region_base_off (read_off of the region's first read: the
 * stager has it, it places the bases by it), ref_off, ref_origin as int64.  The reference windows 4 bits per base in BAM's code
 * ("=ACMGRSVTWYHKDBN"; region r from byte (ref_off[r] >> 1) + r, first base in the high nibble), or -- IHP_SLAB2_REF_2BIT,
 * when every window base is one of A C G T -- 2 bits per base (A C G T = 0 1 2 3; region r from byte (ref_off[r] >> 2) + r,
 * first base in the low bits).  The read bases as in the slab above.  A k_slab_expand launch in front of the batch's first
 * run writes the arrays of ihp_batch_in from it (and checks that a region's lengths add up to its region_base_off step:
 * ihp_batch_sync reports IHP_E_ARG otherwise).
 */
typedef struct {
	int64_t region_read_off, region_base_off, ref_off, ref_origin;                   /* int64 [n_regions + 1] x 3, [n_regions] */
	int64_t start_rel;                                                               /* int32 [n_reads]                    */
	int64_t len, span, trim_lo, trim_hi;                                             /* uint16 [n_reads]                   */
	int64_t mapq, rflags;                                                            /* uint8 [n_reads]                    */
	int64_t ref_packed;                                                              /* (n_ref >> 1) + n_regions bytes, or (n_ref >> 2) + n_regions */
	int64_t bases4;                                                                  /* (n_bases >> 1) + n_reads bytes; IHP_SLAB2_BASES_2BIT: 4 ((n_bases >> 4) + n_reads + 4) */
	int64_t bytes;
} ihp_slab2_layout;
#define IHP_SLAB2_REF_2BIT 2
/* Round 6: the READ bases 2 bits each too -- possible when every base of every read of the batch is upper-case A C G T (the
 * usual batch; one N anywhere and the stager writes the 4-bit form).  The section `bases4` then holds the library's own packed
 * form as it is: 32-bit little-endian words, read i from word (read_off[i] >> 4) + i (read_off = the running sum of `len`
 * from region_base_off), sixteen bases per word, base j of the read in bits 2 (j & 15) .. of word j >> 4, code = (ASCII >> 1) & 3
 * (A 0, C 1, T 2, G 3), unused bits zero.  Half the bytes of the 4-bit form (C2: 59 -> 35 MB per batch), and the device has
 * nothing left to pack: it only writes the ASCII copy the byte-based kernels read.                                            */
#define IHP_SLAB2_BASES_2BIT 4
int  ihp_slab2_layout_for(int32_t n_regions, int64_t n_reads, int64_t n_bases, int64_t n_ref, int32_t flags, ihp_slab2_layout *out);

/* event status: why the tally did or did not run for an alignment event.      */
#define IHP_EV_TALLIED     0
#define IHP_EV_SHORT       1   /* tloc.len < min_event_len     indelope.nim:234 */
#define IHP_EV_SAME_KMER   2   /* indelope.nim:264                              */
#define IHP_EV_LOW_CPLX    3   /* indelope.nim:266                              */
#define IHP_EV_BUG_SAME    4   /* indelope.nim:268-275                          */
#define IHP_EV_OOB         5   /* k-mer slice would index out of range (the
                                  reference would raise/UB); never tallied      */
#define IHP_EV_NON_ACGT    6   /* ref/alt k-mer holds a non-ACGT byte: `kmer`
                                  package behaviour unpinned; never tallied     */

typedef struct {
	int64_t tstart, tstop;        /* target_locations(ctg.start) ksw2.nim:71-80  */
	int64_t qstart, qstop;        /* query_locations()           ksw2.nim:82-91  */
	uint32_t len;
	uint8_t  type;                /* 0 Insertion, 1 Deletion     ksw2.nim:65-67  */
	uint8_t  status;              /* IHP_EV_*                                    */
	uint8_t  fallback_needed;     /* k-mer both_found > 0 (indelope.nim:313)     */
	uint8_t  aligned;             /* `aligned`: the fallback ran (indelope.nim:372) */
	int32_t  cf_offset;           /* `offset`, indelope.nim:243                  */
	/* the reference's variables as they stand at indelope.nim:375: the k-mer tally
	 * (:285-311), or the alignment votes (:353-356) with both_found reset (:316)
	 * when `aligned`                                                              */
	int32_t  ref_support, alt_support, both_found;
	char     ref_kmer[32], alt_kmer[32];             /* NUL padded               */
	int32_t  gt;                  /* genotype(ref,alt,error) indelope.nim:379    */
	int32_t  kmer_ref_support, kmer_alt_support, kmer_both_found;  /* the k-mer tally
	                                 itself (:285-311), kept when the fallback ran */
	double   gl[3];
	double   qual;
} ihp_event;

/* per-contig alignment flags */
#define IHP_ALN_DONE        1   /* align_to ran (passed indelope.nim:209-211)    */
#define IHP_ALN_REF_CLAMPED 2   /* window request fell outside the given slice   */

/*
 * Results, allocated by the library (free with ihp_free_out).  Final contigs
 * of region r are contig_off[r] .. contig_off[r+1]-1, in `combine` order.
 */
typedef struct {
	int32_t  n_regions;
	int64_t  n_contigs, n_events, n_cigar_words, n_bases, n_hits;
	int32_t *status;              /* [R] IHP_OK or IHP_E_*                       */
	int32_t *n_contigs_pre;       /* [R] assemble's n_contigs, indelope.nim:171  */
	int64_t *contig_off;          /* [R+1]                                       */
	/* per final contig */
	int64_t *ctg_start;           /* Contig.start                                */
	int64_t *ctg_nreads;          /* Contig.nreads                               */
	int64_t *ctg_seq_off;         /* [C+1] into ctg_seq / ctg_support            */
	uint8_t *ctg_seq;
	uint32_t *ctg_support;
	int32_t *aln_flags;           /* IHP_ALN_*                                   */
	int64_t *aln_ref_start;       /* window actually aligned to: start, length   */
	int32_t *aln_ref_len;
	ihp_ez  *aln_ez;
	int64_t *cigar_off;           /* [C+1] into cigar (full CIGAR)               */
	uint32_t *cigar;
	int64_t *event_off;           /* [C+1] into events                           */
	ihp_event *events;
	/* Per tallied event, one entry per read of its region (BAM order): the start index in the
	 * (untrimmed) read of the first k-mer whose canonical code is the ref / alt k-mer
	 * (indelope.nim:301-309), or -1 (no such k-mer, or the read was not examined: mapq < 10,
	 * :294).  This is what `rdists/adists/rmapqs/amapqs` (:302-309) are built from: the `kmer`
	 * package's distance `d` is a property of that window, the mapq is the read's.  Events
	 * that were not tallied have no entries.                                               */
	int64_t *hit_off;             /* [E+1] into ref_hit / alt_hit                */
	int32_t *ref_hit, *alt_hit;
	/* IHP_FETCH_COMPACT (round 6): the contigs' bases and supports in 1.5 bytes per base instead of 5 -- ctg_seq and
	 * ctg_support are then NULL.  ctg_seq4: BAM's 4-bit codes ("=ACMGRSVTWYHKDBN"), contig c from byte
	 * (ctg_seq_off[c] >> 1) + c, first base in the high nibble; ctg_sup8[i]: support of base i of the flat base array,
	 * 255 = look it up among the n_sup_escapes (sup_escape_idx ascending, sup_escape_val).  ihp_out_contig expands one
	 * contig; ihp_call_variants takes either form (it needs the bases of the few contigs a variant is emitted from).    */
	uint8_t *ctg_seq4, *ctg_sup8;
	int64_t  n_sup_escapes;
	int64_t *sup_escape_idx;
	uint32_t *sup_escape_val;
} ihp_batch_out;

/* Page-locked host memory for the caller's flat batch arrays: uploads from it are DMA
 * transfers that overlap other batches' kernels (every batch runs on its own stream, and
 * batches may be driven from several host threads at once).  Pageable memory works too,
 * through the runtime's staging copies.  NULL on failure.                                 */
void *ihp_host_alloc(size_t bytes);
void  ihp_host_free(void *p);
/* Device -> host copy of memory this library handed out as a device pointer (ihp_batch_pack_dev, ihp_batch_summary_dev),
 * for callers that do not link the HIP runtime themselves.                                                         */
int   ihp_copy_to_host(const void *dev_ptr, int64_t bytes, void *out);

/* ------------------------------------- post-tally filters and Variant records (row f2) */
/* indelope.nim:375-428 (the filters and Variant fields that follow the tally inside
 * callsemble) and :604-608 (the last-two-variants dedupe of the main loop), as host code over
 * one batch's inputs and results.  Every tallied event yields one record; `filter` says
 * whether the reference would have printed it, and if not which test dropped it (the first
 * one in the reference's order).  AKE/RKE and the :412 filter use the `kmer` package's
 * distance `d`, whose definition is not pinned here (indelope.nimble:10-11): it is taken as
 * the distance of the k-mer window from the closer end of the read, min(i, len - K - i).      */
#define IHP_VF_EMITTED      0
#define IHP_VF_LOW_ALT      2   /* alt_support < min_reads                    :375 */
#define IHP_VF_LOW_FRAC     3   /* alt_support / reads.len < 0.1              :377 */
#define IHP_VF_HOM_REF      4   /*                                            :380 */
#define IHP_VF_BOTH_AT_EDGE 5   /* offset == 0 and both_found >= 0.75 min(ref, alt)  :384 */
#define IHP_VF_SMALL_FLANK  6   /* min_flank - 1 < event span                 :399 */
#define IHP_VF_KMER_AT_END  7   /* mean(adists) < 5                           :412 */
#define IHP_VF_HOMOPOLYMER  8   /* homopolymer insertion in homopolymer k-mers :423-427 */
#define IHP_VF_DUPLICATE    9   /* same as one of the last two printed        :604-608 */
#define IHP_VF_OOB         10   /* allele slice outside the contig / the region's reference slice
                                   (the reference would raise); never printed */
typedef struct {
	int32_t region, contig;       /* indices into ihp_batch_out                  */
	int64_t event;
	int32_t filter;               /* IHP_VF_*                                    */
	int32_t gt;                   /* IHP_GT_*                                    */
	int64_t start;                /* Variant.start = tloc.start                  */
	double  qual;                 /* genotype qual after the scalings of :386-404 */
	double  gq, gl[3];            /* Genotype.qual, GL (the GT:GQ:GL sample column) */
	double  ake, rke;             /* mean(adists), mean(rdists); NaN when empty  */
	int32_t ad[2];                /* ref_support, alt_support                    */
	int32_t dp, bs, mf, cf, nc;   /* DP, BS (0: absent), MF, CF, NC              */
	int32_t amq, rmq;             /* median mapq of the alt / ref reads; -1: absent */
	uint8_t lo, al, event_type, _pad;   /* LO and AL flags; 0 insertion, 1 deletion */
	int32_t ref_len, alt_len, cc_len;   /* REF, ALT, CC strings in `chars`         */
	int64_t ref_off, alt_off, cc_off;
	char    ref_kmer[32], alt_kmer[32];
} ihp_variant;

typedef struct {
	int64_t n;                    /* one record per tallied event, in output order */
	ihp_variant *v;
	int64_t n_chars;
	char *chars;
} ihp_variants;

int  ihp_call_variants(const ihp_params *p, const ihp_batch_in *in, const ihp_batch_out *out, ihp_variants *vars);
void ihp_free_variants(ihp_variants *vars);
/* `$`(v) of indelope.nim:104-113: one VCF line (no newline) into buf; returns the length
 * needed (excluding the NUL), or a negative IHP_E_* code.                                  */
int64_t ihp_format_variant(const ihp_variant *v, const char *chars, const char *chrom, char *buf, int64_t cap);

/* ------------------------------------------------ ROI evidence scan (row f4, the caller's side) */
/* gen_roi / gen_roi_internal / event_locations (indelope.nim:430-445, :461-545) for one run of reads of
 * one target, in BAM order: every non-match CIGAR op of a non-skippable read adds 1 (saturating at 255)
 * to the positions it spans on the reference (1 position for ops that do not consume it), maximal runs of
 * positions with evidence >= min_event_support become regions of interest -- cut at the read starts that
 * follow a coverage gap, where the reference flushes its cache -- and a region keeps the first
 * max_read_coverage + 1 non-skippable reads that overlap it; it is yielded when their number lies in
 * [min_read_coverage, max_read_coverage].  Positions are relative to `origin`; evidence outside
 * [origin, origin + span] is ignored (the reference indexes an array of target.length + 1 there).      */
typedef struct {
	int64_t n_reads;
	const int64_t *read_start, *read_stop;    /* Record.start, Record.stop; sorted by start            */
	const uint8_t *read_skip;                 /* skippable(r), indelope.nim:40-47                      */
	const int64_t *cigar_off;                 /* [n_reads + 1] into cigar                              */
	const uint32_t *cigar;                    /* BAM encoding: len << 4 | op, op in MIDNSHP=X          */
	int64_t origin, span;
	int32_t min_event_support;                /* gen_roi's min_event_support (uint8): 4, CLI max(3, min_reads - 2) */
	int32_t min_read_coverage;                /* 4 (CLI: min_reads)                                    */
	int32_t max_read_coverage;                /* 600                                                   */
} ihp_roi_in;

typedef struct {
	int64_t n_roi, n_read_idx;
	int64_t *roi_start, *roi_stop;            /* roi.start, roi.stop (inclusive), genomic              */
	int64_t *read_off;                        /* [n_roi + 1] into reads                                */
	int64_t *reads;                           /* indices of the region's reads in the input, BAM order */
} ihp_roi_out;

int  ihp_gen_roi(const ihp_roi_in *in, ihp_roi_out *out);
void ihp_free_roi(ihp_roi_out *out);

/* Host buffers in, host buffers out (upload + run + fetch).                   */
int  ihp_run_regions(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out);
void ihp_free_out(ihp_batch_out *out);

/* The same in three steps, so a caller (or bench.py) can keep a batch resident
 * in HBM: upload once, run any number of times, fetch results once.           */
typedef struct ihp_batch ihp_batch;     /* opaque device-resident batch        */
int  ihp_batch_upload(const ihp_params *p, const ihp_batch_in *in, ihp_batch **b);
int  ihp_batch_run(ihp_batch *b);                     /* async on the batch stream */
int  ihp_batch_sync(ihp_batch *b);                    /* waits; IHP_E_CAPACITY if a device pool overflowed in the run */
int  ihp_batch_fetch(ihp_batch *b, ihp_batch_out *out);
/* Contig c of `out` (either form) as ASCII bases and 32-bit supports: seq / sup hold ctg_seq_off[c+1] - ctg_seq_off[c]
 * entries each (either may be NULL).                                                                                  */
int  ihp_out_contig(const ihp_batch_out *out, int64_t c, uint8_t *seq, uint32_t *sup);
void ihp_batch_free(ihp_batch *b);
/* ihp_batch_upload for a slab (see ihp_slab_layout).  The copy is asynchronous on the batch's stream: the slab must stay
 * untouched until the first ihp_batch_sync / fetch of the batch has returned.                                         */
int  ihp_batch_upload_slab2(const ihp_params *p, int32_t n_regions, int64_t n_reads, const void *slab,
                            const ihp_slab2_layout *layout, int32_t flags, ihp_batch **b);   /* the compact slab (above); otherwise as ihp_batch_upload_slab */
int  ihp_batch_upload_slab(const ihp_params *p, int32_t n_regions, int64_t n_reads, const void *slab,
                           const ihp_slab_layout *layout, int32_t flags, ihp_batch **out);
/* What ihp_batch_fetch / ihp_batch_pack_dev bring back.  IHP_FETCH_NO_BASES: everything but the contigs' bases and
 * supports (n_bases = 0; ctg_seq_off still holds the lengths) -- events, k-mer counts, CIGARs, alignment records and the
 * contig directory, a tenth of the bytes; a later fetch with flags 0 brings the bases of the same run.               */
#define IHP_FETCH_NO_BASES 1
/* IHP_FETCH_EAGER: every ihp_batch_run from now on also counts what its results will take (two small kernels behind the
 * last stage, totals into page-locked memory), so that the fetch after it is one enqueue (compaction + copy) and one wait
 * instead of three round trips -- for callers that fetch every run (a sweep); leave it off for runs nobody fetches.      */
#define IHP_FETCH_EAGER 2
/* IHP_FETCH_COMPACT: ihp_batch_fetch brings the contigs' bases 4 bits each and their supports one byte each (see
 * ihp_batch_out: ctg_seq4 / ctg_sup8): the copy of a batch's full results is a third of what it was, and the host expands
 * only what it reads (ihp_out_contig).  A batch with a contig base outside BAM's 16-letter alphabet (lower case: only
 * possible when the reads came as ASCII arrays) is fetched in the plain form all the same.  ihp_batch_pack_dev ignores it.  */
#define IHP_FETCH_COMPACT 4
int  ihp_batch_set_fetch(ihp_batch *b, int32_t flags);
/* Hand the batch's scratch and result buffers back to the device pool; its inputs and the per-region summary
 * records (ihp_batch_summary_dev) stay.  Waits for the run and confirms it first (a run that left launches out is
 * repeated here when a region needed them, exactly as ihp_batch_sync would): the records that stay are final, and a
 * pool overflow of the run is reported as IHP_E_CAPACITY (the buffers are released either way).  For callers that walk through more regions than one GPU holds results for
 * (C4 on fewer than 8 GPUs): every chunk's inputs stay resident, one chunk's results at a time.  The next
 * ihp_batch_run takes buffers again; fetch / pack need that run first.                                           */
int  ihp_batch_release_outputs(ihp_batch *b);
/* The results of a batch compacted ON THE DEVICE into one slab that holds every array of ihp_batch_out (what
 * ihp_batch_fetch copies to the host in one piece): device pointer, size, and the six counts {regions, contigs,
 * bases, CIGAR words, events, hit entries} that define its layout.  Valid until the batch runs, packs or is freed
 * again.  This is the variable-length payload of the multi-GPU gather: each rank sends its slab over xGMI, the root
 * turns the copies back into ihp_batch_out views with ihp_unpack_slab (arrays point into the caller's buffer: do
 * not pass them to ihp_free_out).  ihp_pack_out builds the same slab from host results (buf NULL: size query).     */
int  ihp_batch_pack_dev(ihp_batch *b, void **dev_ptr, int64_t *bytes, int64_t counts[6]);
int  ihp_unpack_slab(void *slab, int64_t bytes, const int64_t counts[6], double error, ihp_batch_out *out);
int  ihp_pack_out(const ihp_batch_out *src, void *buf, int64_t cap, int64_t *bytes, int64_t counts[6]);
/* Per-stage device time of the most recent ihp_batch_run+sync, from HIP events
 * on the batch stream: ms[0] assemble, ms[1] ksw2, ms[2] tally, ms[3] total.  */
int  ihp_batch_stage_ms(ihp_batch *b, float ms[4]);
/* The same per stage as EXECUTION time -- from the moment the stage's first workgroup starts on the device to the moment
 * the next stage's does (the summary kernel's for the last one), read from the device's wall clock (ms[0] assembly incl.
 * the 2-bit packing and the overflow passes, [1] ksw2, [2] tally, [3] fallback) -- for callers that keep several batches
 * in flight: a kernel of one batch can wait for wave slots held by another's, and an event interval includes that
 * wait.  One clock read per stage by the kernels themselves; off by default.                                          */
int  ihp_batch_set_timing(ihp_batch *b, int on);
int  ihp_batch_kernel_ms(ihp_batch *b, float ms[4]);
/* The same four times as a mean over the runs waited for with ihp_batch_sync since timing was switched on (or since the
 * last call with reset != 0): the stamps arrive in page-locked host memory, ihp_batch_sync adds them up without a HIP
 * call, so a timed loop needs no read-out inside it.  n_runs (may be NULL) = runs in the mean.                       */
int  ihp_batch_kernel_ms_mean(ihp_batch *b, float ms[4], int64_t *n_runs, int reset);
/* Device time of the alignment-fallback kernel (indelope.nim:312-372) in the same run; it is
 * included in ms[3] and runs between the tally and the summary.                               */
int  ihp_batch_fallback_ms(ihp_batch *b, float *ms);
/* Diagnostics: after ihp_debug_set("profile", 1) the kernels sum shader-clock cycles
 * per phase over all waves: [0] assemble, [1] combine, [2] assemble+output, [3] regions;
 * [8] ksw2 init, [9] ksw2 DP, [10] ksw2 traceback, [11] alignments; packed read phase: [12] read preparation,
 * [13] target-offset filter, [14] query-offset phase, [15] insert, [27] set-up.  Always filled: [24]/[25]/[26] regions
 * forwarded at run time (arena / slot overflow) to the 2nd/3rd/4th assembly pass, [23] regions the packed pass handed
 * back to the byte-based class-1 kernel, [28] regions sent on to the roomy combine launch, [29] regions the read phase
 * filed under the second (larger-arena) combine launch, [30] under the third, [21] 1 if the last run left the retry
 * launches out (bit 0 the assembly retry launches, 1 the roomy ksw2 launch, 2 the roomy launch of the alignment fallback), [31] runs of this batch repeated in full because of that, [47] items of the alignment fallback that took its roomy launch, [22] ksw2 kernel mode; [32..39] event counts of the combine kernel
 * (best_match calls, exact candidates, verification passes, vote scans, merges, filter passes, query-phase target looks, trims that read supports). */
int  ihp_batch_profile(ihp_batch *b, int64_t out[64]);
/* The same with the caller's capacity: the first min(cap, 64) counters.  (ihp_batch_profile writes 64 int64 since round 3;
 * a caller with a smaller buffer uses this one.)                                                                   */
int  ihp_batch_profile_n(ihp_batch *b, int64_t *out, int32_t cap);
/* Diagnostics: which ksw2 kernel the most recent ksw_extz2_sse / ihp_ksw_extz2_batch /
 * ihp_batch_run used: 3/4 = top-byte register sweep (left/right gaps; the production
 * kernels), 0/1 = masked register sweep (scoring schemes or base codes the former does
 * not cover), 5 = ring sweep for bands > 62 or unbanded (ksw_wide.h; per job the LDS sweep
 * where the ring does not fit), 2 = LDS sweep.  Results are identical in every mode.       */
int  ihp_debug_last_ksw_mode(void);
/* How many PAIRS of alignments the last ihp_ksw_extz2_batch call ran two to a wavefront (ksw_pair.h; 0 when the
 * parameters or the jobs did not allow it, or after ihp_debug_set("ksw_pair", 0)).                             */
int  ihp_debug_last_ksw_pairs(void);
/* Diagnostics: n (read, target 0, target 1) items through the two-target sweep the alignment fallback runs per (event, read)
 * (ksw_duo.h; indelope.nim:336-344 calls align_to twice, ksw2.nim:151-164, on the same read) -- on caller-made strings, so that
 * a test can hold the sweep against the reference's ksw_extz2_sse (ksw2.h:54) directly.  Strings are base codes 0..4, q_off /
 * t0_off / t1_off have n + 1 entries.  ez[2 i], ez[2 i + 1]: max, max_q, max_t and n_cigar of item i's two alignments (the
 * fields the fallback reads, indelope.nim:343-344; the others stay reset); n_cigar = -2: the sweep does not take such an item
 * (banded, z-drop, a read over 320 bases ...: k_fallback runs those one at a time).  cigar: 2 n slots of cig_slot words.      */
int  ihp_debug_ksw_duo_batch(int32_t n, const uint8_t *reads, const int64_t *q_off, const uint8_t *t0, const int64_t *t0_off,
                             const uint8_t *t1, const int64_t *t1_off, int8_t m, const int8_t *mat, int8_t q, int8_t e,
                             int w, int zdrop, int flag, ihp_ez *ez, uint32_t *cigar, int32_t cig_slot);
/* Test hook: upper limits for the device pools of batches uploaded from now on -- {CIGAR bump-pool words,
 * event-pool entries, hit-pool ints, ksw2 traceback bytes per wave}; 0 = the library's own sizing.  NULL resets.
 * Lets the overflow paths (IHP_E_CAPACITY from ihp_batch_sync / fetch) be driven by small inputs.               */
int  ihp_debug_limits(const int64_t limits[4]);
/* Test / diagnostics switches for batches uploaded (and runs started) from now on; results never depend on them.
 *   "asm_v1" 1      class-1 regions through the byte-based k_assemble passes only (no packed assembly)
 *   "no_hint" 1     every combine launch with its full grid (default: a launch the previous batch left empty gets a token grid)
 *   "no_spec" 1     the retry launches (roomy combine, byte-based overflow passes) are always enqueued (default: left out of a
 *                   run when the previous batch needed none of them; whoever waits for the run checks, and repeats it in full)
 *   "spec_fail" 1   test hook: such a run is treated as if a region had needed them
 *   "verbose" 1     a line on stderr per run: the combine tiers (waves per CU, arenas, grids) it was launched with
 *   "ksw_p_cap" n   bytes of traceback scratch per wave of the main ksw2 launch (0 = library sizing): jobs that need more take the
 *                   roomy launch behind it
 *   "fb_p_cap" n    the same for the main launch of the alignment fallback: items that need more take ITS roomy launch
 *   "comb_waves" n  waves per workgroup of the combine kernel, 1 / 2 / 4 (0 = by the tier's occupancy): wave 0 runs the region,
 *                   the others take shares of its best_match calls
 *   "no_rich" 1     read-rich regions (assembly classes 2-4) stay with the byte-based passes instead of the packed path
 *   "tally_pk" 0    k_tally on the ASCII bases even when the 2-bit reads are at hand
 *   "lpt" 0         k_asm_combine3 in input order: no cost classes, no arena tiers
 *   "prepack_fast" n  0: the plain k_prepack for every batch; 1..4: reads per 16-lane group the pipelined k_prepack_fast keeps in
 *                   flight (default 2; taken when the bases are ASCII and the batch brought its trim bounds)
 *   "fb_duo" 0      the alignment fallback runs the two alignments of an item one after the other (ksw_wide.h) instead of in one
 *                   sweep (ksw_duo.h); same votes
 *   "ksw_pair" 0    every ksw2 alignment through the single sweep (default: jobs of equal contig length share a wavefront where
 *                   their windows allow it, ksw_pair.h)
 *   "asm_waves", "asmr_waves", "comb_occ", "ksw_waves", "tally_waves"   waves per CU of a kernel (0 = library sizing)
 *   "v2_arena", "v2_pdw"   LDS bytes / dwords per wave of the packed assembly (0 = library sizing)
 *   "profile" 1     per-phase cycle counters (ihp_batch_profile)
 *   "strict_ksw" 1  ksw_extz2_sse aborts when it fails (same as IHP_KSW_STRICT=1 in the environment)
 * key == NULL resets everything.  Returns IHP_E_ARG for an unknown key.  These replace the IHP_* environment
 * variables of earlier rounds; the library reads no environment variable besides IHP_KSW_STRICT.                  */
int  ihp_debug_set(const char *key, int64_t value);
/* Fixed-size per-region summary record left on the device for the multi-GPU
 * gather (one RCCL gather of these at the end).     */
typedef struct {
	int32_t status, n_contigs_pre, n_contigs, n_aligned, n_events, n_tallied;
	int32_t ref_support, alt_support;     /* of the first tallied event, else -1 */
} ihp_region_summary;
/* Device pointer (valid until the next run/free) + count of the summaries.  The call waits for the batch's run and
 * confirms it (a run may be repeated at that point, see "no_spec"): the records behind the pointer are final.         */
int  ihp_batch_summary_dev(ihp_batch *b, void **dev_ptr, int64_t *n);
/* The same pointer WITHOUT the wait: the address is fixed from upload to ihp_batch_free, so a caller that orders its own work
 * by stream (an RCCL gather enqueued on another stream behind an event, ...) takes it once and never stalls the host here.  The
 * records behind it are final only after a call that waits for and confirms the run (ihp_batch_sync, fetch, summary_dev).      */
int  ihp_batch_summary_ptr(ihp_batch *b, void **dev_ptr, int64_t *n);
/* The same records copied to the host (cap >= n_regions entries).             */
int  ihp_batch_summary_host(ihp_batch *b, ihp_region_summary *out, int64_t cap);


/* ------------------------------------------------- multi-GPU: one process per GPU, ONE gather at the end
 * Regions are independent past indelope.nim:601-603, so every rank runs a contiguous range of them through the entry points
 * above with no collective on the data path.  What the reference's main loop needs afterwards is every region's records in
 * REGION ORDER ON ONE RANK: its last-two-variants dedupe (indelope.nim:604-608) is sequential.  These entry points are that
 * one exchange, over librccl directly (ncclCommInitRank; grouped ncclSend / ncclRecv to the root -- every peer over its own
 * xGMI link, no ring: a ring gather is bound by one link, the root's seven links take seven peers at once).  librccl.so.1 is
 * opened with dlopen by the first ihp_dist_* call: a single-GPU caller never loads it, and a box without it gets
 * IHP_E_UNSUPPORTED (text in ihp_last_hip_error), not a failure to load this library.
 *
 *   rank 0:        ihp_dist_unique_id(id)         -> the caller carries the 128 bytes to the other ranks (file, pipe, MPI, env ...)
 *   every rank:    ihp_init(local_gpu); ihp_dist_init(rank, world, id, &d)
 *                  ... ihp_batch_upload / ihp_batch_run over the rank's regions ...
 *                  ihp_dist_gather_summaries(d, b, 0, ...)        the 32-byte records, rank (= region) order, on the root's host
 *                  ihp_dist_gather_payload(d, b, 0, outs)         the full results: one ihp_batch_out per rank on the root
 *                  ihp_dist_finalize(d)
 * Calls on one communicator are collective: every rank makes the same calls in the same order.  One host thread per
 * communicator at a time.                                                                                                     */
typedef struct ihp_dist ihp_dist;
#define IHP_DIST_ID_BYTES 128
/* ncclGetUniqueId (no GPU needed yet).  cap >= IHP_DIST_ID_BYTES.                                                             */
int  ihp_dist_unique_id(void *id, int64_t cap);
/* ncclCommInitRank on the library's device (ihp_init first: rank r of a node binds GPU r).  Blocks until all `world` ranks
 * have arrived.                                                                                                               */
int  ihp_dist_init(int32_t rank, int32_t world, const void *id, int64_t id_bytes, ihp_dist **out);
int  ihp_dist_rank(const ihp_dist *d);
int  ihp_dist_world(const ihp_dist *d);
/* n records of 32 bytes in device memory (final: the caller has waited for the runs that wrote them -- ihp_batch_sync) from
 * every rank to `root`.  counts_in: records per rank [world] when every rank knows them (contiguous shards: it does), or NULL:
 * they are exchanged first (one ncclAllGather of an int64 each).  On the root: out (host, cap records) gets the records in
 * rank order, *n_total their number, counts_out [world] (optional) the per-rank counts; the other ranks pass what they like
 * there (ignored).  IHP_E_CAPACITY when cap is short (after the exchange: the communicator stays usable).                     */
int  ihp_dist_gather_records(ihp_dist *d, const void *dev_records, int64_t n, int32_t root, const int64_t *counts_in,
                             ihp_region_summary *out, int64_t cap, int64_t *n_total, int64_t *counts_out);
/* The same for a batch: waits for and confirms b's run (as ihp_batch_summary_dev does), then gathers its n_regions records.   */
int  ihp_dist_gather_summaries(ihp_dist *d, ihp_batch *b, int32_t root, const int64_t *counts_in,
                               ihp_region_summary *out, int64_t cap, int64_t *n_total, int64_t *counts_out);
/* The variable-length half: every rank compacts b's results on its device (ihp_batch_pack_dev), the sizes go round in one
 * ncclAllGather of seven int64, the slabs travel to the root point to point -- all receives posted before any is waited for --
 * and arrive as outs[0 .. world) on the root (rank order = region order; each is released with ihp_free_out; genotypes filled
 * in as by ihp_batch_fetch).  Other ranks: outs is ignored.  bytes_out (optional, root): bytes received per rank.               */
int  ihp_dist_gather_payload(ihp_dist *d, ihp_batch *b, int32_t root, ihp_batch_out *outs, int64_t *bytes_out);
/* ncclCommDestroy and the communicator's buffers.  NULL is fine.                                                              */
int  ihp_dist_finalize(ihp_dist *d);

#ifdef __cplusplus
}
#endif
#endif /* INDELOPE_HIP_H_ */
