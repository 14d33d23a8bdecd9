/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of indelope's per-region hot path, used as the
 * parity checker for the HIP implementation.  Only tests/, the smoke() entry
 * and bench.py's cpu_baseline leg may load this library; nothing under
 * indelope_amd/ links, imports or calls it.
 *
 * The entry points mirror include/indelope_hip.h one for one (orc_ prefix) so
 * that a parity test is "call both, compare the buffers".
 *
 * Pinning: the ksw2 part is checked against the reference's own C file compiled
 * into oracle/_ref (see Makefile) and against the KAT of ksw2.nim:171-214; the
 * contig part against every vector of contig.nim:292-430; genotype against
 * genotyper.nim:49-67.  NOT pinned by any reference test or runnable reference:
 * multi-read assemble/combine, Ez.cigar truncation at production flags, k-mer
 * selection, and the k-mer tally (third-party Nim package `kmer`, un-vendored,
 * no version pinned: indelope.nimble:10-11) -- "parity unpinned" for those; the
 * restatement follows the Nim source line by line instead.
 */
#ifndef INDELOPE_ORACLE_H_
#define INDELOPE_ORACLE_H_

#include "../include/indelope_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ksw2_extz2_sse.c:113-388 restated lane by lane in scalar C.
 * variant 0 = the SSE2 code path (what Nim's default gcc flags compile),
 * variant 1 = the SSE4.1 code path (-msse4.1).                                */
void orc_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                   ksw_extz_t *ez);
void orc_ksw_set_variant(int variant);
/* Optional: route the batch path's alignments through another implementation of
 * the same signature (bench.py plugs in oracle/_ref's compiled reference C so
 * the CPU baseline times the real SSE code).  NULL restores the restatement.  */
typedef void (*orc_ksw_fn)(void *, int, const uint8_t *, int, const uint8_t *, int8_t,
                           const int8_t *, int8_t, int8_t, int, int, int, ksw_extz_t *);
void orc_set_ksw_impl(orc_ksw_fn fn);

int orc_ksw_extz2_batch(int32_t n, const uint8_t *queries, const int64_t *q_off,
                        const uint8_t *targets, const int64_t *t_off,
                        int8_t m, const int8_t *mat, int8_t q, int8_t e,
                        int w, int zdrop, int flag,
                        ihp_ez *ez, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_off);

void orc_encode(const uint8_t *dna, int64_t n, uint8_t *out);
void orc_matrix(int8_t match, int8_t mismatch, int8_t out25[25]);

int orc_slide_align(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                    int64_t max_mismatch, int allow_rule, ihp_match *out);
int orc_contig_insert(ihp_contig *t, ihp_contig *q, const ihp_match *m);
int orc_contig_trim(ihp_contig *c, int64_t min_support);
/* trim(sequence, base_qualities, min_quality) of indelope.nim:23-38: returns a,
 * writes the kept range [*lo, *hi) (empty when the read is emptied).          */
int64_t orc_read_trim(const uint8_t *quals, int64_t n, int min_quality, int64_t *lo, int64_t *hi);

int orc_kmer_tally(int32_t n_reads, const uint8_t *bases, const int64_t *read_off,
                   const uint8_t *mapq, int32_t min_mapq, int32_t K,
                   const char *ref_kmer, const char *alt_kmer, int32_t counts[3]);

int    orc_genotype(int64_t r, int64_t a, double error, ihp_genotype_t *out);
double orc_genotype_qual(const ihp_genotype_t *g);

void orc_params_default(ihp_params *p);
/* nthreads > 1 splits the regions over pthreads (regions are independent).    */
int  orc_run_regions(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out);
int  orc_run_regions_mt(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out, int nthreads);
void orc_free_out(ihp_batch_out *out);
/* indelope.nim:375-428 + :604-608 + `$`(Variant): mirrors of ihp_call_variants & co. (oracle_variants.c) */
int  orc_call_variants(const ihp_params *p, const ihp_batch_in *in, const ihp_batch_out *out, ihp_variants *vars);
void orc_free_variants(ihp_variants *vars);
/* gen_roi (indelope.nim:430-545): mirror of ihp_gen_roi (oracle_roi.c) */
int  orc_gen_roi(const ihp_roi_in *in, ihp_roi_out *out);
void orc_free_roi(ihp_roi_out *out);
int64_t orc_format_variant(const ihp_variant *v, const char *chars, const char *chrom, char *buf, int64_t cap);
/* cpu_baseline probe: each thread runs its share of the regions `reps` times, results discarded. */
int  orc_bench_regions(const ihp_params *p, const ihp_batch_in *in, int nthreads, int reps);

/* Deterministic work counters of the last orc_run_regions call (single thread):
 * [0] char compares in slide_align, [1] ksw2 DP cells, [2] k-mer steps.       */
void orc_counters(int64_t c[3]);

#ifdef __cplusplus
}
#endif
#endif
