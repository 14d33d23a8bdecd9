/* oracle_internal.h -- TEST INFRASTRUCTURE ONLY (see oracle.h). */
#ifndef INDELOPE_ORACLE_INTERNAL_H_
#define INDELOPE_ORACLE_INTERNAL_H_
#include "oracle.h"

typedef struct { ihp_correction *v; int64_t n, cap; } orc_corrvec;
typedef struct { int64_t matches, offset, mismatches, contig_i; orc_corrvec corr; } orc_match;
typedef struct { ihp_contig **v; int64_t n, cap; } orc_list;

extern _Thread_local int64_t orc_cnt_compares, orc_cnt_cells, orc_cnt_kmers;

int  orc_allowed(int rule, uint32_t qsup, uint32_t tsup, int64_t qreads, int64_t treads);
void orc_corr_push(orc_corrvec *v, int64_t qoff, int64_t toff, int qbest);
void orc_slide_core(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                    int64_t max_mismatch, int rule, orc_match *out);
int  orc_insert_core(ihp_contig *t, ihp_contig *q, const orc_match *m, int grow);
void orc_trim_core(ihp_contig *c, int64_t min_support);
void orc_list_push(orc_list *l, ihp_contig *c);
ihp_contig *orc_make_contig(const uint8_t *dna, int64_t n, int64_t start, uint32_t support);
void orc_contig_free(ihp_contig *c);
int  orc_best_match(orc_list *contigs, const ihp_contig *q, int64_t min_overlap,
                    int64_t max_mismatch, orc_match *best);
void orc_list_insert(orc_list *contigs, ihp_contig *q, int64_t min_overlap, int64_t max_mismatch);
orc_list orc_combine(orc_list contigs, int64_t max_mismatch, int64_t min_support, int again,
                     int64_t combine_min_overlap);
void orc_ksw_dispatch(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                      int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                      ksw_extz_t *ez);
#endif
