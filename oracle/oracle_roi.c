/*
 * oracle_roi.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the ROI evidence scan: event_locations (src/indelope.nim:430-445), gen_roi_internal
 * (:461-499) and gen_roi (:515-545), kept in the reference's own shape -- one pass over the reads with a
 * cache that is flushed (and its evidence window scanned) at every coverage gap -- so that the device
 * implementation, which works on the whole run of reads at once, is checked against the sequential
 * semantics and not against itself.  PARITY UNPINNED: no reference test covers gen_roi and the Nim
 * program cannot be built here.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

typedef struct {
	int64_t n_roi, cap_roi, n_idx, cap_idx;
	int64_t *start, *stop, *off, *idx;
} roi_acc;

static void acc_roi(roi_acc *a, int64_t s, int64_t e, const int64_t *reads, int64_t n)
{
	if (a->n_roi + 2 > a->cap_roi) {
		a->cap_roi = a->cap_roi * 2 + 64;
		a->start = (int64_t *)realloc(a->start, sizeof(int64_t) * (size_t)a->cap_roi);
		a->stop = (int64_t *)realloc(a->stop, sizeof(int64_t) * (size_t)a->cap_roi);
		a->off = (int64_t *)realloc(a->off, sizeof(int64_t) * (size_t)(a->cap_roi + 1));
	}
	if (a->n_idx + n > a->cap_idx) {
		a->cap_idx = (a->n_idx + n) * 2 + 64;
		a->idx = (int64_t *)realloc(a->idx, sizeof(int64_t) * (size_t)a->cap_idx);
	}
	a->start[a->n_roi] = s; a->stop[a->n_roi] = e; a->off[a->n_roi] = a->n_idx;
	memcpy(a->idx + a->n_idx, reads, sizeof(int64_t) * (size_t)n);
	a->n_idx += n; a->n_roi++;
	a->off[a->n_roi] = a->n_idx;
}

/* gen_roi_internal, :461-499: evidence window [cache_start, cache_end), `cache` = indices of the cached reads */
static void roi_internal(const ihp_roi_in *in, const uint8_t *evidence, const int64_t *cache, int64_t n_cache,
                         int64_t cache_start, int64_t cache_end, roi_acc *acc, int64_t *reads)
{
	const uint8_t min_evidence = (uint8_t)in->min_event_support;
	int in_roi = 0;
	int64_t roi_start = 0, roi_end = 0;
	for (int64_t i = cache_start; i <= cache_end; ++i) {
		if (i < cache_end && evidence[i] >= min_evidence) {      /* :470-475 */
			if (!in_roi) { in_roi = 1; roi_start = i; }
			roi_end = i;
			continue;
		}
		if (!in_roi) continue;                                   /* :478 / :489: the same block closes a region inside and at the end of the window */
		int64_t n = 0;
		for (int64_t k = 0; k < n_cache; ++k) {
			const int64_t r = cache[k];
			const int64_t rs = in->read_start[r] - in->origin, re = in->read_stop[r] - in->origin;
			if (!(rs > roi_end) && !(re < roi_start)) {          /* overlaps, :447-450 */
				reads[n++] = r;
				if (n > in->max_read_coverage) break;            /* :483 */
			}
			if (rs > roi_end) break;                             /* :484 */
		}
		if (n >= in->min_read_coverage && n <= in->max_read_coverage)   /* :485 */
			acc_roi(acc, roi_start + in->origin, roi_end + in->origin, reads, n);
		in_roi = 0;
	}
}

int orc_gen_roi(const ihp_roi_in *in, ihp_roi_out *out)
{
	if (!in || !out || in->n_reads < 0 || in->span < 0) return IHP_E_ARG;
	memset(out, 0, sizeof(*out));
	const int64_t len = in->span + 1;                            /* new_seq[uint8](t.length + 1), :522 */
	uint8_t *evidence = (uint8_t *)calloc((size_t)len, 1);
	int64_t *cache = (int64_t *)malloc(sizeof(int64_t) * (size_t)(in->n_reads ? in->n_reads : 1));
	int64_t *reads = (int64_t *)malloc(sizeof(int64_t) * (size_t)(in->max_read_coverage + 2));
	roi_acc acc; memset(&acc, 0, sizeof(acc));
	acc.off = (int64_t *)calloc(1, sizeof(int64_t)); acc.cap_roi = 0;
	int64_t n_cache = 0, cache_stop = 0, last_start = 0;
	for (int64_t r = 0; r < in->n_reads; ++r) {
		const int64_t rs = in->read_start[r] - in->origin, re = in->read_stop[r] - in->origin;
		if (n_cache > 0 && rs > cache_stop) {                    /* :529-535 */
			int64_t end = rs < len ? rs : len;
			roi_internal(in, evidence, cache, n_cache, last_start < 0 ? 0 : last_start, end < 0 ? 0 : end, &acc, reads);
			last_start = rs;
			n_cache = 0; cache_stop = 0;
		}
		if (in->read_skip && in->read_skip[r]) continue;         /* :537 */
		cache[n_cache++] = r;                                    /* cache.add, :505-507 */
		if (re > cache_stop) cache_stop = re;
		int64_t off = 0;                                         /* event_locations, :430-445 */
		for (int64_t c = in->cigar_off[r]; c < in->cigar_off[r + 1]; ++c) {
			const uint32_t op = in->cigar[c] & 0xf; const int64_t clen = (int64_t)(in->cigar[c] >> 4);
			const int cons = op == 0 || op == 2 || op == 3 || op == 7 || op == 8;   /* consumes.reference */
			if (op != 0) {
				const int64_t es = rs + off, ee = cons ? rs + off + clen : rs + off + 1;
				for (int64_t i = es; i < ee; ++i) {
					if (i < 0 || i >= len) continue;
					evidence[i] += 1;                            /* :540-543 */
					if (evidence[i] == 0) evidence[i] = 255;
				}
			}
			if (cons) off += clen;
		}
	}
	roi_internal(in, evidence, cache, n_cache, last_start < 0 ? 0 : (last_start > len ? len : last_start), len, &acc, reads);   /* :544 */
	free(evidence); free(cache); free(reads);
	out->n_roi = acc.n_roi; out->n_read_idx = acc.n_idx;
	out->roi_start = acc.start ? acc.start : (int64_t *)calloc(1, 8);
	out->roi_stop = acc.stop ? acc.stop : (int64_t *)calloc(1, 8);
	out->read_off = acc.off;
	out->reads = acc.idx ? acc.idx : (int64_t *)calloc(1, 8);
	return 0;
}

void orc_free_roi(ihp_roi_out *out)
{
	if (!out) return;
	free(out->roi_start); free(out->roi_stop); free(out->read_off); free(out->reads);
	memset(out, 0, sizeof(*out));
}
