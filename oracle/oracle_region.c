/*
 * oracle_region.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of the per-region caller: assemble (src/indelope.nim:157-183),
 * callsemble up to and including the k-mer tally (:201-311), the CIGAR
 * truncation and event iterators of src/ksw2/ksw2.nim:22-33,71-91, the
 * canonical k-mer arithmetic of the third-party `kmer` package (indelope.nimble:
 * 10-11; un-vendored, no version pinned => PARITY UNPINNED for the tally: the
 * counts are implementation independent on upper-case ACGT input, which is all
 * this file claims), and genotype (src/genotyper.nim:22-47).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "oracle_internal.h"

_Thread_local int64_t orc_cnt_kmers;
static int64_t g_counters[3];

void orc_counters(int64_t c[3]) { c[0] = g_counters[0]; c[1] = g_counters[1]; c[2] = g_counters[2]; }

void orc_params_default(ihp_params *p)
{
	memset(p, 0, sizeof(*p));
	p->struct_size = (int32_t)sizeof(*p);
	p->min_overlap_pct = 0.88; p->min_mapq_assemble = 20; p->min_mapq_stop = 5; p->min_mapq_tally = 10;
	p->trim_min_qual = 15; p->combine_min_support = 3; p->combine_min_overlap = 65; p->max_mismatch = 0;
	p->max_pre_contigs = 20; p->min_ctg_len = 74; p->min_reads = 4; p->min_event_len = 4;
	p->K = 27; p->max_events = 4; p->ref_pad = 50;
	p->match = 1; p->mismatch = -2; p->gap_open = 4; p->gap_ext = 1;
	p->bw = 50; p->zdrop = 400; p->ksw_flag = 0;
	p->error = 1e-3;
	p->fallback = 1;
	p->fb_match = 1; p->fb_mismatch = -2; p->fb_gap_open = 5; p->fb_gap_ext = 1;   /* indelope.nim:318-319 */
	p->fb_bw = -1; p->fb_zdrop = -1; p->fb_flag = 0;                                /* ksw2.nim:159 */
}

/* ---- k-mers (G1, G2) ------------------------------------------------------ */
static int base2(uint8_t c)
{
	switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; }
	return -1;
}

/* canonical code = min(forward, reverse complement), 2 bits per base.  Returns
 * 0 and leaves *code alone if a byte is not upper-case ACGT. */
static int mincode(const char *kmer, int K, uint64_t *code)
{
	uint64_t f = 0, rc = 0;
	for (int i = 0; i < K; ++i) {
		int b = base2((uint8_t)kmer[i]);
		if (b < 0) return 0;
		f = (f << 2) | (uint64_t)b;
		rc |= (uint64_t)(3 - b) << (2 * i);
	}
	*code = f < rc ? f : rc;
	return 1;
}

/* indelope.nim:293-311 for one read: rolling canonical code over every k-mer */
static void tally_read(const uint8_t *seq, int64_t n, int K, uint64_t refe, uint64_t alte,
                       int *ref_found, int *alt_found, int32_t *ref_pos, int32_t *alt_pos)
{
	const uint64_t mask = K < 32 ? (((uint64_t)1 << (2 * K)) - 1) : ~(uint64_t)0;
	uint64_t f = 0, rc = 0;
	int valid = 0;
	*ref_found = *alt_found = 0;
	*ref_pos = *alt_pos = -1;
	for (int64_t i = 0; i < n; ++i) {
		int b = base2(seq[i]);
		if (b < 0) { valid = 0; f = rc = 0; continue; }
		f = ((f << 2) | (uint64_t)b) & mask;
		rc = (rc >> 2) | ((uint64_t)(3 - b) << (2 * (K - 1)));
		if (++valid < K) continue;
		orc_cnt_kmers++;
		uint64_t e = f < rc ? f : rc;
		if (!*ref_found && e == refe) { *ref_found = 1; *ref_pos = (int32_t)(i - K + 1); }   /* :301-304 */
		if (!*alt_found && e == alte) { *alt_found = 1; *alt_pos = (int32_t)(i - K + 1); }   /* :306-309 */
	}
}

int orc_kmer_tally(int32_t n_reads, const uint8_t *bases, const int64_t *read_off,
                   const uint8_t *mapq, int32_t min_mapq, int32_t K,
                   const char *ref_kmer, const char *alt_kmer, int32_t counts[3])
{
	uint64_t refe, alte;
	if (K < 1 || K > 31) return IHP_E_ARG;
	counts[0] = counts[1] = counts[2] = 0;
	if (!mincode(ref_kmer, K, &refe) || !mincode(alt_kmer, K, &alte)) return IHP_E_UNSUPPORTED;
	for (int32_t i = 0; i < n_reads; ++i) {
		if (mapq && mapq[i] < min_mapq) continue;    /* :294 */
		int rf, af; int32_t rp, apos;
		tally_read(bases + read_off[i], read_off[i + 1] - read_off[i], K, refe, alte, &rf, &af, &rp, &apos);
		counts[0] += rf; counts[1] += af;
		if (rf && af) counts[2] += 1;
	}
	return 0;
}

/* ---- genotyper.nim --------------------------------------------------------- */
int orc_genotype(int64_t r, int64_t a, double error, ihp_genotype_t *out)
{                                                    /* :36-47 */
	const double log2_ = log(2.0);
	double total = (double)(r + a);
	out->gt = IHP_GT_HOM_REF; out->_pad = 0;
	out->gl[0] = out->gl[1] = out->gl[2] = 0.0;
	if (total == 0) { out->gt = IHP_GT_UNKNOWN; return 0; }
	for (int G = 0; G <= 2; ++G) {
		double g = (double)G, h = (double)(2 - G);
		out->gl[G] = -total * log2_ + (double)r * log(g * error + h * (1 - error))
		             + (double)a * log(g * (1 - error) + h * error);
		if (out->gl[G] > out->gl[out->gt]) out->gt = G;
	}
	return 0;
}

double orc_genotype_qual(const ihp_genotype_t *g)
{                                                    /* :22-29 */
	if (g->gt == IHP_GT_HOM_REF) return g->gl[0] - fmax(g->gl[1], g->gl[2]);
	if (g->gt == IHP_GT_HET) return g->gl[1] - fmax(g->gl[0], g->gl[2]);
	if (g->gt == IHP_GT_HOM_ALT) return g->gl[2] - fmax(g->gl[0], g->gl[1]);
	return 0;
}

/* ---- one region ------------------------------------------------------------ */
typedef struct {
	int32_t flags; int64_t ref_start; int32_t ref_len;
	ihp_ez ez; uint32_t *cigar; int32_t n_events; ihp_event *events;
	int32_t **hits;              /* per event: NULL, or [2 * n_reads of the region]: ref then alt first-hit positions */
} ctg_res;

typedef struct {
	int32_t status, n_pre;
	orc_list contigs;
	ctg_res *res;
} region_res;

static int distinct_bytes(const char *s, int n)
{
	int seen[256] = {0}, d = 0;
	for (int i = 0; i < n; ++i) if (!seen[(uint8_t)s[i]]++) d++;
	return d;
}

/* trim() done by the stager (ihp_batch_in.trim_lo/trim_hi), clamped to the read */
static void trim_given(const ihp_batch_in *in, int64_t i, int64_t n, int64_t *lo, int64_t *hi)
{
	int64_t l = in->trim_lo[i], h = in->trim_hi[i];
	if (l < 0) l = 0;
	if (l > n) l = n;
	if (h > n) h = n;
	if (h < l) h = l;
	*lo = l; *hi = h;
}

/* count_flanked_cigar, indelope.nim:185-199, over Ez.cigar (ksw2.nim:22-33: the CIGAR
 * truncated at max_q) */
static int count_flanked_cigar(const ksw_extz_t *ez)
{
	int matched = 0, n = 0, last_op = 0;
	uint32_t max_off = (uint32_t)ez->max_q, off = 0;
	for (int i = 0; i < ez->n_cigar; ++i) {
		if (off >= max_off) break;
		uint32_t op = ez->cigar[i] & 0xf, len = ez->cigar[i] >> 4;
		if (op != 2) off += len;
		if (!matched) { if (op == 0) { n += 1; matched = 1; } }
		else n += 1;
		last_op = (int)op;
	}
	if (last_op != 0) n -= 1;
	return n;
}

/* The alignment fallback of indelope.nim:312-372 for one event: every read with mapq >= 10
 * is quality-trimmed and aligned (gap open 5, unbanded, no z-drop) to the reference window
 * and to the contig, both cut at the read's start; a read votes for the side whose
 * alignment is a single M run while the other side needs more ops.
 * Degenerate cases the Nim code would trap on (an emptied read: `query[0].addr` on an empty
 * seq, ksw2.nim:155; a start beyond the contig: negative slice) are aligned as empty strings,
 * i.e. ksw returns with n_cigar = 0 (ksw2_extz2_sse.c:146-147) and the read cannot vote.   */
static void fallback_align(const ihp_params *p, const ihp_batch_in *in, int64_t r0, int64_t r1,
                           const ihp_contig *ctg, const uint8_t *reference, int64_t reflen,
                           ihp_event *ev, ksw_extz_t *ez_ref, ksw_extz_t *ez_alt)
{
	int8_t mat[25];
	orc_matrix(p->fb_match, p->fb_mismatch, mat);
	int8_t gapo = p->fb_gap_open < 0 ? -p->fb_gap_open : p->fb_gap_open;
	int8_t gape = p->fb_gap_ext < 0 ? -p->fb_gap_ext : p->fb_gap_ext;
	uint8_t *renc = (uint8_t *)malloc((size_t)reflen + 1), *cenc = (uint8_t *)malloc((size_t)ctg->len + 1);
	orc_encode(reference, reflen, renc);
	orc_encode(ctg->sequence, ctg->len, cenc);
	uint8_t *qenc = 0;
	int ref_support = 0, alt_support = 0;
	for (int64_t ri = r0; ri < r1; ++ri) {
		if (in->mapq[ri] < p->min_mapq_tally) continue;             /* :325 */
		const int64_t n = in->read_off[ri + 1] - in->read_off[ri];
		int64_t lo = 0, hi = n, a = 0;
		if (in->trim_lo) { trim_given(in, ri, n, &lo, &hi); a = lo; }
		else if (in->quals) a = orc_read_trim(in->quals + in->read_off[ri], n, p->trim_min_qual, &lo, &hi);
		else if (n == 1) hi = 0;                                    /* quals NULL = all 255: trim() empties a 1-base read (:28-30) */
		const int64_t rs = in->read_start[ri] + a, rl = hi - lo;    /* :328 */
		if (rs > ev->tstop) continue;                               /* :329 */
		const int64_t L = ev->type == 0 ? (int64_t)ev->len : 0;     /* :330-332 */
		if (rs + rl + L < ev->tstart) continue;                     /* :333 */
		int64_t start = (rs > ctg->start ? rs : ctg->start) - ctg->start;   /* :336 */
		int64_t rsub = reflen - start, csub = ctg->len - start;     /* :337-338 */
		if (rsub < 0) rsub = 0;
		if (csub < 0) csub = 0;
		qenc = (uint8_t *)realloc(qenc, (size_t)rl + 1);
		orc_encode(in->bases + in->read_off[ri] + lo, rl, qenc);
		ez_ref->n_cigar = 0; ez_alt->n_cigar = 0;                   /* ksw2.nim:153 */
		orc_ksw_dispatch((int)rl, qenc, (int)rsub, renc + (rsub ? start : 0), 5, mat, gapo, gape,
		                 p->fb_bw, p->fb_zdrop, p->fb_flag, ez_ref);   /* :340 */
		orc_ksw_dispatch((int)rl, qenc, (int)csub, cenc + (csub ? start : 0), 5, mat, gapo, gape,
		                 p->fb_bw, p->fb_zdrop, p->fb_flag, ez_alt);   /* :341 */
		const int rn = count_flanked_cigar(ez_ref), an = count_flanked_cigar(ez_alt);   /* :343-344 */
		if (rn == 1 && an > 1) ref_support += 1;                    /* :353-356 */
		else if (an == 1 && rn > 1) alt_support += 1;
	}
	free(renc); free(cenc); free(qenc);
	ev->ref_support = ref_support; ev->alt_support = alt_support; ev->both_found = 0;   /* :316,:320-321 */
	ev->aligned = 1;                                                /* :372 */
}

/* indelope.nim:157-183 */
static orc_list assemble(const ihp_params *p, const ihp_batch_in *in, int64_t r0, int64_t r1, int32_t *n_pre)
{
	orc_list contigs = {0, 0, 0};
	for (int64_t i = r0; i < r1; ++i) {
		if (in->mapq[i] < p->min_mapq_assemble) continue;          /* :164 */
		if (in->read_skip && in->read_skip[i]) continue;           /* :165 */
		const uint8_t *seq = in->bases + in->read_off[i];
		int64_t n = in->read_off[i + 1] - in->read_off[i], lo = 0, hi = n, o = 0;
		if (in->trim_lo) { trim_given(in, i, n, &lo, &hi); o = lo; }
		else if (in->quals) o = orc_read_trim(in->quals + in->read_off[i], n, p->trim_min_qual, &lo, &hi);  /* :168 */
		else if (n == 1) hi = 0;                                   /* quals NULL = all 255: a == high == 0 empties the read (:28-30) */
		int64_t tl = hi - lo;
		int64_t min_overlap = (int64_t)(p->min_overlap_pct * (double)tl);   /* :169 */
		ihp_contig *qc = orc_make_contig(seq + lo, tl, in->read_start[i] + o, 1);
		orc_list_insert(&contigs, qc, min_overlap, p->max_mismatch);
	}
	*n_pre = (int32_t)contigs.n;                                   /* :171 */
	return orc_combine(contigs, p->max_mismatch, p->combine_min_support, 1, p->combine_min_overlap); /* :176 */
}

static void run_region(const ihp_params *p, const ihp_batch_in *in, int32_t r, region_res *out)
{
	const int64_t r0 = in->region_read_off[r], r1 = in->region_read_off[r + 1];
	const uint8_t *slice = in->ref_bases + in->ref_off[r];
	const int64_t L = in->ref_off[r + 1] - in->ref_off[r], origin = in->ref_origin[r];
	const int K = p->K;
	memset(out, 0, sizeof(*out));
	out->contigs = assemble(p, in, r0, r1, &out->n_pre);
	out->res = (ctg_res *)calloc((size_t)(out->contigs.n ? out->contigs.n : 1), sizeof(ctg_res));
	int8_t mat[25];
	orc_matrix(p->match, p->mismatch, mat);
	ksw_extz_t ez; memset(&ez, 0, sizeof(ez));
	ksw_extz_t ez_ref, ez_alt; memset(&ez_ref, 0, sizeof(ez_ref)); memset(&ez_alt, 0, sizeof(ez_alt));
	uint8_t *qenc = 0, *tenc = 0;
	for (int64_t ci = 0; ci < out->contigs.n; ++ci) {
		ihp_contig *ctg = out->contigs.v[ci];
		ctg_res *cr = &out->res[ci];
		if (out->n_pre > p->max_pre_contigs) continue;             /* :209 */
		if (ctg->nreads < p->min_reads || ctg->len < p->min_ctg_len) continue;   /* :211 */
		int64_t max_stop = ctg->start;                             /* :213-216 */
		for (int64_t i = r0; i < r1; ++i) {
			if (in->mapq[i] <= p->min_mapq_stop) continue;
			if (in->read_stop[i] > max_stop) max_stop = in->read_stop[i];
		}
		const int width = (int)((double)(K + 1) / 2.0 - 1.0);       /* :218 */
		/* :220 fai.get(chrom, ctg.start, max_stop+width+50): 0-based, end
		 * inclusive, clamped like htslib's faidx_fetch_seq */
		int64_t beg = ctg->start - origin, end = max_stop + width + p->ref_pad - origin;
		int clamped = 0;
		if (end < beg) { beg = end; clamped = 1; }
		if (beg < 0) { beg = 0; clamped = 1; } else if (L <= beg) { beg = L - 1; clamped = 1; }
		if (end < 0) { end = 0; clamped = 1; } else if (L <= end) { end = L - 1; clamped = 1; }
		int64_t reflen = L > 0 ? end - beg + 1 : 0;
		if (L <= 0) { beg = 0; clamped = 1; }
		const uint8_t *reference = slice + beg;
		cr->flags = IHP_ALN_DONE | (clamped ? IHP_ALN_REF_CLAMPED : 0);
		cr->ref_start = origin + beg; cr->ref_len = (int32_t)reflen;
		qenc = (uint8_t *)realloc(qenc, (size_t)ctg->len + 1);
		tenc = (uint8_t *)realloc(tenc, (size_t)reflen + 1);
		orc_encode(ctg->sequence, ctg->len, qenc);                 /* ksw2.nim:162-163 */
		orc_encode(reference, reflen, tenc);
		ez.n_cigar = 0;                                            /* ksw2.nim:153 */
		orc_ksw_dispatch((int)ctg->len, qenc, (int)reflen, tenc, 5, mat, p->gap_open, p->gap_ext,
		                 p->bw, p->zdrop, p->ksw_flag, &ez);       /* :221 */
		cr->ez.max = (int32_t)ez.max; cr->ez.zdropped = (int32_t)ez.zdropped;
		cr->ez.max_q = ez.max_q; cr->ez.max_t = ez.max_t; cr->ez.mqe = ez.mqe; cr->ez.mqe_t = ez.mqe_t;
		cr->ez.mte = ez.mte; cr->ez.mte_q = ez.mte_q; cr->ez.score = ez.score; cr->ez.n_cigar = ez.n_cigar;
		cr->cigar = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(ez.n_cigar ? ez.n_cigar : 1));
		memcpy(cr->cigar, ez.cigar, sizeof(uint32_t) * (size_t)ez.n_cigar);

		/* Ez.cigar truncation (ksw2.nim:22-33) + query/target_locations (:71-91) */
		int nev = 0, ntrunc = 0;
		{
			uint32_t max_off = (uint32_t)ez.max_q, off = 0;
			for (int i = 0; i < ez.n_cigar; ++i) {
				if (off >= max_off) break;
				uint32_t op = ez.cigar[i] & 0xf, len = ez.cigar[i] >> 4;
				if (op != 2) off += len;
				ntrunc++;
				if (op == 1 || op == 2) nev++;
			}
		}
		if (nev == 0 || nev > p->max_events) continue;             /* :229 */
		cr->events = (ihp_event *)calloc((size_t)nev, sizeof(ihp_event));
		cr->hits = (int32_t **)calloc((size_t)nev, sizeof(int32_t *));
		cr->n_events = nev;
		int64_t toff = ctg->start, qoff = 0;
		int ii = -1;
		for (int i = 0; i < ntrunc; ++i) {
			uint32_t op = ez.cigar[i] & 0xf, len = ez.cigar[i] >> 4;
			if (op == 0) { toff += len; qoff += len; continue; }
			ihp_event *ev = &cr->events[++ii];
			ev->len = len;
			if (op == 1) {                                         /* I */
				ev->type = 0;
				ev->tstart = toff; ev->tstop = toff + 1;
				ev->qstart = qoff; ev->qstop = qoff + len;
				qoff += len;
			} else {                                               /* D */
				ev->type = 1;
				ev->tstart = toff; ev->tstop = toff + len;
				ev->qstart = qoff; ev->qstop = qoff + 1;
				toff += len;
			}
			ev->ref_support = ev->alt_support = ev->both_found = 0;
			ev->gt = IHP_GT_UNKNOWN;
			if ((int64_t)len < p->min_event_len) { ev->status = IHP_EV_SHORT; continue; }   /* :234 */
			if (reflen < K || ctg->len < K) { ev->status = IHP_EV_OOB; continue; }
			int64_t tstart = ev->tstart - ctg->start - width;      /* :236-238 */
			if (tstart < 0) tstart = 0;
			if (tstart + K > reflen) tstart = reflen - K;
			memcpy(ev->ref_kmer, reference + tstart, (size_t)K);   /* :240 */
			int64_t o1 = ev->qstart, o2 = ctg->len - ev->qstop - 1;
			ev->cf_offset = (int32_t)(o1 < o2 ? o1 : o2);          /* :243 */
			int64_t qstart = ev->qstart - width;                   /* :244-246 */
			if (qstart < 0) qstart = 0;
			if (qstart + K > ctg->len) qstart = ctg->len - K;
			memcpy(ev->alt_kmer, ctg->sequence + qstart, (size_t)K);   /* :248 */
			if (memcmp(ev->alt_kmer, ev->ref_kmer, (size_t)K) == 0) {   /* :255-262 */
				qstart = ev->qstart - 3;
				if (qstart < 0) qstart = 0;
				if (qstart + K > ctg->len) {
					int64_t qend = ev->qstop + 4 < ctg->len ? ev->qstop + 4 : ctg->len;
					if (qend - K < 0) { ev->status = IHP_EV_OOB; continue; }
					memcpy(ev->alt_kmer, ctg->sequence + qend - K, (size_t)K);
				} else {
					memcpy(ev->alt_kmer, ctg->sequence + qstart, (size_t)K);
				}
			}
			int same = memcmp(ev->alt_kmer, ev->ref_kmer, (size_t)K) == 0;
			if (same && (ev->qstart == 0 || distinct_bytes(ev->alt_kmer, K) == 1)) {
				ev->status = IHP_EV_SAME_KMER; continue;           /* :264 */
			}
			if (distinct_bytes(ev->ref_kmer, K) < 3) { ev->status = IHP_EV_LOW_CPLX; continue; }   /* :266 */
			if (same) { ev->status = IHP_EV_BUG_SAME; continue; }  /* :268-275 */
			uint64_t refe, alte;
			if (!mincode(ev->ref_kmer, K, &refe) || !mincode(ev->alt_kmer, K, &alte)) {
				ev->status = IHP_EV_NON_ACGT; continue;
			}
			int32_t *hits = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)(r1 - r0 ? r1 - r0 : 1));
			cr->hits[ii] = hits;
			for (int64_t ri = r0; ri < r1; ++ri) {                 /* :293-311 */
				hits[ri - r0] = hits[(r1 - r0) + ri - r0] = -1;
				if (in->mapq[ri] < p->min_mapq_tally) continue;
				int rf, af;
				tally_read(in->bases + in->read_off[ri], in->read_off[ri + 1] - in->read_off[ri],
				           K, refe, alte, &rf, &af, &hits[ri - r0], &hits[(r1 - r0) + ri - r0]);
				ev->ref_support += rf; ev->alt_support += af;
				if (rf && af) ev->both_found += 1;
			}
			ev->status = IHP_EV_TALLIED;
			ev->fallback_needed = ev->both_found > 0;              /* :313 */
			ev->kmer_ref_support = ev->ref_support; ev->kmer_alt_support = ev->alt_support;
			ev->kmer_both_found = ev->both_found;
			if (ev->fallback_needed && p->fallback)
				fallback_align(p, in, r0, r1, ctg, reference, reflen, ev, &ez_ref, &ez_alt);   /* :313-372 */
			ihp_genotype_t g;
			orc_genotype(ev->ref_support, ev->alt_support, p->error, &g);   /* :379 */
			ev->gt = g.gt; ev->gl[0] = g.gl[0]; ev->gl[1] = g.gl[1]; ev->gl[2] = g.gl[2];
			ev->qual = orc_genotype_qual(&g);
		}
	}
	free(ez.cigar); free(ez_ref.cigar); free(ez_alt.cigar); free(qenc); free(tenc);
}

/* ---- batch driver ----------------------------------------------------------- */
typedef struct {
	const ihp_params *p; const ihp_batch_in *in; region_res *res;
	int32_t lo, hi; int64_t cnt[3];
} job_t;

static void *worker(void *arg)
{
	job_t *j = (job_t *)arg;
	orc_cnt_compares = orc_cnt_cells = orc_cnt_kmers = 0;
	for (int32_t r = j->lo; r < j->hi; ++r) run_region(j->p, j->in, r, &j->res[r]);
	j->cnt[0] = orc_cnt_compares; j->cnt[1] = orc_cnt_cells; j->cnt[2] = orc_cnt_kmers;
	return 0;
}

int orc_run_regions_mt(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out, int nthreads)
{
	if (!p || !in || !out || p->struct_size != (int32_t)sizeof(ihp_params)) return IHP_E_ARG;
	if (p->K < 1 || p->K > 31) return IHP_E_ARG;
	const int32_t R = in->n_regions;
	memset(out, 0, sizeof(*out));
	region_res *res = (region_res *)calloc((size_t)(R ? R : 1), sizeof(region_res));
	if (nthreads < 1) nthreads = 1;
	if (nthreads > R) nthreads = R ? R : 1;
	job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
	pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
	for (int t = 0; t < nthreads; ++t) {
		jobs[t].p = p; jobs[t].in = in; jobs[t].res = res;
		jobs[t].lo = (int32_t)((int64_t)R * t / nthreads);
		jobs[t].hi = (int32_t)((int64_t)R * (t + 1) / nthreads);
	}
	if (nthreads == 1) worker(&jobs[0]);
	else {
		for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, worker, &jobs[t]);
		for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
	}
	g_counters[0] = g_counters[1] = g_counters[2] = 0;
	for (int t = 0; t < nthreads; ++t)
		for (int k = 0; k < 3; ++k) g_counters[k] += jobs[t].cnt[k];
	free(jobs); free(th);

	/* flatten */
	int64_t C = 0, E = 0, W = 0, B = 0, Hn = 0;
	for (int32_t r = 0; r < R; ++r) {
		C += res[r].contigs.n;
		const int64_t nr = in->region_read_off[r + 1] - in->region_read_off[r];
		for (int64_t c = 0; c < res[r].contigs.n; ++c) {
			B += res[r].contigs.v[c]->len;
			W += res[r].res[c].ez.n_cigar;
			E += res[r].res[c].n_events;
			for (int e = 0; e < res[r].res[c].n_events; ++e) if (res[r].res[c].hits[e]) Hn += nr;
		}
	}
	out->n_regions = R; out->n_contigs = C; out->n_events = E; out->n_cigar_words = W; out->n_bases = B; out->n_hits = Hn;
#define ALLOC(T, n) ((T *)calloc((size_t)((n) ? (n) : 1), sizeof(T)))
	out->status = ALLOC(int32_t, R); out->n_contigs_pre = ALLOC(int32_t, R);
	out->contig_off = ALLOC(int64_t, R + 1);
	out->ctg_start = ALLOC(int64_t, C); out->ctg_nreads = ALLOC(int64_t, C);
	out->ctg_seq_off = ALLOC(int64_t, C + 1);
	out->ctg_seq = ALLOC(uint8_t, B); out->ctg_support = ALLOC(uint32_t, B);
	out->aln_flags = ALLOC(int32_t, C); out->aln_ref_start = ALLOC(int64_t, C);
	out->aln_ref_len = ALLOC(int32_t, C); out->aln_ez = ALLOC(ihp_ez, C);
	out->cigar_off = ALLOC(int64_t, C + 1); out->cigar = ALLOC(uint32_t, W);
	out->event_off = ALLOC(int64_t, C + 1); out->events = ALLOC(ihp_event, E);
	out->hit_off = ALLOC(int64_t, E + 1); out->ref_hit = ALLOC(int32_t, Hn); out->alt_hit = ALLOC(int32_t, Hn);
#undef ALLOC
	int64_t c = 0, b = 0, wd = 0, ev = 0, hn = 0;
	for (int32_t r = 0; r < R; ++r) {
		const int64_t nr = in->region_read_off[r + 1] - in->region_read_off[r];
		out->status[r] = res[r].status; out->n_contigs_pre[r] = res[r].n_pre;
		out->contig_off[r] = c;
		for (int64_t k = 0; k < res[r].contigs.n; ++k, ++c) {
			ihp_contig *g = res[r].contigs.v[k];
			ctg_res *cr = &res[r].res[k];
			out->ctg_start[c] = g->start; out->ctg_nreads[c] = g->nreads;
			out->ctg_seq_off[c] = b;
			memcpy(out->ctg_seq + b, g->sequence, (size_t)g->len);
			memcpy(out->ctg_support + b, g->support, (size_t)g->len * sizeof(uint32_t));
			b += g->len;
			out->aln_flags[c] = cr->flags; out->aln_ref_start[c] = cr->ref_start;
			out->aln_ref_len[c] = cr->ref_len; out->aln_ez[c] = cr->ez;
			out->cigar_off[c] = wd;
			if (cr->ez.n_cigar) memcpy(out->cigar + wd, cr->cigar, (size_t)cr->ez.n_cigar * 4);
			wd += cr->ez.n_cigar;
			out->event_off[c] = ev;
			if (cr->n_events) memcpy(out->events + ev, cr->events, (size_t)cr->n_events * sizeof(ihp_event));
			for (int e = 0; e < cr->n_events; ++e) {
				out->hit_off[ev + e] = hn;
				if (!cr->hits[e]) continue;
				memcpy(out->ref_hit + hn, cr->hits[e], sizeof(int32_t) * (size_t)nr);
				memcpy(out->alt_hit + hn, cr->hits[e] + nr, sizeof(int32_t) * (size_t)nr);
				hn += nr;
				free(cr->hits[e]);
			}
			ev += cr->n_events;
			free(cr->cigar); free(cr->events); free(cr->hits);
			orc_contig_free(g);
		}
		free(res[r].contigs.v); free(res[r].res);
	}
	out->contig_off[R] = c; out->ctg_seq_off[C] = b; out->cigar_off[C] = wd; out->event_off[C] = ev;
	out->hit_off[E] = hn;
	free(res);
	return 0;
}

/* Throughput probe for bench.py's cpu_baseline: every thread runs its contiguous share of the regions
 * `reps` times and throws the results away (no flattening), so the timed work is the path itself. */
typedef struct { const ihp_params *p; const ihp_batch_in *in; int32_t lo, hi; int reps; } bjob_t;

static void free_region(region_res *rr)
{
	for (int64_t k = 0; k < rr->contigs.n; ++k) {
		for (int e = 0; e < rr->res[k].n_events; ++e) free(rr->res[k].hits[e]);
		free(rr->res[k].cigar); free(rr->res[k].events); free(rr->res[k].hits);
		orc_contig_free(rr->contigs.v[k]);
	}
	free(rr->contigs.v); free(rr->res);
}

static void *bworker(void *arg)
{
	bjob_t *j = (bjob_t *)arg;
	region_res rr;
	for (int rep = 0; rep < j->reps; ++rep)
		for (int32_t r = j->lo; r < j->hi; ++r) { run_region(j->p, j->in, r, &rr); free_region(&rr); }
	return 0;
}

int orc_bench_regions(const ihp_params *p, const ihp_batch_in *in, int nthreads, int reps)
{
	if (!p || !in || p->struct_size != (int32_t)sizeof(ihp_params)) return IHP_E_ARG;
	const int32_t R = in->n_regions;
	if (nthreads < 1) nthreads = 1;
	if (nthreads > R) nthreads = R ? R : 1;
	bjob_t *jobs = (bjob_t *)calloc((size_t)nthreads, sizeof(bjob_t));
	pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
	for (int t = 0; t < nthreads; ++t) {
		jobs[t].p = p; jobs[t].in = in; jobs[t].reps = reps;
		jobs[t].lo = (int32_t)((int64_t)R * t / nthreads);
		jobs[t].hi = (int32_t)((int64_t)R * (t + 1) / nthreads);
	}
	for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, bworker, &jobs[t]);
	for (int t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
	free(jobs); free(th);
	return 0;
}

int orc_run_regions(const ihp_params *p, const ihp_batch_in *in, ihp_batch_out *out)
{
	return orc_run_regions_mt(p, in, out, 1);
}

void orc_free_out(ihp_batch_out *out)
{
	if (!out) return;
	free(out->status); free(out->n_contigs_pre); free(out->contig_off);
	free(out->ctg_start); free(out->ctg_nreads); free(out->ctg_seq_off);
	free(out->ctg_seq); free(out->ctg_support);
	free(out->aln_flags); free(out->aln_ref_start); free(out->aln_ref_len); free(out->aln_ez);
	free(out->cigar_off); free(out->cigar); free(out->event_off); free(out->events);
	free(out->hit_off); free(out->ref_hit); free(out->alt_hit);
	memset(out, 0, sizeof(*out));
}
