/*
 * oracle_contig.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * CPU restatement of src/contig.nim:1-281 and of the read trim of
 * src/indelope.nim:23-38.  Each function cites the lines it follows.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

_Thread_local int64_t orc_cnt_compares;

/* contig.nim:44-47 (default rule) and :287-290 (test rule). */
int orc_allowed(int rule, uint32_t qsup, uint32_t tsup, int64_t qreads, int64_t treads)
{
	if (rule == IHP_ALLOW_SUPPORT)
		return (qsup < 3u && tsup > 3u * qsup) || (tsup < 3u && qsup > 3u * tsup);
	/* uint32 products wrap exactly as Nim's 3'u32 * qsup does */
	return (qsup < 3u && tsup > 3u * qsup && qreads > 3 * (int64_t)qsup) ||
	       (tsup < 3u && qsup > 3u * tsup && treads > 3 * (int64_t)tsup);
}

void orc_corr_push(orc_corrvec *v, int64_t qoff, int64_t toff, int qbest)
{
	if (v->n == v->cap) {
		v->cap = v->cap ? v->cap * 2 : 4;
		v->v = (ihp_correction *)realloc(v->v, (size_t)v->cap * sizeof(ihp_correction));
	}
	v->v[v->n].qoff = qoff; v->v[v->n].toff = toff; v->v[v->n].qbest = qbest; v->v[v->n]._pad = 0;
	v->n++;
}

static void corr_copy(orc_corrvec *dst, const orc_corrvec *src)
{
	dst->n = 0;
	for (int64_t i = 0; i < src->n; ++i)
		orc_corr_push(dst, src->v[i].qoff, src->v[i].toff, src->v[i].qbest);
}

/* One offset of the scan: contig.nim:87-105 (phase 1, qo0 = 0, to0 = o) and
 * :115-133 (phase 2, qo0 = o, to0 = 0).  Returns mm; *ma_out = matches. */
static int64_t walk(const ihp_contig *q, const ihp_contig *t, int64_t qo, int64_t to,
                    int64_t max_mismatch, int rule, orc_corrvec *cur, int64_t *ma_out)
{
	int64_t mm = 0, ma = 0;
	cur->n = 0;
	while (qo < q->len && to < t->len) {
		orc_cnt_compares++;
		if (q->sequence[qo] != t->sequence[to]) {
			if (!orc_allowed(rule, q->support[qo], t->support[to], q->nreads, t->nreads)) {
				mm += 1;
				if (mm > max_mismatch) break;
			} else {
				orc_corr_push(cur, qo, to, q->support[qo] > t->support[to]);
			}
		} else {
			ma += 1;
		}
		qo += 1; to += 1;
	}
	*ma_out = ma;
	return mm;
}

/* contig.nim:70-141 */
void orc_slide_core(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                    int64_t max_mismatch, int rule, orc_match *out)
{
	int64_t omin = -(q->len - min_overlap);          /* :78 */
	int64_t omax = t->len - min_overlap;             /* :79 */
	int64_t obest = IHP_UNALIGNED;                   /* :80 */
	int64_t best_ma = min_overlap - 1;               /* :81 */
	int64_t best_mm = max_mismatch + 1;              /* :82 */
	orc_corrvec cur = {0, 0, 0};
	out->corr.n = 0;
	for (int64_t o = 0; o <= omax; ++o) {            /* :86 */
		int64_t ma, mm = walk(q, t, 0, o, max_mismatch, rule, &cur, &ma);
		/* :107 -- `and` binds tighter than `or` */
		if (mm <= max_mismatch && (ma > best_ma || (ma == best_ma && mm < best_mm))) {
			best_ma = ma; best_mm = mm; obest = o;
			corr_copy(&out->corr, &cur);             /* :111 value copy */
		}
	}
	int64_t lim = omin < 0 ? -omin : omin;           /* :114 abs(omin) */
	for (int64_t o = 1; o <= lim; ++o) {
		int64_t ma, mm = walk(q, t, o, 0, max_mismatch, rule, &cur, &ma);
		if (mm <= max_mismatch && (ma > best_ma || (ma == best_ma && mm < best_mm))) {
			best_ma = ma; best_mm = mm; obest = -o;
			corr_copy(&out->corr, &cur);
		}
	}
	free(cur.v);
	out->matches = best_ma; out->offset = obest; out->mismatches = best_mm; out->contig_i = -1; /* :141 */
}

static int ensure_cap(ihp_contig *c, int64_t need, int grow)
{
	if (need <= c->cap) return 0;
	if (!grow) return IHP_E_CAPACITY;
	int64_t ncap = c->cap ? c->cap : 64;
	while (ncap < need) ncap *= 2;
	c->sequence = (uint8_t *)realloc(c->sequence, (size_t)ncap);
	c->support = (uint32_t *)realloc(c->support, (size_t)ncap * sizeof(uint32_t));
	c->cap = ncap;
	return 0;
}

static int in_set(const int64_t *set, int64_t n, int64_t x)
{
	for (int64_t i = 0; i < n; ++i) if (set[i] == x) return 1;
	return 0;
}

/* contig.nim:156-222 */
int orc_insert_core(ihp_contig *t, ihp_contig *q, const orc_match *m, int grow)
{
	if (m->offset == IHP_UNALIGNED) return 0;       /* :159 */
	int64_t off = m->offset, aoff = off < 0 ? -off : off;
	/* capacity first, so a failure modifies nothing */
	int64_t newlen;
	if (off < 0) {
		newlen = aoff + t->len;
		if (q->len > newlen) newlen = q->len;
	} else {
		newlen = t->len;
		if (off + q->len > newlen) newlen = off + q->len;
	}
	int rc = ensure_cap(t, newlen, grow);
	if (rc) return rc;

	int64_t *dont = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->corr.n ? m->corr.n : 1));
	int64_t ndont = 0;
	for (int64_t i = 0; i < m->corr.n; ++i) {        /* :161-173 */
		const ihp_correction *c = &m->corr.v[i];
		if (c->qbest) {
			t->sequence[c->toff] = q->sequence[c->qoff];
			t->support[c->toff] = q->support[c->qoff];
		} else {
			q->sequence[c->qoff] = t->sequence[c->toff];
			q->support[c->qoff] = t->support[c->toff];
		}
		dont[ndont++] = off < 0 ? c->qoff : c->toff;
	}
	if (off < 0) {                                   /* :180-205 */
		int64_t tl = t->len;
		memmove(t->sequence + aoff, t->sequence, (size_t)tl);
		memmove(t->support + aoff, t->support, (size_t)tl * sizeof(uint32_t));
		memcpy(t->sequence, q->sequence, (size_t)aoff);
		memcpy(t->support, q->support, (size_t)aoff * sizeof(uint32_t));
		int64_t nl = aoff + tl;
		if (q->len > nl) {                           /* :191-195 */
			int64_t d = q->len - nl;
			memcpy(t->sequence + nl, q->sequence + (q->len - d), (size_t)d);
			memset(t->support + nl, 0, (size_t)d * sizeof(uint32_t));
			nl += d;
		}
		for (int64_t i = aoff; i < q->len; ++i) {    /* :198-200 */
			if (in_set(dont, ndont, i)) continue;
			t->support[i] += q->support[i];
		}
		t->len = nl;
		t->nreads += q->nreads;                      /* :203 */
		t->start = q->start;                         /* :204 */
		free(dont);
		return 0;
	}
	int64_t original_len = t->len;                   /* :210 */
	if (off + q->len > t->len) {                     /* :211-213 setLen zero-fills */
		int64_t nl = off + q->len;
		memset(t->sequence + t->len, 0, (size_t)(nl - t->len));
		memset(t->support + t->len, 0, (size_t)(nl - t->len) * sizeof(uint32_t));
		t->len = nl;
	}
	int64_t stop = q->len + off < t->len ? q->len + off : t->len;
	for (int64_t i = off; i < stop; ++i) {           /* :216-221 */
		if (in_set(dont, ndont, i)) continue;
		int64_t qoff = i - off;
		t->support[i] += q->support[qoff];
		if (i >= original_len) t->sequence[i] = q->sequence[qoff];
	}
	t->nreads += q->nreads;                          /* :222 */
	free(dont);
	return 0;
}

/* contig.nim:49-68 */
void orc_trim_core(ihp_contig *c, int64_t min_support)
{
	int64_t a = 0;
	uint32_t ms = (uint32_t)min_support;             /* uint32(min_support) */
	while (a < c->len - 1 && c->support[a] < ms) a += 1;
	c->start += a;                                   /* :54 */
	if (a >= c->len - 1) {                           /* :56-60 */
		c->len = 0; c->nreads = 0;
		return;
	}
	int64_t b = c->len - 1;
	while (c->support[b] < ms && b > a) b -= 1;      /* :63 */
	if (a > 0 || b <= c->len - 1) {                  /* :66-68 */
		int64_t nl = b - a + 1;
		memmove(c->sequence, c->sequence + a, (size_t)nl);
		memmove(c->support, c->support + a, (size_t)nl * sizeof(uint32_t));
		c->len = nl;
	}
}

/* indelope.nim:23-38 */
int64_t orc_read_trim(const uint8_t *quals, int64_t n, int min_quality, int64_t *lo, int64_t *hi)
{
	int64_t high = n - 1, a = 0;
	uint8_t mq = (uint8_t)min_quality;
	while (a < high && quals[a] < mq) a += 1;        /* :25 */
	if (a == high) { *lo = 0; *hi = 0; return a; }   /* :28-30 */
	int64_t b = high;
	while (b > a && quals[b] < mq) b -= 1;           /* :33 */
	if (a != 0 || b != high) { *lo = a; *hi = b + 1; }
	else { *lo = 0; *hi = n; }
	return a;
}

/* ---- contig lists (seq[Contig] of refs) ---------------------------------- */
void orc_list_push(orc_list *l, ihp_contig *c)
{
	if (l->n == l->cap) {
		l->cap = l->cap ? l->cap * 2 : 8;
		l->v = (ihp_contig **)realloc(l->v, (size_t)l->cap * sizeof(*l->v));
	}
	l->v[l->n++] = c;
}

ihp_contig *orc_make_contig(const uint8_t *dna, int64_t n, int64_t start, uint32_t support)
{                                                    /* contig.nim:143-150 */
	ihp_contig *c = (ihp_contig *)calloc(1, sizeof(*c));
	c->cap = n > 64 ? n : 64;
	c->sequence = (uint8_t *)malloc((size_t)c->cap);
	c->support = (uint32_t *)malloc((size_t)c->cap * sizeof(uint32_t));
	if (n) memcpy(c->sequence, dna, (size_t)n);
	for (int64_t i = 0; i < n; ++i) c->support[i] = support;
	c->len = n; c->nreads = (int64_t)support; c->start = start;
	return c;
}

void orc_contig_free(ihp_contig *c)
{
	if (!c) return;
	free(c->sequence); free(c->support); free(c);
}

/* contig.nim:32-36 */
static int64_t match_cmp(const orc_match *a, const orc_match *b)
{
	if (a->matches == b->matches) return a->mismatches - b->mismatches;
	return b->matches - a->matches;
}

/* contig.nim:224-240.  Returns 1 and fills *best if something aligned. */
int orc_best_match(orc_list *contigs, const ihp_contig *q, int64_t min_overlap,
                   int64_t max_mismatch, orc_match *best)
{
	int found = 0;
	orc_match cand; memset(&cand, 0, sizeof(cand));
	for (int64_t i = 0; i < contigs->n; ++i) {
		if (contigs->v[i] == q) continue;            /* :227 ref equality */
		orc_slide_core(q, contigs->v[i], min_overlap, max_mismatch, IHP_ALLOW_DEFAULT, &cand);
		if (cand.offset == IHP_UNALIGNED) continue;
		cand.contig_i = i;
		/* stable sort + [0]  ==  keep the first element that no later one
		 * beats strictly under match_sort (:239-240) */
		if (!found || match_cmp(&cand, best) < 0) {
			orc_corrvec keep = best->corr;
			*best = cand; best->corr = keep;
			corr_copy(&best->corr, &cand.corr);
			found = 1;
		}
	}
	free(cand.corr.v);
	if (!found) best->offset = IHP_UNALIGNED;        /* :234-237 */
	return found;
}

/* contig.nim:243-248 ; takes ownership of q (freed when merged). */
void orc_list_insert(orc_list *contigs, ihp_contig *q, int64_t min_overlap, int64_t max_mismatch)
{
	orc_match ma; memset(&ma, 0, sizeof(ma));
	if (orc_best_match(contigs, q, min_overlap, max_mismatch, &ma)) {
		orc_insert_core(contigs->v[ma.contig_i], q, &ma, 1);
		orc_contig_free(q);
	} else {
		orc_list_push(contigs, q);
	}
	free(ma.corr.v);
}

/* contig.nim:254-281.  `contigs` is consumed; contigs that were merged away or
 * dropped are freed, the survivors move to the returned list. */
orc_list orc_combine(orc_list contigs, int64_t max_mismatch, int64_t min_support, int again,
                     int64_t combine_min_overlap)
{
	if (again)                                       /* :259-260 */
		contigs = orc_combine(contigs, max_mismatch, 0, 0, combine_min_overlap);
	orc_list result = {0, 0, 0};
	int64_t usedi = 0;
	for (int64_t i = 0; i < contigs.n; ++i) {        /* :265-271 */
		ihp_contig *c = contigs.v[i];
		if (min_support > 0)
			orc_trim_core(c, c->nreads < min_support ? c->nreads : min_support);
		if (c->nreads > 0 && result.n == 0) { orc_list_push(&result, c); usedi = i; }
	}
	if (result.n == 0) {                             /* :272 */
		for (int64_t i = 0; i < contigs.n; ++i) orc_contig_free(contigs.v[i]);
		free(contigs.v);
		return result;
	}
	orc_match ma; memset(&ma, 0, sizeof(ma));
	for (int64_t i = 0; i < contigs.n; ++i) {        /* :274-281 */
		if (i == usedi) continue;
		if (orc_best_match(&result, contigs.v[i], combine_min_overlap, max_mismatch, &ma)) {
			orc_insert_core(result.v[ma.contig_i], contigs.v[i], &ma, 1);
			orc_contig_free(contigs.v[i]);
		} else if (contigs.v[i]->nreads > 0) {
			orc_list_push(&result, contigs.v[i]);
		} else {
			orc_contig_free(contigs.v[i]);
		}
	}
	free(ma.corr.v);
	free(contigs.v);
	return result;
}

/* ---- public wrappers over caller-owned buffers ---------------------------- */
int orc_slide_align(const ihp_contig *q, const ihp_contig *t, int64_t min_overlap,
                    int64_t max_mismatch, int allow_rule, ihp_match *out)
{
	if (!q || !t || !out) return IHP_E_ARG;
	orc_match m; memset(&m, 0, sizeof(m));
	orc_slide_core(q, t, min_overlap, max_mismatch, allow_rule, &m);
	out->matches = m.matches; out->offset = m.offset; out->mismatches = m.mismatches;
	out->contig_i = -1; out->n_corrections = m.corr.n;
	int rc = 0;
	if (m.corr.n > out->corr_cap) rc = IHP_E_CAPACITY;
	else if (m.corr.n) memcpy(out->corrections, m.corr.v, (size_t)m.corr.n * sizeof(ihp_correction));
	free(m.corr.v);
	return rc;
}

int orc_contig_insert(ihp_contig *t, ihp_contig *q, const ihp_match *m)
{
	if (!t || !q || !m) return IHP_E_ARG;
	orc_match om; memset(&om, 0, sizeof(om));
	om.matches = m->matches; om.offset = m->offset; om.mismatches = m->mismatches;
	om.corr.v = m->corrections; om.corr.n = m->n_corrections; om.corr.cap = m->n_corrections;
	return orc_insert_core(t, q, &om, 0);
}

int orc_contig_trim(ihp_contig *c, int64_t min_support)
{
	if (!c) return IHP_E_ARG;
	orc_trim_core(c, min_support);
	return 0;
}
