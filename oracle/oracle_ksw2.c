/*
 * oracle_ksw2.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar restatement of src/ksw2/csrc/ksw2_extz2_sse.c:113-388 (the only native
 * code on the path).  The SSE file computes 16-lane blocks; this file walks the
 * same cells one byte lane at a time but keeps everything that is observable in
 * the results identical: the single zeroed work block u|v|x|y|s|sf|qr (:173-175)
 * with its 16-wide unaligned score stores that run past the band (:215-228) and
 * leave stale bytes behind, the 16-rounded computed band [st,en] recorded in
 * off/off_end (:205,:261), int8 wrap-around arithmetic, the unsigned max/min of
 * block 2 (:131-132), the H[en0] special case (:318) and the 4-strided tie order
 * of the exact max (:323-348).  Validated against the compiled reference in
 * oracle/_ref (tests/test_oracle_ksw2.py).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle_internal.h"

_Thread_local int64_t orc_cnt_cells;
static int g_variant = 0;          /* 0: SSE2 path, 1: SSE4.1 path */
static orc_ksw_fn g_impl = 0;

void orc_ksw_set_variant(int variant) { g_variant = variant ? 1 : 0; }
void orc_set_ksw_impl(orc_ksw_fn fn) { g_impl = fn; }

static void push_cigar(ksw_extz_t *ez, int *n_cigar, uint32_t op, int len)
{                                                   /* :31-41 */
	if (*n_cigar == 0 || op != (ez->cigar[*n_cigar - 1] & 0xf)) {
		if (*n_cigar == ez->m_cigar) {
			ez->m_cigar = ez->m_cigar ? ez->m_cigar << 1 : 4;
			ez->cigar = (uint32_t *)realloc(ez->cigar, (size_t)ez->m_cigar << 2);
		}
		ez->cigar[(*n_cigar)++] = (uint32_t)len << 4 | op;
	} else ez->cigar[*n_cigar - 1] += (uint32_t)len << 4;
}

/* :47-79 with is_rot = 1, with_N = 0 */
static void backtrack(int rev_cigar, const uint8_t *p, const int *off, const int *off_end,
                      int n_col, int i0, int j0, ksw_extz_t *ez)
{
	int n_cigar = 0, i = i0, j = j0, state = 0;
	while (i >= 0 && j >= 0) {
		int force_state = -1, r = i + j;
		uint32_t tmp;
		if (i < off[r]) force_state = 2;
		if (i > off_end[r]) force_state = 1;
		tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) { push_cigar(ez, &n_cigar, 0, 1); --i; --j; }
		else if (state == 1 || state == 3) { push_cigar(ez, &n_cigar, 2, 1); --i; }
		else { push_cigar(ez, &n_cigar, 1, 1); --j; }
	}
	if (i >= 0) push_cigar(ez, &n_cigar, 2, i + 1);
	if (j >= 0) push_cigar(ez, &n_cigar, 1, j + 1);
	if (!rev_cigar)
		for (i = 0; i < n_cigar >> 1; ++i) {
			uint32_t t = ez->cigar[i];
			ez->cigar[i] = ez->cigar[n_cigar - 1 - i]; ez->cigar[n_cigar - 1 - i] = t;
		}
	ez->n_cigar = n_cigar;
}

/* :88-104 with is_rot = 1 */
static int apply_zdrop(ksw_extz_t *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{
	if (H > (int32_t)ez->max) {
		ez->max = (uint32_t)H; ez->max_t = t; ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
			ez->zdropped = 1;
			return 1;
		}
	}
	return 0;
}

static inline uint8_t maxu8(uint8_t a, uint8_t b) { return a > b ? a : b; }
static inline uint8_t minu8(uint8_t a, uint8_t b) { return a < b ? a : b; }

void orc_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                   int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                   ksw_extz_t *ez)
{
	int r, t, qe = q + e, n_col_, tlen_, qlen_, last_st, last_en, max_sc, min_sc;
	int with_cigar = !(flag & KSW_EZ_SCORE_ONLY), approx_max = !!(flag & KSW_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	uint8_t *mem, *u, *v, *x, *y, *s, *sf, *qr, *p = 0;
	int *off = 0, *off_end = 0;

	/* ksw_reset_extz :81-86 */
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0; ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0; ez->zdropped = 0;
	if (m <= 0 || qlen <= 0 || tlen <= 0) return;    /* :147 */

	const uint8_t qe2 = (uint8_t)((q + e) * 2);
	const uint8_t sc_mch = (uint8_t)mat[0], sc_mis = (uint8_t)mat[1];
	const uint8_t m1 = (uint8_t)(m - 1);
	const uint8_t max_sc8 = (uint8_t)(mat[0] + (q + e) * 2);

	if (w < 0) w = tlen > qlen ? tlen : qlen;        /* :161 */
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;               /* :171 */

	const size_t T = (size_t)tlen_ * 16;
	mem = (uint8_t *)calloc((size_t)tlen_ * 6 + qlen_ + 1, 16);
	u = mem; v = u + T; x = v + T; y = x + T; s = y + T; sf = s + T; qr = sf + T;
	if (!approx_max) {
		H = (int32_t *)malloc(T * 4);
		for (t = 0; t < (int)T; ++t) H[t] = KSW_NEG_INF;
	}
	const int ncol = n_col_ * 16;
	if (with_cigar) {
		p = (uint8_t *)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int *)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];   /* :187 */
	memcpy(sf, target, (size_t)tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		uint8_t x1, v1;
		uint8_t *qrr = qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;    /* :196-199 */
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }    /* :200-203 */
		st0 = st; en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;   /* :205 */
		if (st > 0) {                                /* :207-211 */
			if (st - 1 >= last_st && st - 1 <= last_en) { x1 = x[st - 1]; v1 = v[st - 1]; }
			else x1 = v1 = 0;
		} else { x1 = 0; v1 = r ? (uint8_t)q : 0; }
		if (en >= r) { y[r] = 0; u[r] = r ? (uint8_t)q : 0; }   /* :212 */
		if (!(flag & KSW_EZ_GENERIC_SC)) {           /* :214-228, 16 bytes per step */
			for (t = st0; t <= en0; t += 16) {
				uint8_t tmp16[16];
				for (int k = 0; k < 16; ++k) {
					uint8_t sq = sf[t + k], sq2 = qrr[t + k];
					uint8_t val = sq == sq2 ? sc_mch : sc_mis;
					if (sq == m1 || sq2 == m1) val = 0;
					tmp16[k] = val;
				}
				memcpy(s + t, tmp16, 16);
			}
		} else {
			for (t = st0; t <= en0; ++t) s[t] = (uint8_t)mat[sf[t] * m + qrr[t]];
		}
		orc_cnt_cells += en - st + 1;
		uint8_t *pr = with_cigar ? p + (size_t)r * ncol - st : 0;
		if (with_cigar) { off[r] = st; off_end[r] = en; }   /* :261 */
		const int right = !!(flag & KSW_EZ_RIGHT);
		uint8_t xprev = x1, vprev = v1;              /* lane t-1 of the previous diagonal */
		for (t = st; t <= en; ++t) {                 /* :116-137 + :262-310 */
			uint8_t z = (uint8_t)(s[t] + qe2);
			uint8_t xt1 = xprev, vt1 = vprev;
			xprev = x[t]; vprev = v[t];
			uint8_t a = (uint8_t)(xt1 + vt1);
			uint8_t ut = u[t];
			uint8_t b = (uint8_t)(y[t] + ut);
			uint8_t d = 0;
			if (with_cigar) {
				if (!right) d = (int8_t)a > (int8_t)z ? 1 : 0;          /* :265 */
				else        d = (int8_t)z > (int8_t)a ? 0 : 1;          /* :291 */
			}
			if (g_variant) z = (int8_t)z > (int8_t)a ? z : a;           /* SSE4.1 :267 */
			else { z = (int8_t)z > 0 ? z : 0; z = maxu8(z, a); }        /* SSE2 :271-272 */
			if (with_cigar) {
				if (!right) { if ((int8_t)b > (int8_t)z) d = 2; }       /* :268-269 */
				else        { if (!((int8_t)z > (int8_t)b)) d = 2; }    /* :294-295 */
			}
			z = maxu8(z, b);                         /* :131 */
			z = minu8(z, max_sc8);                   /* :132 */
			u[t] = (uint8_t)(z - vt1);               /* :133 */
			v[t] = (uint8_t)(z - ut);                /* :134 */
			z = (uint8_t)(z - (uint8_t)q);
			a = (uint8_t)(a - z);
			b = (uint8_t)(b - z);
			if (!with_cigar || !right) {
				/* score-only stores max(a,0): the same value as the left-align form */
				int ta = (int8_t)a > 0, tb = (int8_t)b > 0;
				x[t] = ta ? a : 0; y[t] = tb ? b : 0;
				if (ta) d |= 0x08;
				if (tb) d |= 0x10;
			} else {
				int ta = 0 > (int8_t)a, tb = 0 > (int8_t)b;             /* :303-308 */
				x[t] = ta ? 0 : a; y[t] = tb ? 0 : b;
				if (!ta) d |= 0x08;
				if (!tb) d |= 0x10;
			}
			if (with_cigar) pr[t] = d;
		}
		if (!approx_max) {                           /* :312-357 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] - qe : H[en0] + v[en0] - qe;   /* :318 */
				max_t = en0;
				for (i = 0; i < 4; ++i) { HH[i] = max_H; tt[i] = max_t; }
				for (t = st0; t < en1; t += 4) {     /* :323-339: four stride classes */
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i] - qe;
						if (H[t + i] > HH[i]) { HH[i] = H[t + i]; tt[i] = t; }
					}
				}
				for (i = 0; i < 4; ++i)              /* :342-343 */
					if (max_H < HH[i]) { max_H = HH[i]; max_t = tt[i] + i; }
				for (; t < en0; ++t) {               /* :344-348 */
					H[t] += (int32_t)v[t] - qe;
					if (H[t] > max_H) { max_H = H[t]; max_t = t; }
				}
			} else { H[0] = v[0] - qe - qe; max_H = H[0]; max_t = 0; }
			if (en0 == tlen - 1 && H[en0] > ez->mte) { ez->mte = H[en0]; ez->mte_q = r - en; }   /* rounded en, :352 */
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) { ez->mqe = H[st0]; ez->mqe_t = st0; }
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                     /* :358-374 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t] - qe, d1 = u[last_H0_t + 1] - qe;
					if (d0 > d1) H0 += d0;
					else { H0 += d1; ++last_H0_t; }
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t] - qe;
				} else {
					++last_H0_t; H0 += u[last_H0_t] - qe;
				}
				if ((flag & KSW_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e)) break;
			} else { H0 = v[0] - qe - qe; last_H0_t = 0; }
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st; last_en = en;
	}
	free(mem);
	if (!approx_max) free(H);
	if (with_cigar) {                                /* :380-387 */
		int rev_cigar = !!(flag & KSW_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & KSW_EZ_EXTZ_ONLY))
			backtrack(rev_cigar, p, off, off_end, ncol, tlen - 1, qlen - 1, ez);
		else if (ez->max_t >= 0 && ez->max_q >= 0)
			backtrack(rev_cigar, p, off, off_end, ncol, ez->max_t, ez->max_q, ez);
		free(p); free(off);
	}
}

void orc_ksw_dispatch(int qlen, const uint8_t *query, int tlen, const uint8_t *target,
                      int8_t m, const int8_t *mat, int8_t q, int8_t e, int w, int zdrop, int flag,
                      ksw_extz_t *ez)
{
	if (g_impl) g_impl(0, qlen, query, tlen, target, m, mat, q, e, w, zdrop, flag, ez);
	else orc_ksw_extz2(qlen, query, tlen, target, m, mat, q, e, w, zdrop, flag, ez);
}

int orc_ksw_extz2_batch(int32_t n, const uint8_t *queries, const int64_t *q_off,
                        const uint8_t *targets, const int64_t *t_off,
                        int8_t m, const int8_t *mat, int8_t q, int8_t e,
                        int w, int zdrop, int flag,
                        ihp_ez *ez, uint32_t *cigar, int64_t cigar_cap, int64_t *cigar_off)
{
	ksw_extz_t z; memset(&z, 0, sizeof(z));
	int64_t used = 0; int rc = 0;
	cigar_off[0] = 0;
	for (int32_t i = 0; i < n; ++i) {
		z.n_cigar = 0;                               /* ksw2.nim:153 */
		orc_ksw_dispatch((int)(q_off[i + 1] - q_off[i]), queries + q_off[i],
		                 (int)(t_off[i + 1] - t_off[i]), targets + t_off[i],
		                 m, mat, q, e, w, zdrop, flag, &z);
		ez[i].max = (int32_t)z.max; ez[i].zdropped = (int32_t)z.zdropped;
		ez[i].max_q = z.max_q; ez[i].max_t = z.max_t; ez[i].mqe = z.mqe; ez[i].mqe_t = z.mqe_t;
		ez[i].mte = z.mte; ez[i].mte_q = z.mte_q; ez[i].score = z.score; ez[i].n_cigar = z.n_cigar;
		if (used + z.n_cigar <= cigar_cap) { if (z.n_cigar) memcpy(cigar + used, z.cigar, (size_t)z.n_cigar * 4); }
		else rc = IHP_E_CAPACITY;
		used += z.n_cigar;
		cigar_off[i + 1] = used;
	}
	free(z.cigar);
	return rc;
}

/* ksw2.nim:127-132: A/a 0, C/c 1, G/g 2, T/t 3, everything else 4 */
void orc_encode(const uint8_t *dna, int64_t n, uint8_t *out)
{
	for (int64_t i = 0; i < n; ++i) {
		switch (dna[i]) {
		case 'A': case 'a': out[i] = 0; break;
		case 'C': case 'c': out[i] = 1; break;
		case 'G': case 'g': out[i] = 2; break;
		case 'T': case 't': out[i] = 3; break;
		default: out[i] = 4;
		}
	}
}

/* ksw2.nim:135-140 */
void orc_matrix(int8_t match, int8_t mismatch, int8_t out25[25])
{
	for (int i = 0; i < 5; ++i)
		for (int j = 0; j < 5; ++j)
			out25[i * 5 + j] = (i == 4 || j == 4) ? 0 : (i == j ? match : mismatch);
}
