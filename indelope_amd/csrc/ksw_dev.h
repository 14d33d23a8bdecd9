// ksw_dev.h -- one ksw2 extension alignment per wavefront
// (reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388, the only native code on the path).
//
// The reference sweeps anti-diagonals r and, on each, 16-byte SSE blocks of the
// band; here the band cells of one anti-diagonal are the lanes of the wave.  To
// be bit-exact the kernel keeps the reference's observable layout: one zeroed
// byte block u|v|x|y|s|sf|qr in LDS (:173-175) indexed by absolute target
// position, the 16-rounded computed band [st,en] (:205), the 16-wide score
// stores that run past the band and leave stale bytes (:215-228), wrapping int8
// arithmetic with the SSE2 path's unsigned max/min (:131-132,:271-272), the
// H[en0] special case (:318) and the 4-strided tie order of the exact max
// (:323-348).  The traceback matrix p (80 B per diagonal at w=50) streams to a
// per-wave HBM scratch row by row, coalesced; the walk back is done by lane 0.
#pragma once
#include "ihp_common.h"

namespace ihp {

struct KswOut {
	int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, n_cigar;
};

__device__ __forceinline__ uint8_t enc_base(uint8_t c)
{                                                       // ksw2.nim:127-132
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	}
	return 4;
}

__device__ __forceinline__ size_t ksw_lds_bytes(int qlen, int tlen)
{
	const size_t T = (size_t)((tlen + 15) / 16) * 16, Q = (size_t)((qlen + 15) / 16) * 16 + 16;
	return 6 * T + Q + 4 * T;                            // bytes block + H
}

// Band of anti-diagonal r (:196-205).  Returns false when st > en (band exit).
__device__ __forceinline__ bool ksw_band(int r, int qlen, int tlen, int w, int &st0, int &en0, int &st, int &en)
{
	int s = 0, e = tlen - 1;
	if (s < r - qlen + 1) s = r - qlen + 1;
	if (e > r) e = r;
	if (s < (r - w + 1) >> 1) s = (r - w + 1) >> 1;
	if (e > (r + w) >> 1) e = (r + w) >> 1;
	if (s > e) return false;
	st0 = s; en0 = e;
	st = s / 16 * 16; en = (e + 16) / 16 * 16 - 1;
	return true;
}

// cig_tmp: per-wave scratch for the reversed CIGAR (capacity cig_cap words).
// Returns n_cigar in out.n_cigar with the CIGAR (final order) in cig_tmp[0..n),
// or out.n_cigar = -1 if cig_cap was too small.
__device__ inline void ksw_wave(const uint8_t *query, int qlen, const uint8_t *target, int tlen,
                                const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap,
                                KswOut &out)
{
	const int lane = lane_id();
	int w = P.w;
	const int q = P.q, e = P.e, qe = q + e, flag = P.flag;
	out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;   // :81-86
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = 0;
	if (P.m <= 0 || qlen <= 0 || tlen <= 0) return;      // :147
	if (-P.min_sc > 2 * (q + e)) return;                 // :171
	if (w < 0) w = tlen > qlen ? tlen : qlen;            // :161
	const int tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	const int T = tlen_ * 16;
	uint8_t *u = lds, *v = u + T, *x = v + T, *y = x + T, *s = y + T, *sf = s + T, *qr = sf + T;
	const int QR = qlen_ * 16 + 16;
	int32_t *H = (int32_t *)(lds + 6 * T + QR);
	const uint8_t qe2 = (uint8_t)(qe * 2), sc_mch = (uint8_t)P.sc_mch, sc_mis = (uint8_t)P.sc_mis;
	const uint8_t m1 = (uint8_t)(P.m - 1), max_sc8 = (uint8_t)(P.sc_mch + qe * 2);
	const bool with_cigar = !(flag & KSW_EZ_SCORE_ONLY), right = (flag & KSW_EZ_RIGHT) != 0;

	for (int i = lane; i < 5 * T; i += 64) lds[i] = 0;   // kcalloc :173
	for (int i = lane; i < T; i += 64) { sf[i] = i < tlen ? (P.encode_ascii ? enc_base(target[i]) : target[i]) : 0; H[i] = KSW_NEG_INF; }
	for (int i = lane; i < QR; i += 64) {                // :187
		uint8_t b = 0;
		if (i < qlen) { b = query[qlen - 1 - i]; if (P.encode_ascii) b = enc_base(b); }
		qr[i] = b;
	}
	WSYNC();

	int last_st = -1, last_en = -1;
	int ez_max = 0, ez_max_t = -1, ez_max_q = -1, mqe = KSW_NEG_INF, mqe_t = -1, mte = KSW_NEG_INF, mte_q = -1;
	int score = KSW_NEG_INF, zdropped = 0;
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0, st, en;
		if (!ksw_band(r, qlen, tlen, w, st0, en0, st, en)) { zdropped = 1; break; }   // :200-203
		uint8_t x1, v1;                                  // :207-211
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) { x1 = x[st - 1]; v1 = v[st - 1]; }
			else x1 = v1 = 0;
		} else { x1 = 0; v1 = r ? (uint8_t)q : 0; }
		WSYNC();
		if (en >= r && lane == 0) { y[r] = 0; u[r] = r ? (uint8_t)q : 0; }   // :212
		// scores (:214-228): 16-byte groups starting at st0, past en0 up to 15 bytes
		{
			const uint8_t *qrr = qr + (qlen - 1 - r);
			const int nsc = ((en0 - st0) / 16 + 1) * 16;
			for (int j0 = 0; j0 < nsc; j0 += 64) {
				const int j = st0 + j0 + lane;
				if (j0 + lane < nsc) {
					const uint8_t sq = sf[j], sq2 = qrr[j];
					uint8_t val = sq == sq2 ? sc_mch : sc_mis;
					if (sq == m1 || sq2 == m1) val = 0;
					s[j] = val;
				}
			}
		}
		WSYNC();
		// core recurrence over [st,en], top chunk first so lane t still sees x[t-1], v[t-1] of r-1
		const int nch = (en - st + 64) / 64;
		uint8_t *pr = p + (size_t)r * ncol - st;
		for (int c = nch - 1; c >= 0; --c) {
			const int t = st + c * 64 + lane;
			const bool act = t <= en;
			uint8_t z = 0, a = 0, b = 0, ut = 0, vt1 = 0, d = 0;
			if (act) {
				z = (uint8_t)(s[t] + qe2);
				const uint8_t xt1 = t == st ? x1 : x[t - 1];
				vt1 = t == st ? v1 : v[t - 1];
				a = (uint8_t)(xt1 + vt1);
				ut = u[t];
				b = (uint8_t)(y[t] + ut);
			}
			WSYNC();
			if (act) {
				if (!right) d = (int8_t)a > (int8_t)z ? 1 : 0;              // :265
				else        d = (int8_t)z > (int8_t)a ? 0 : 1;              // :291
				z = (int8_t)z > 0 ? z : 0;                                  // :271 (SSE2 path)
				z = z > a ? z : a;                                          // :272 unsigned max
				if (!right) { if ((int8_t)b > (int8_t)z) d = 2; }           // :273-274
				else        { if (!((int8_t)z > (int8_t)b)) d = 2; }        // :299-300
				z = z > b ? z : b;                                          // :131
				z = z < max_sc8 ? z : max_sc8;                              // :132
				u[t] = (uint8_t)(z - vt1);                                  // :133
				v[t] = (uint8_t)(z - ut);                                   // :134
				z = (uint8_t)(z - (uint8_t)q);
				a = (uint8_t)(a - z);
				b = (uint8_t)(b - z);
				if (!right) {
					const bool ta = (int8_t)a > 0, tb = (int8_t)b > 0;      // :277-282
					x[t] = ta ? a : 0; y[t] = tb ? b : 0;
					d |= (ta ? 0x08 : 0) | (tb ? 0x10 : 0);
				} else {
					const bool ta = 0 > (int8_t)a, tb = 0 > (int8_t)b;      // :303-308
					x[t] = ta ? 0 : a; y[t] = tb ? 0 : b;
					d |= (ta ? 0 : 0x08) | (tb ? 0 : 0x10);
				}
				if (with_cigar) pr[t] = d;                                  // :283
			}
		}
		WSYNC();
		// exact max with the 32-bit score array (:312-357)
		int max_H, max_t;
		if (r > 0) {
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			const int ncell = en0 - st0 + 1, nhc = (ncell + 63) / 64;
			int bh = -0x7fffffff - 1, br = 0x7fffffff, bt = 0;
			for (int c = nhc - 1; c >= 0; --c) {
				const int t = st0 + c * 64 + lane;
				const bool act = t <= en0;
				int h = -0x7fffffff - 1, rk = 0x7fffffff;
				if (act) {
					if (t == en0) {                                         // :318
						h = (en0 > 0 ? H[en0 - 1] + (int)u[en0] : H[en0] + (int)v[en0]) - qe;
						rk = 0;
					} else {
						h = H[t] + (int)v[t] - qe;                          // :323-329, :345
						rk = t < en1 ? 1 + (((t - st0) & 3) << 24) + ((t - st0) >> 2)
						             : 1 + (4 << 24) + (t - en1);
					}
				}
				WSYNC();
				if (act) H[t] = h;
				int ht = h, rt = rk, tt = t;
				for (int dd = 32; dd >= 1; dd >>= 1) {
					const int oh = __shfl_xor(ht, dd, 64), orr = __shfl_xor(rt, dd, 64), ot = __shfl_xor(tt, dd, 64);
					if (oh > ht || (oh == ht && orr < rt)) { ht = oh; rt = orr; tt = ot; }
				}
				if (ht > bh || (ht == bh && rt < br)) { bh = ht; br = rt; bt = tt; }
			}
			max_H = bh; max_t = bt;
		} else {
			WSYNC();
			const int h0 = (int)v[0] - qe - qe;                             // :349
			if (lane == 0) H[0] = h0;
			max_H = h0; max_t = 0;
		}
		WSYNC();
		if (en0 == tlen - 1) { const int h = H[en0]; if (h > mte) { mte = h; mte_q = r - en; } }       // :351-352
		if (r - st0 == qlen - 1) { const int h = H[st0]; if (h > mqe) { mqe = h; mqe_t = st0; } }      // :353-354
		{                                                                   // ksw_apply_zdrop :88-104
			const int t = max_t;
			if (max_H > ez_max) { ez_max = max_H; ez_max_t = t; ez_max_q = r - t; }
			else if (t >= ez_max_t && r - t >= ez_max_q) {
				const int tl = t - ez_max_t, ql = (r - t) - ez_max_q;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (P.zdrop >= 0 && ez_max - max_H > P.zdrop + l * e) { zdropped = 1; break; }
			}
		}
		if (r == qlen + tlen - 2 && en0 == tlen - 1) score = H[tlen - 1];   // :356-357
		last_st = st; last_en = en;
	}
	out.max = ez_max; out.zdropped = zdropped; out.max_q = ez_max_q; out.max_t = ez_max_t;
	out.mqe = mqe; out.mqe_t = mqe_t; out.mte = mte; out.mte_q = mte_q; out.score = score;
	WSYNC();
	if (!with_cigar) return;
	// backtrack (:47-79, :380-385); off[r]/off_end[r] are recomputed from r
	int i0, j0;
	if (!zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) { i0 = tlen - 1; j0 = qlen - 1; }
	else if (ez_max_t >= 0 && ez_max_q >= 0) { i0 = ez_max_t; j0 = ez_max_q; }
	else return;
	__threadfence_block();
	int n_cigar = 0;
	if (lane == 0) {
		int i = i0, j = j0, state = 0, ok = 1;
		uint32_t cur = 0;                                    // run being built (len<<4|op), 0 = none
		auto push = [&](uint32_t op, int len) {              // ksw_push_cigar :31-41
			if (cur && (cur & 0xf) == op) { cur += (uint32_t)len << 4; return; }
			if (cur) { if (n_cigar < cig_cap) cig_tmp[n_cigar] = cur; else ok = 0; n_cigar++; }
			cur = (uint32_t)len << 4 | op;
		};
		while (i >= 0 && j >= 0) {
			const int r = i + j;
			int st0, en0, st, en, force_state = -1;
			ksw_band(r, qlen, tlen, w, st0, en0, st, en);
			if (i < st) force_state = 2;
			if (i > en) force_state = 1;
			const uint32_t tmp = force_state < 0 ? p[(size_t)r * ncol + i - st] : 0;
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) { push(0, 1); --i; --j; }
			else if (state == 1 || state == 3) { push(2, 1); --i; }
			else { push(1, 1); --j; }
		}
		if (i >= 0) push(2, i + 1);
		if (j >= 0) push(1, j + 1);
		if (cur) { if (n_cigar < cig_cap) cig_tmp[n_cigar] = cur; else ok = 0; n_cigar++; }
		if (!ok) n_cigar = -1;
		else if (!(flag & KSW_EZ_REV_CIGAR))
			for (int k = 0; k < n_cigar >> 1; ++k) {
				const uint32_t t = cig_tmp[k];
				cig_tmp[k] = cig_tmp[n_cigar - 1 - k]; cig_tmp[n_cigar - 1 - k] = t;
			}
	}
	n_cigar = bcast(n_cigar, 0);
	out.n_cigar = n_cigar;
	WSYNC();
}

}  // namespace ihp
