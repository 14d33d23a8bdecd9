// ksw_dev.h -- one ksw2 extension alignment per wavefront
// (reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388, the only native code on the path).
//
// The reference sweeps anti-diagonals r and, on each, 16-byte SSE blocks of the
// band; here the band cells of one anti-diagonal are the lanes of the wave.  To be
// bit-exact the kernel reproduces everything observable of the reference's work
// arrays: values indexed by absolute target position t, the 16-rounded computed
// band [st,en] (:205) whose padding cells feed real cells at the band edge and can
// be walked by the traceback, score bytes s[] refreshed only in 16-byte groups from
// st0 (:215-228) and stale elsewhere, wrapping int8 arithmetic with the SSE2 path's
// unsigned max/min (:131-132,:271-272), the H[en0] special case (:318) and the
// 4-strided tie order of the exact max (:323-348).
//
// LDS layout per wave: one 8-byte record per target position
//   {x, v, u, y | s, target base, 0, 0}            (the u,v,x,y,s,sf arrays of :173-175)
// + the reversed query (:187) + the 32-bit H[] (:177-178).  A cell reads its own
// record (ds_read_b64), x/v of t-1 (ds_read_b32), one query byte and H[t]; it writes
// its record and H[t].  The per-diagonal max is a DPP wave reduction + ballot.  The
// traceback matrix p (n_col*16 B per diagonal, 80 B at w=50) streams to per-wave HBM
// scratch, coalesced; the walk back is wave-parallel: 64 lanes speculate along the
// current move direction and a ballot finds the run length.
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

__host__ __device__ __forceinline__ size_t ksw_lds_bytes(int qlen, int tlen)
{
	const size_t T = (size_t)((tlen + 15) / 16) * 16, Q = (size_t)((qlen + 15) / 16) * 16 + 16;
	return 8 * T + Q + 4 * T + 64;                       // records, reversed query, H[], generic score matrix (KSW_EZ_GENERIC_SC)
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

#define IHP_DPP(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)

// max over the 64 lanes (all lanes active); result is wave-uniform.  Four v_max_i32 with a DPP
// operand (xor 1, xor 2, 8-lane mirror, 16-lane mirror) leave each row's max in all its lanes.
__device__ __forceinline__ int wave_max_i32(int v)
{
	asm("s_nop 4\n\t"          // covers VALU-writes-EXEC -> DPP (5 wait states) as well as VGPR -> DPP (2)
	    "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1"
	    : "+v"(v));
	const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
	const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
	const int ab = a > b ? a : b, cd = c > d ? c : d;
	return ab > cd ? ab : cd;
}

__device__ __forceinline__ int sext8(unsigned v) { return (int)(signed char)(v & 0xff); }

// Wave-parallel walk back through p (:47-79, :380-385).  off[r]/off_end[r] are recomputed from r.
// Lane k speculates on the cell reached after k moves in the direction of the current state; a
// ballot gives the length of the run, so a CIGAR of n ops costs O(n + len/64) round trips to HBM.
// PACKED 1: p is ksw_narrow.h's slot matrix (80 dwords per slot, a nibble of four compare bits per cell, eight diagonals per
// dword) instead of the reference's n_col*16 bytes per diagonal; 2: ksw_pair.h's (four diagonals per 16-bit half, HALF = this
// alignment's half); 3: ksw_duo.h's (one byte per (diagonal, QUERY position): a nibble per alignment, ncol = bytes per diagonal).
template <int PACKED = 0, int HALF = 0>
__device__ inline void ksw_backtrack_wave(const uint8_t *p, int ncol, int qlen, int tlen, int w, int flag,
                                          int zdropped, int ez_max_t, int ez_max_q,
                                          uint32_t *cig_tmp, int cig_cap, KswOut &out)
{
	const int lane = lane_id();
	int i, j;
	if (!zdropped && !(flag & KSW_EZ_EXTZ_ONLY)) { i = tlen - 1; j = qlen - 1; }
	else if (ez_max_t >= 0 && ez_max_q >= 0) { i = ez_max_t; j = ez_max_q; }
	else return;
	int n_cigar = 0, ok = 1, state = 0;
	uint32_t cur = 0;                                    // run being built (len<<4|op); 0 = none
	auto push = [&](uint32_t op, int len) {              // ksw_push_cigar :31-41 (wave-uniform)
		if (cur && (cur & 0xf) == op) { cur += (uint32_t)len << 4; return; }
		if (cur) { if (n_cigar < cig_cap) { if (lane == 0) cig_tmp[n_cigar] = cur; } else ok = 0; n_cigar++; }
		cur = (uint32_t)len << 4 | op;
	};
	while (i >= 0 && j >= 0) {
		// lane k looks at the cell reached after k moves in the direction of the current state
		const int di = state == 2 ? 0 : 1, dj = state == 1 ? 0 : 1;
		const int ik = i - lane * di, jk = j - lane * dj;
		int outst = -1;
		if (ik >= 0 && jk >= 0) {
			const int rr = ik + jk;
			int st0, en0, st, en, force_state = -1;
			ksw_band(rr, qlen, tlen, w, st0, en0, st, en);
			if (ik < st) force_state = 2;                // :56
			if (ik > en) force_state = 1;                // :57
			unsigned tmp = 0;
			if (force_state < 0) {
				if (!PACKED) tmp = p[(size_t)rr * ncol + ik - st];
				else if (PACKED == 3) {
					const unsigned nib = (unsigned)p[(size_t)rr * ncol + jk] >> (4 * HALF) & 15u;
					tmp = ((nib & 1) ? 2u : ((nib >> 1) & 1u)) | ((nib & 12) << 1);
				} else {
					const unsigned nib = PACKED == 1 ? ((const unsigned *)p)[(size_t)((rr >> 3) + (st >> 4)) * 80 + (ik - st)] >> (4 * (7 - (rr & 7))) & 15u
					                                 : ((const unsigned *)p)[(size_t)((rr >> 2) + (st >> 4)) * 80 + (ik - st)] >> (16 * HALF + 4 * (3 - (rr & 3))) & 15u;
					if (PACKED == 1) tmp = ((nib & 4) ? 2u : (nib >> 3)) | ((nib & 2) << 2) | ((nib & 1) << 4);
					else tmp = ((nib & 1) ? 2u : ((nib >> 1) & 1u)) | ((nib & 12) << 1);   // ksw_pair.h: bit 1 = :265, bit 0 = :273, bits 2, 3
				}
			}
			int s = state;
			if (s == 0) s = tmp & 7;                     // :64
			else if (!((tmp >> (s + 2)) & 1)) s = 0;     // :65
			if (s == 0) s = tmp & 7;                     // :66
			if (force_state >= 0) s = force_state;       // :67
			outst = s;
		}
		const unsigned long long cont = ballot(outst == state);
		int n = cont == ~0ull ? 64 : ctz64(~cont);
		if (n == 0) { state = __builtin_amdgcn_readlane(outst, 0); n = 1; }
		const int mi = state == 2 ? 0 : 1, mj = (state == 1 || state == 3) ? 0 : 1;
		push(state == 0 ? 0u : ((state == 1 || state == 3) ? 2u : 1u), n);           // :68-71
		i -= n * mi; j -= n * mj;
	}
	if (i >= 0) push(2, i + 1);                          // :73
	if (j >= 0) push(1, j + 1);                          // :74
	if (cur) { if (n_cigar < cig_cap) { if (lane == 0) cig_tmp[n_cigar] = cur; } else ok = 0; n_cigar++; }
	WSYNC();
	if (!ok) { out.n_cigar = -1; return; }
	if (!(flag & KSW_EZ_REV_CIGAR)) {                    // :75-77
		for (int k0 = 0; k0 < n_cigar >> 1; k0 += 64) {
			const int k = k0 + lane;
			if (k < n_cigar >> 1) {
				const uint32_t a = cig_tmp[k], b = cig_tmp[n_cigar - 1 - k];
				cig_tmp[k] = b; cig_tmp[n_cigar - 1 - k] = a;
			}
		}
		WSYNC();
	}
	out.n_cigar = n_cigar;
}

// cig_tmp: per-wave scratch for the CIGAR (capacity cig_cap words).  out.n_cigar = -1 if too small.
// gmat (KSW_EZ_GENERIC_SC, m <= 8): the m x m score matrix; the scores of a diagonal are then mat[target][query] on the
// true band [st0, en0] only (:229-232), not match / mismatch / wildcard on 16-byte groups.  KSW_EZ_APPROX_MAX (and
// KSW_EZ_APPROX_DROP with it) replaces the exact maximum by the reference's H0 walk (:358-374); KSW_EZ_SCORE_ONLY skips
// the traceback.  The reference never sets these flags on the path; they are here for the FFI seam.
__device__ inline void ksw_wave(const uint8_t *query, int qlen, const uint8_t *target, int tlen,
                                const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap,
                                KswOut &out, long long *pacc = nullptr, const signed char *gmat = nullptr)
{
	const long long tc0 = pacc ? (long long)clock64() : 0;
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
	const int T = tlen_ * 16, QR = qlen_ * 16 + 16;
	uint2 *rec = (uint2 *)lds;                           // [T]
	uint8_t *qr = lds + 8 * (size_t)T;                   // [QR]
	int *H = (int *)(lds + 8 * (size_t)T + QR);          // [T]
	uint8_t *gm = lds + 8 * (size_t)T + QR + 4 * (size_t)T;   // [64] generic score matrix
	const bool generic = (flag & KSW_EZ_GENERIC_SC) != 0 && gmat != nullptr, approx = (flag & KSW_EZ_APPROX_MAX) != 0;
	if (generic && lane < 64) gm[lane] = lane < P.m * P.m ? (uint8_t)gmat[lane] : 0;
	const unsigned qe2 = (unsigned)(qe * 2) & 0xff, sc_mch = (unsigned)P.sc_mch & 0xff, sc_mis = (unsigned)P.sc_mis & 0xff;
	const unsigned m1 = (unsigned)(P.m - 1) & 0xff, max_sc8 = (unsigned)(P.sc_mch + qe * 2) & 0xff, q8 = (unsigned)q & 0xff;
	const bool with_cigar = !(flag & KSW_EZ_SCORE_ONLY), right = (flag & KSW_EZ_RIGHT) != 0;

	for (int i = lane; i < T; i += 64) {                 // kcalloc :173, memcpy :188, H init :177-178
		const unsigned tb = i < tlen ? (P.encode_ascii ? enc_base(target[i]) : target[i]) : 0;
		rec[i] = make_uint2(0u, tb << 8);
		H[i] = KSW_NEG_INF;
	}
	for (int i = lane; i < QR; i += 64) {                // :187
		uint8_t b = 0;
		if (i < qlen) { b = query[qlen - 1 - i]; if (P.encode_ascii) b = enc_base(b); }
		qr[i] = b;
	}
	WSYNC();

	const long long tc1 = pacc ? (long long)clock64() : 0;
	int last_st = -1, last_en = -1;
	int ez_max = 0, ez_max_t = -1, ez_max_q = -1, mqe = KSW_NEG_INF, mqe_t = -1, mte = KSW_NEG_INF, mte_q = -1;
	int score = KSW_NEG_INF, zdropped = 0;
	int H0 = 0, last_H0_t = 0;                           // KSW_EZ_APPROX_MAX
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0, st, en;
		if (!ksw_band(r, qlen, tlen, w, st0, en0, st, en)) { zdropped = 1; break; }   // :200-203
		// left boundary of the computed band (:207-211)
		const bool nb_valid = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
		const unsigned xv_edge = st > 0 ? 0u : (r ? q8 << 8 : 0u);                    // x1 | v1<<8 when not read from LDS
		const int sc_hi = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;                      // last refreshed score byte (:215)
		const int qoff = qlen - 1 - r;                                                // qrr = qr + qoff (:193)
		const int Hen0m1 = (r > 0 && en0 > 0) ? H[en0 - 1] : 0;                       // old value, for :318
		const int en1 = st0 + (en0 - st0) / 4 * 4;
		uint8_t *pr = p + (size_t)r * ncol - st;
		int bh = -0x7fffffff - 1, brk = 0x7fffffff, bt = 0, Hen0 = 0, Hst0 = 0;
		// the 16-byte score stores run up to 15 bytes past en (:215-228); those bytes stay behind as the
		// stale s[] of later diagonals, so lanes in (en, sc_hi] refresh their score byte and nothing else
		const int hi = generic ? en : sc_hi < T ? (sc_hi > en ? sc_hi : en) : (T - 1 > en ? T - 1 : en);
		const int nch = (hi - st + 64) / 64;
		for (int c = nch - 1; c >= 0; --c) {             // top chunk first: t-1 of r-1 is still intact below
			const int tb = st + c * 64, t = tb + lane;
			const bool act = t <= en;
			int h = -0x7fffffff - 1;
			if (!act && t <= hi) {
				const unsigned ry = rec[t].y, sfb = (ry >> 8) & 0xff, qb = qr[qoff + t];
				unsigned sv = sfb == qb ? sc_mch : sc_mis;
				if (sfb == m1 || qb == m1) sv = 0;
				rec[t].y = sv | sfb << 8;
			}
			if (act) {
				const uint2 R = rec[t];
				unsigned xv;
				if (t == st) xv = nb_valid ? (rec[t - 1].x & 0xffffu) : xv_edge;
				else xv = rec[t - 1].x & 0xffffu;
				unsigned sv = R.y & 0xff;
				const unsigned sfb = (R.y >> 8) & 0xff;
				if (generic) {                           // :229-232
					if (t >= st0 && t <= en0) sv = gm[sfb * (unsigned)P.m + qr[qoff + t]];
				} else if (t >= st0 && t <= sc_hi) {     // :214-228
					const unsigned qb = qr[qoff + t];
					sv = sfb == qb ? sc_mch : sc_mis;
					if (sfb == m1 || qb == m1) sv = 0;
				}
				unsigned ut = (R.x >> 16) & 0xff, yt = R.x >> 24;
				if (t == r) { yt = 0; ut = r ? q8 : 0; } // :212 (en >= r whenever t == r is computed)
				const unsigned xt1 = xv & 0xff, vt1 = xv >> 8;
				unsigned z = (sv + qe2) & 0xff;
				unsigned a = (xt1 + vt1) & 0xff, b = (yt + ut) & 0xff;
				unsigned d;
				if (!right) d = sext8(a) > sext8(z) ? 1 : 0;                          // :265
				else        d = sext8(z) > sext8(a) ? 0 : 1;                          // :291
				z = sext8(z) > 0 ? z : 0;                                             // :271 (SSE2 path)
				z = z > a ? z : a;                                                    // :272 unsigned max
				if (!right) { if (sext8(b) > sext8(z)) d = 2; }                       // :273-274
				else        { if (!(sext8(z) > sext8(b))) d = 2; }                    // :299-300
				z = z > b ? z : b;                                                    // :131
				z = z < max_sc8 ? z : max_sc8;                                        // :132
				const unsigned un = (z - vt1) & 0xff, vn = (z - ut) & 0xff;           // :133-134
				z = (z - q8) & 0xff;
				a = (a - z) & 0xff; b = (b - z) & 0xff;
				unsigned xn, yn;
				if (!right) {
					const bool ta = sext8(a) > 0, tb2 = sext8(b) > 0;                 // :277-282
					xn = ta ? a : 0; yn = tb2 ? b : 0;
					d |= (ta ? 0x08u : 0u) | (tb2 ? 0x10u : 0u);
				} else {
					const bool ta = 0 > sext8(a), tb2 = 0 > sext8(b);                 // :303-308
					xn = ta ? 0 : a; yn = tb2 ? 0 : b;
					d |= (ta ? 0u : 0x08u) | (tb2 ? 0u : 0x10u);
				}
				rec[t] = make_uint2(xn | vn << 8 | un << 16 | yn << 24, sv | sfb << 8);
				if (with_cigar) pr[t] = (uint8_t)d;                                   // :283
				// exact max, 32-bit (:312-349)
				if (r > 0) {
					if (t == en0) h = (en0 > 0 ? Hen0m1 + (int)un : H[t] + (int)vn) - qe;   // :318
					else if (t >= st0 && t < en0) h = H[t] + (int)vn - qe;            // :323-329, :345
				} else if (t == 0) h = (int)vn - qe - qe;                             // :349
				if (t >= st0 && t <= en0) H[t] = h;
			}
			const bool intrue = act && t >= st0 && t <= en0;
			const int hm = wave_max_i32(intrue ? h : -0x7fffffff - 1);
			const unsigned long long m = ballot(intrue && h == hm);
			if (m) {
				// tie order of :320-348: en0 first, then the vector part by stride class, then the tail
				int rk, tt;
				const int ben0 = en0 - tb;
				if (ben0 >= 0 && ben0 < 64 && ((m >> ben0) & 1)) { rk = 0; tt = en0; }
				else {
					const int nv = en1 - tb;                                          // lanes below nv are in the vector part
					const unsigned long long mv = nv <= 0 ? 0ull : (nv >= 64 ? m : (m & ((1ull << nv) - 1)));
					rk = 0x7fffffff; tt = 0;
					if (mv) {
						for (int j = 0; j < 4; ++j) {
							const int sh = (j - (tb - st0)) & 3;
							const unsigned long long cm = mv & (0x1111111111111111ull << sh);
							if (cm) { tt = tb + ctz64(cm); rk = 1 + (j << 24) + ((tt - st0) >> 2); break; }
						}
					} else {
						const unsigned long long mt = m & ~mv;
						tt = tb + ctz64(mt); rk = 1 + (4 << 24) + (tt - en1);
					}
				}
				if (hm > bh || (hm == bh && rk < brk)) { bh = hm; brk = rk; bt = tt; }
			}
			{
				const int l1 = en0 - tb, l2 = st0 - tb;
				if (l1 >= 0 && l1 < 64) Hen0 = __builtin_amdgcn_readlane(h, l1);
				if (l2 >= 0 && l2 < 64) Hst0 = __builtin_amdgcn_readlane(h, l2);
			}
		}
		LDS_ORDER();                                     // p stores keep streaming; LDS is in order per wave
		// ksw_apply_zdrop :88-104
		auto apply_zdrop = [&](int Hv, int t) -> bool {
			if (Hv > ez_max) { ez_max = Hv; ez_max_t = t; ez_max_q = r - t; }
			else if (t >= ez_max_t && r - t >= ez_max_q) {
				const int tl = t - ez_max_t, ql = (r - t) - ez_max_q;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (P.zdrop >= 0 && ez_max - Hv > P.zdrop + l * e) { zdropped = 1; return true; }
			}
			return false;
		};
		if (!approx) {
			const int max_H = bh, max_t = bt;
			if (en0 == tlen - 1 && Hen0 > mte) { mte = Hen0; mte_q = r - en; }            // :351-352 (rounded en)
			if (r - st0 == qlen - 1 && Hst0 > mqe) { mqe = Hst0; mqe_t = st0; }           // :353-354
			if (apply_zdrop(max_H, max_t)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) score = Hen0;                    // :356-357 (en0 == tlen-1)
		} else {                                                                          // approximate max (:358-374)
			if (r > 0) {
				const bool in0 = last_H0_t >= st0 && last_H0_t <= en0, in1 = last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0;
				if (in0 && in1) {
					const int d0 = (int)((rec[last_H0_t].x >> 8) & 0xffu) - qe;             // v8[last_H0_t]
					const int d1 = (int)((rec[last_H0_t + 1].x >> 16) & 0xffu) - qe;        // u8[last_H0_t + 1]
					if (d0 > d1) H0 += d0;
					else { H0 += d1; ++last_H0_t; }
				} else if (in0) H0 += (int)((rec[last_H0_t].x >> 8) & 0xffu) - qe;
				else { ++last_H0_t; H0 += (int)((rec[last_H0_t].x >> 16) & 0xffu) - qe; }
				H0 = uni(H0); last_H0_t = uni(last_H0_t);
				if ((flag & KSW_EZ_APPROX_DROP) && apply_zdrop(H0, last_H0_t)) break;
			} else { H0 = uni((int)((rec[0].x >> 8) & 0xffu)) - qe - qe; last_H0_t = 0; }
			if (r == qlen + tlen - 2 && en0 == tlen - 1) score = H0;
		}
		last_st = st; last_en = en;
	}
	out.max = ez_max; out.zdropped = zdropped; out.max_q = ez_max_q; out.max_t = ez_max_t;
	out.mqe = mqe; out.mqe_t = mqe_t; out.mte = mte; out.mte_q = mte_q; out.score = zdropped ? KSW_NEG_INF : score;   // (the reference tests the z-drop before it takes the score of the last diagonal, :355-357: a sweep that stopped has none)
	WSYNC();
	const long long tc2 = pacc ? (long long)clock64() : 0;
	if (pacc && lane == 0) { pacc[0] += tc1 - tc0; pacc[1] += tc2 - tc1; pacc[3] += 1; }
	if (!with_cigar) return;
	ksw_backtrack_wave(p, ncol, qlen, tlen, w, flag, zdropped, ez_max_t, ez_max_q, cig_tmp, cig_cap, out);
	if (pacc && lane == 0) pacc[2] += (long long)clock64() - tc2;
}

}  // namespace ihp
