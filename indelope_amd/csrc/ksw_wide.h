// ksw_wide.h -- register-resident ksw2 sweep for bands wider than one wavefront: the unbanded read-vs-window
// alignments of the genotyper's alignment fallback (src/indelope.nim:312-372; align_to defaults bw = -1, z = -1,
// src/ksw2/ksw2.nim:159).  Reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388.
//
// Same arithmetic as ksw_narrow.h (int8 work values in the top byte of a VGPR, z from one v_perm_b32 into the
// lane's score table, wave-uniform state on the scalar unit), different geometry: NS registers per array hold
// 64*NS consecutive target positions as a RING -- position t lives in slot (t / 64) % NS, lane t % 64, for as
// long as st <= t < st + 64*NS, where st is the 16-rounded band origin of :205.  When st moves on by one SSE
// block the 16 lanes that fell out of the band are re-initialised in place as the positions 64*NS further on
// (x = v = u = y = 0, score never refreshed, H = -inf: the calloc state of :173-178); nothing rotates.  Every
// lane carries its current position T, so "refreshed" (:214-228), "computed" (:205) and "inside the true band"
// (:196-199) are two compares against scalars.  The neighbour t-1 of the previous diagonal comes from a
// one-lane wave rotation (DPP wave_ror:1), lane 0 taking the rotated value of the slot below.
//
// Preconditions (ksw_wide_ok): the 5-letter alphabet and z = s + 2(q+e) > 0 of ksw_narrow_ok, and a band that fits
// the ring: min(qlen, tlen, w+1) + 31 <= 64*NS (15 cells of rounding below st0, the band, 15 cells of refreshed
// scores past en0, and the 16 lanes re-initialised on a move stay outside the computed range of that diagonal).
// NS = 3 covers reads up to 161 bp unbanded, NS = 6 up to 353 bp.
#pragma once
#include "ksw_narrow.h"

namespace ihp {

template <int NS>
__host__ __device__ __forceinline__ size_t ksw_wide_lds_bytes(int qlen, int tlen)
{   // z table + target codes + one selector word per (padded) query position
	return 64 + (size_t)((tlen + 15) / 16) * 16 + 64 * NS + 16 + 4 * ((size_t)((qlen + 15) / 16) * 16 + 64 * NS + 48);
}

template <int NS>
__host__ __device__ __forceinline__ bool ksw_wide_ok(const KswParams &P, int qlen, int tlen)
{
	const int qe2 = 2 * (P.q + P.e);
	const int zm = (int)(signed char)((qe2 + P.sc_mch) & 0xff), zx = (int)(signed char)((qe2 + P.sc_mis) & 0xff);
	const int zw = (int)(signed char)(qe2 & 0xff);
	if (!(P.m == 5 && zm > 0 && zx > 0 && zw > 0)) return false;
	int w = P.w;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int b = qlen < tlen ? qlen : tlen;
	b = b < w + 1 ? b : w + 1;
	return qlen > 0 && tlen > 0 && b + 31 <= 64 * NS;
}

// v_writelane_b32 with a wave-uniform lane select (the compiler routes the select through M0: two SGPR operands
// would exceed the constant bus).  This clang has no __builtin_amdgcn_writelane; bind the intrinsic by name.
extern "C" __device__ int ihp_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

__device__ __forceinline__ int dpp_ror1(int v)
{   // lane l gets v[l-1]; lane 0 gets v[63]
	return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false);
}

// value held for position t (wave-uniform) in a ring array
template <int NS>
__device__ __forceinline__ int ring_pick(const int (&a)[NS], int t)
{
	const int k = (int)(((unsigned)t >> 6) % (unsigned)NS), l = t & 63;
	int v = 0;
#pragma unroll
	for (int j = 0; j < NS; ++j) if (j == k) v = __builtin_amdgcn_readlane(a[j], l);
	return v;
}

// a[t] = s for the wave-uniform position t
template <int NS>
__device__ __forceinline__ void ring_poke(int (&a)[NS], int t, int s)
{
	const int k = (int)(((unsigned)t >> 6) % (unsigned)NS), l = t & 63;
#pragma unroll
	for (int j = 0; j < NS; ++j) if (j == k) a[j] = ihp_writelane_i32(s, l, a[j]);
}

template <int NS>
struct WideState {
	int X[NS], V[NS], U[NS], Y[NS], Z[NS], H[NS], T[NS];
	unsigned T0[NS], T1[NS];
	int st, edge_h;
	int ez_max, ez_max_t, ez_max_q, mqe, mqe_t, mte, mte_q, score;
};

// One anti-diagonal.  Returns true when the sweep must stop (every such exit is a z-drop for the caller, :98-101, :200-203).
template <int NS, bool RIGHT>
__device__ __forceinline__ bool wide_diag(WideState<NS> &F, const NarrowEnv &E, const int r)
{
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	int st0, en0, nst, en;
	if (!ksw_band(r, E.qlen, E.tlen, E.w, st0, en0, nst, en)) return true;              // :200-203
	// block edge x[st-1], v[st-1] (:207-211): taken on the diagonal where st moves (then st-1 is in [last_st, last_en]),
	// 0 otherwise; at st == 0 it is the left boundary x1 = 0, v1 = r ? q : 0
	int ex = 0, ev = 0;
	const bool moved = nst != F.st;
	if (moved) { ex = ring_pick<NS>(F.X, nst - 1); ev = ring_pick<NS>(F.V, nst - 1); F.edge_h = ring_pick<NS>(F.H, nst - 1); }
	else if (nst == 0) ev = r ? E.q24 : 0;
	// H[en0-1] of the previous diagonal for the special case of :318
	const bool has_spec = r > 0 && en0 > 0;
	int Hsp = 0;
	if (has_spec) Hsp = en0 == nst ? F.edge_h : ring_pick<NS>(F.H, en0 - 1);
	// neighbours t-1 of diagonal r-1, before anything is overwritten
	int xp[NS], vp[NS];
	{
		int rx[NS], rv[NS];
#pragma unroll
		for (int k = 0; k < NS; ++k) { rx[k] = dpp_ror1(F.X[k]); rv[k] = dpp_ror1(F.V[k]); }
		const bool l0 = lane == 0;
#pragma unroll
		for (int k = 0; k < NS; ++k) {
			xp[k] = l0 ? rx[(k + NS - 1) % NS] : rx[k];
			vp[k] = l0 ? rv[(k + NS - 1) % NS] : rv[k];
		}
		ring_poke<NS>(xp, nst, ex);
		ring_poke<NS>(vp, nst, ev);
	}
	if (moved) {
		// the 16 positions below the new origin leave the ring; their lanes become the positions 64*NS further on
#pragma unroll
		for (int k = 0; k < NS; ++k) {
			const bool gone = F.T[k] < nst;
			if (ballot(gone)) {
				const int tn = F.T[k] + 64 * NS;
				const uint2 tb = E.tbl[E.tg[gone ? tn : F.T[k]]];
				F.T[k] = gone ? tn : F.T[k];
				F.X[k] = gone ? 0 : F.X[k]; F.V[k] = gone ? 0 : F.V[k]; F.U[k] = gone ? 0 : F.U[k]; F.Y[k] = gone ? 0 : F.Y[k];
				F.Z[k] = gone ? E.ZW24 : F.Z[k]; F.H[k] = gone ? KSW_NEG_INF : F.H[k];
				F.T1[k] = tb.x; F.T0[k] = tb.y;
			}
		}
		F.st = nst;
	}
	const int st = nst;
	const int sc = st0 + ((en0 - st0) / 16 + 1) * 16 - 1;       // last refreshed score position (:215)
	const int top = sc > en ? sc : en;
	uint8_t *pr = E.pb + (size_t)r * E.ncol - st;
	const unsigned *qrow = E.qs + (E.qlen - 1 - r);             // qrr of :193 (selector words)
	const int ur = r ? E.q24 : 0;
	int hk[NS];
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		hk[k] = INTMIN;
		const int t = F.T[k];
		if (!ballot(t <= top)) continue;                         // nothing of this slot is touched on this diagonal
		const int znew = narrow_z(F.T0[k], F.T1[k], qrow[t]);
		const bool ge0 = t >= st0;
		F.Z[k] = (ge0 && t <= sc) ? znew : F.Z[k];              // :214-228
		int ut = F.U[k], yt = F.Y[k];
		if (r <= en) { const bool tr = t == r; yt = tr ? 0 : yt; ut = tr ? ur : ut; }   // :212
		int xn, vn, un, yn; unsigned nib = 0;
		narrow_cell<RIGHT>(F.Z[k], xp[k], vp[k], ut, yt, E.M24, E.q24, xn, vn, un, yn, nib);
		const unsigned d = narrow_p_byte(nib);
		const bool act = t <= en;                                // t >= st always
		F.X[k] = act ? xn : F.X[k]; F.V[k] = act ? vn : F.V[k]; F.U[k] = act ? un : F.U[k]; F.Y[k] = act ? yn : F.Y[k];
		if (act) pr[t] = (uint8_t)d;                             // :283
		const bool inT = ge0 && t <= en0;
		const bool sp = has_spec && t == en0;
		int h;
		if (r) h = (sp ? Hsp : F.H[k]) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329 (u8, v8 are uint8_t: :193)
		else h = (int)((unsigned)vn >> 24) - E.qe - E.qe;        // :349
		hk[k] = inT ? h : INTMIN;
		F.H[k] = inT ? h : F.H[k];
	}
	// ---- exact max (:320-348) ----------------------------------------------------------
	int hm = hk[0];
#pragma unroll
	for (int k = 1; k < NS; ++k) hm = hk[k] > hm ? hk[k] : hm;
	// as in ksw_narrow.h: the reduction is needed only when the diagonal beats the running maximum or may have fallen
	// more than zdrop below it
	bool quiet = ballot(hm > F.ez_max) == 0ull;
	if (quiet && E.zdrop >= 0) quiet = ballot(hm >= F.ez_max - E.zdrop) != 0ull;
	const int max_H = quiet ? 0 : wave_max_i32_keep(hm);
	// ---- ez updates (:351-357) -----------------------------------------------------------
	{
		int Hen0 = 0;
		if (en0 == E.tlen - 1 || r - st0 == E.qlen - 1) {
			Hen0 = ring_pick<NS>(hk, en0);
			const int Hst0 = ring_pick<NS>(hk, st0);
			if (en0 == E.tlen - 1 && Hen0 > F.mte) { F.mte = Hen0; F.mte_q = r - en; }        // rounded en (:352)
			if (r - st0 == E.qlen - 1 && Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st0; }
		}
		if (r == E.qlen + E.tlen - 2 && en0 == E.tlen - 1) F.score = Hen0;                  // :356-357
	}
	// ksw_apply_zdrop (:88-104) only looks at max_t when the maximum improves or has fallen more than zdrop below it
	if (quiet) return false;
	const bool improves = max_H > F.ez_max;
	if (!improves && (E.zdrop < 0 || F.ez_max - max_H <= E.zdrop)) return false;
	int max_t = en0;
	{
		// tie order of :320-348: en0 first, then the vector part [st0, st0+nv) by stride class (i & 3) and lowest i,
		// then the scalar tail ascending.  key = class << 16 | order, smallest wins.
		const int nv = (en0 - st0) / 4 * 4;
		unsigned key = 0xffffffffu;
#pragma unroll
		for (int k = 0; k < NS; ++k) {
			const int i = F.T[k] - st0;
			const unsigned kk = F.T[k] == en0 ? 0u : i < nv ? ((unsigned)((i & 3) + 1) << 16 | (unsigned)(i >> 2)) : (5u << 16 | (unsigned)(i - nv));
			key = (hk[k] == max_H && kk < key) ? kk : key;
		}
		key = wave_min_u32(key);
		const unsigned cls = key >> 16, ord = key & 0xffffu;
		if (cls == 0) max_t = en0;
		else if (cls <= 4) max_t = st0 + (int)ord * 4 + (int)cls - 1;
		else max_t = st0 + nv + (int)ord;
	}
	const int t = max_t, dq = r - max_t;
	if (improves) { F.ez_max = max_H; F.ez_max_t = t; F.ez_max_q = dq; return false; }
	if (t < F.ez_max_t || dq < F.ez_max_q) return false;
	const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
	const int l = tl > ql ? tl - ql : ql - tl;
	return F.ez_max - max_H > E.zdrop + l * E.e;
}

// Returns false when the job is not for this sweep (a code outside the 5-letter alphabet; nothing useful in `out`).
// Precondition: ksw_wide_ok<NS>(P, qlen, tlen).
template <int NS, bool RIGHT>
__device__ inline bool ksw_wave_wide(const uint8_t *query, int qlen, const uint8_t *target, int tlen,
                                     const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap,
                                     KswOut &out)
{
	const int lane = lane_id();
	int w = P.w;
	const int q = P.q, e = P.e, qe = q + e, flag = P.flag;
	out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;   // :81-86
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = 0;
	if (qlen <= 0 || tlen <= 0) return true;             // :147
	if (-P.min_sc > 2 * (q + e)) return true;            // :171
	if (w < 0) w = tlen > qlen ? tlen : qlen;            // :161
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	const int TP = (tlen + 15) / 16 * 16 + 64 * NS + 16, QR = (qlen + 15) / 16 * 16 + 64 * NS + 32;
	uint2 *tbl = (uint2 *)lds;                           // 5 entries, 64 bytes reserved
	uint8_t *tg = lds + 64;                              // target codes, zero padded (sf of :175,:188)
	unsigned *qs = (unsigned *)(tg + TP) + 16;           // selector words of the reversed query, padded with code 0 on both sides (:187)
	const unsigned ZW = (unsigned)(2 * qe) & 0xff, ZM = (unsigned)(2 * qe + P.sc_mch) & 0xff, ZX = (unsigned)(2 * qe + P.sc_mis) & 0xff;
	WSYNC();                                             // the previous job's LDS reads are done
	if (lane < 5) {
		uint2 t;
		if (lane == 4) { t.x = ZW * 0x01010100u; t.y = ZW * 0x0101u; }     // :219-226 wildcard target
		else {
			t.x = (lane == 0 ? ZM : ZX) << 8 | (lane == 1 ? ZM : ZX) << 16 | (lane == 2 ? ZM : ZX) << 24;
			t.y = (lane == 3 ? ZM : ZX) | ZW << 8;
		}
		tbl[lane] = t;
	}
	bool bad = false;                                    // a code outside the alphabet: not for this sweep
	for (int i = lane; i < TP; i += 64) {
		uint8_t b = 0;
		if (i < tlen) { b = target[i]; if (P.encode_ascii) b = enc_base(b); }
		bad |= b > 4;
		tg[i] = b;
	}
	if (lane < 16) qs[lane - 16] = 1u << 24 | 0x000c0c0cu;
	for (int i = lane; i < QR; i += 64) {
		unsigned b = 0;
		if (i < qlen) { b = query[qlen - 1 - i]; if (P.encode_ascii) b = enc_base((uint8_t)b); }
		bad |= b > 4;
		qs[i] = (b + 1) << 24 | 0x000c0c0cu;
	}
	if (ballot(bad)) return false;
	WSYNC();

	WideState<NS> F;
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		F.X[k] = F.V[k] = F.U[k] = F.Y[k] = 0; F.Z[k] = (int)(ZW << 24); F.H[k] = KSW_NEG_INF; F.T[k] = 64 * k + lane;
		const uint2 ta = tbl[tg[64 * k + lane]];
		F.T1[k] = ta.x; F.T0[k] = ta.y;
	}
	F.st = 0; F.edge_h = KSW_NEG_INF;
	F.ez_max = 0; F.ez_max_t = F.ez_max_q = -1; F.mqe = F.mte = F.score = KSW_NEG_INF; F.mqe_t = F.mte_q = -1;
	NarrowEnv E;
	E.tg = tg; E.qs = qs; E.tbl = tbl; E.p = nullptr; E.pb = p; E.qlen = qlen; E.tlen = tlen; E.w = w; E.ncol = ncol; E.qe = qe; E.e = e;
	E.zdrop = P.zdrop; E.ZW24 = (int)(ZW << 24); E.M24 = ZM << 24; E.q24 = (int)(((unsigned)q & 0xff) << 24);
	const int total = qlen + tlen - 1;
	bool stop = false;
	for (int r = 0; r < total; ++r) if (wide_diag<NS, RIGHT>(F, E, r)) { stop = true; break; }
	WSYNC();
	out.max = F.ez_max; out.zdropped = stop ? 1 : 0; out.max_q = F.ez_max_q; out.max_t = F.ez_max_t;
	out.mqe = F.mqe; out.mqe_t = F.mqe_t; out.mte = F.mte; out.mte_q = F.mte_q; out.score = stop ? KSW_NEG_INF : F.score;   // (the reference tests the z-drop before it takes the score of the last diagonal, :355-357: a sweep that stopped has none)
	ksw_backtrack_wave(p, ncol, qlen, tlen, w, flag, stop ? 1 : 0, F.ez_max_t, F.ez_max_q, cig_tmp, cig_cap, out);
	return true;
}

}  // namespace ihp
