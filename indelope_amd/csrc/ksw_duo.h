// ksw_duo.h -- ONE read against TWO targets in one sweep: the two unbanded alignments of an alignment-fallback item
// (src/indelope.nim:336-344: the read against the reference window and against the contig, both from the read's start).
// Reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388 with w = -1, zdrop = -1 (src/ksw2/ksw2.nim:159).
//
// ksw_wide.h runs such an alignment on the layout of the banded sweeps (lanes <-> target positions of a moving window,
// every artefact of the reference's 16-byte blocks reproduced: stale scores, rounded origins, the ring of registers that
// follows the band) and needs ~100 VALU instructions per 64 cells.  Without a band and without a z-drop none of that can
// reach a result: the cells of the true range [st0, en0] of a diagonal read only cells of the true range of the previous
// one or the explicit boundaries (:207-212) -- the 16-rounded margins feed margins --, the exact maximum is only the FIRST
// diagonal that reaches the final maximum, and no diagonal can stop the sweep.  So this sweep is the plain recurrence,
// laid out for the two things the item's alignments share:
//   * lanes <-> QUERY positions (lane i of slot k holds read base 64 k + i; three slots for reads up to 192, five up to 320).  Nothing moves: on diagonal r a
//     lane works on target position t = r - i, its x / v of the previous diagonal are its own (cell (t-1, i)), its u / y
//     come from the lane below (cell (t, i-1): one DPP rotation each), its score row is a constant of the lane and the
//     target base arrives as a selector word read from LDS at r - i.
//   * both targets in one register: alignment 0 in bits 15..8, alignment 1 in bits 31..24 of every value (ksw_pair.h's
//     packed cell); one v_perm_b32 looks up both scores, one byte per cell pair holds both traceback nibbles.
//   * a target shorter than the other is continued with wildcards (score 0, :224): those cells are a true alignment matrix
//     of the longer string, every step into them costs >= 0, so none of them can be the first to reach a maximum and none
//     is on a traceback path that starts inside the real matrix.  The same holds past the longer target's end, so no lane
//     ever needs an "am I still inside" test; lanes that have not started (i > r) keep their boundary state by a select on
//     the one slot that is filling up.
//   * H per lane in the u form (H(t, i) = H(t-1, i) + u8 - (q+e), the reference's own formula for the top cell, :318),
//     packed 16 bit; every lane keeps, per alignment, the key (its largest H << 16 | 0xffff - the diagonal that first
//     reached it) with one signed max; the reference's ez.max / max_t / max_q (:88-104, :312-349) are resolved from those once, after the
//     last diagonal, with the reference's tie order inside the diagonal.
// What it returns is what the fallback reads: max, max_t, max_q and the CIGAR (mqe / mte / score are left at their initial
// values; they would need the end-of-target tests this layout does without).
#pragma once
#include "ksw_pair.h"
#include "ksw_wide.h"

namespace ihp {

constexpr int DUO_NS_MAX = 5;                     // slots of 64 query positions: the sweep is built for 3 (reads up to 192) and 5 (up to 320)
constexpr int DUO_PAD = 64 * DUO_NS_MAX;

__host__ __device__ __forceinline__ int duo_slots(int qlen) { return (qlen + 63) >> 6; }
// selector words for t in [-DUO_PAD, tmax + DUO_PAD)
__host__ __device__ __forceinline__ size_t ksw_duo_lds_bytes(int tmax) { return 4 * ((size_t)tmax + 2 * DUO_PAD + 64); }
// one byte per (diagonal, query position)
__host__ __device__ __forceinline__ size_t ksw_duo_p_bytes(int qlen, int tmax) { return (size_t)(qlen + tmax) * 64 * duo_slots(qlen) + 64; }

__host__ __device__ __forceinline__ bool ksw_duo_ok(const KswParams &P, int qlen, int tl0, int tl1)
{
	const int tmax = tl0 > tl1 ? tl0 : tl1;
	const int qe = P.q + P.e, qe2 = 2 * qe;
	const int zm = (int)(signed char)((qe2 + P.sc_mch) & 0xff), zx = (int)(signed char)((qe2 + P.sc_mis) & 0xff);
	const int zw = (int)(signed char)(qe2 & 0xff);
	if (!(P.m == 5 && zm > 0 && zx > 0 && zw > 0 && P.sc_mch > 0 && P.sc_mis <= 0 && qe > 0 && qe < 64)) return false;
	if (P.flag & (KSW_EZ_SCORE_ONLY | KSW_EZ_RIGHT | KSW_EZ_GENERIC_SC | KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP)) return false;
	if (P.zdrop >= 0 || -P.min_sc > qe2) return false;
	if (qlen < 1 || qlen > 64 * DUO_NS_MAX || tl0 < 1 || tl1 < 1 || qlen + tmax > 0xffff) return false;
	if (P.w >= 0 && P.w < (qlen > tmax ? qlen : tmax)) return false;       // the band must never cut the matrix
	int big = qe > P.sc_mch ? qe : P.sc_mch;
	big = big > -P.sc_mis ? big : -P.sc_mis;
	return (long long)(qlen + tmax + 4) * big < 32000;                        // H stays inside 16 bits
}

struct DuoResult { int max[2], max_t[2], max_q[2]; };

// lane l gets v[l-1], lane 0 gets v[63] (every lane is written: no value to start from, no v_mov in front of it)
__device__ __forceinline__ unsigned duo_ror1(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x13C, 0xf, 0xf, false); }

template <int NS>
struct DuoState {
	unsigned X[NS], V[NS], U[NS], Y[NS], H[NS];
	int B0[NS], B1[NS];                                                 // per alignment: the lane's maximum of H so far << 16 | 0xffff - the diagonal it was first reached on
	unsigned T0[NS];                                                        // z by target code 0..3 for the lane's read base
};

// Slot K of diagonal r.  MASKED: the slot is still filling up -- lanes above `started` keep x, v, H at their boundary values.
// tag: 0xffff - r in every lane (a register: with the mask of the select it would be a second scalar operand).
template <int NS, int K, bool MASKED>
__device__ __forceinline__ void duo_slot(DuoState<NS> &S, const PairEnv &E, const unsigned T1, const unsigned QEp, const unsigned sel,
                                         const unsigned uin, const unsigned yin, const unsigned tag, const unsigned long long started, uint8_t *prow)
{
	const unsigned z = pair_z(S.T0[K], T1, sel);
	unsigned xn, vn, un, yn, nib;
	pair_cell(z, S.X[K], S.V[K], uin, yin, E, xn, vn, un, yn, nib);
	unsigned h = pk_sub(pk_add(S.H[K], pk_shr<8>(un)), QEp);                  // :318 for every cell
	if (MASKED) {
		const bool on = lane_in(started);
		xn = on ? xn : S.X[K]; vn = on ? vn : S.V[K]; h = on ? h : S.H[K];
	}
	S.X[K] = xn; S.V[K] = vn; S.U[K] = un; S.Y[K] = yn; S.H[K] = h;
	// a larger H wins, then the earlier diagonal: one signed max per alignment
	const int k0 = (int)((h << 16) | tag), k1 = (int)__builtin_amdgcn_perm(h, tag, 0x07060100u);   // H1 : tag
	S.B0[K] = k0 > S.B0[K] ? k0 : S.B0[K];
	S.B1[K] = k1 > S.B1[K] ? k1 : S.B1[K];
	prow[64 * K] = (uint8_t)(nib | (nib >> 12));                                // :283: alignment 0 in the low nibble, 1 in the high one
}

// ez.max / max_t / max_q of alignment K from the lanes' maxima (:88-104 with zdrop < 0; tie order of :320-348)
template <int NS, int K>
__device__ inline void duo_resolve(const DuoState<NS> &S, int qlen, int tlen, DuoResult &R)
{
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	int bv[NS], br[NS];
	int m = INTMIN;
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		const int key = K == 0 ? S.B0[k] : S.B1[k];
		bv[k] = 64 * k + lane < qlen ? key >> 16 : INTMIN;
		br[k] = 0xffff - (key & 0xffff);
		m = bv[k] > m ? bv[k] : m;
	}
	const int M = wave_max_i32_keep(m);
	R.max[K] = 0; R.max_t[K] = R.max_q[K] = -1;                                 // :81-86
	if (M <= 0) return;
	unsigned rm = 0xffffffffu;
#pragma unroll
	for (int k = 0; k < NS; ++k) rm = (bv[k] == M && (unsigned)br[k] < rm) ? (unsigned)br[k] : rm;
	const int rs = (int)wave_min_u32(rm);
	const int st0 = rs - qlen + 1 > 0 ? rs - qlen + 1 : 0, en0 = rs < tlen - 1 ? rs : tlen - 1;
	const int nv = (en0 - st0) / 4 * 4;
	unsigned key = 0xffffffffu;
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		const int t = rs - (64 * k + lane), i = t - st0;
		const bool cand = bv[k] == M && br[k] == rs && t >= st0 && t <= en0;
		const unsigned kk = t == en0 ? 0u : i < nv ? ((unsigned)((i & 3) + 1) << 16 | (unsigned)(i >> 2)) : (5u << 16 | (unsigned)(i - nv));
		key = (cand && kk < key) ? kk : key;
	}
	key = wave_min_u32(key);
	const unsigned cls = key >> 16, ord = key & 0xffffu;
	int max_t;
	if (cls == 0) max_t = en0;
	else if (cls <= 4) max_t = st0 + (int)ord * 4 + (int)cls - 1;
	else max_t = st0 + nv + (int)ord;
	R.max[K] = M; R.max_t[K] = max_t; R.max_q[K] = rs - max_t;
}

// The slots of one diagonal, lowest first: slot K takes u / y of the lane below -- lane 0 from the slot below's lane 63
// (`ru`, `ry`: the rotations of the PREVIOUS diagonal's values, all taken before any slot is overwritten), slot 0's lane 0 the
// boundary.  Slots above `hi` have not started; slot `hi` is the one filling up.
// LO (compile time): the slots below it have been left behind (see ksw_duo_sweep) and are not computed.
template <int NS, int K, int LO>
struct DuoSlots {
	static __device__ __forceinline__ void run(DuoState<NS> &S, const PairEnv &E, const unsigned T1, const unsigned QEp, const unsigned *sp, const int r,
	                                           const unsigned (&ru)[NS], const unsigned (&ry)[NS], const unsigned ub, const bool l0, const int hi,
	                                           const unsigned tag, const unsigned long long started, uint8_t *prow)
	{
		if (K < LO) { DuoSlots<NS, K + 1, LO>::run(S, E, T1, QEp, sp, r, ru, ry, ub, l0, hi, tag, started, prow); return; }
		if (K <= hi) {
			const unsigned uin = l0 ? (K ? ru[K ? K - 1 : 0] : ub) : ru[K], yin = l0 ? (K ? ry[K ? K - 1 : 0] : 0u) : ry[K];
			const unsigned sel = sp[r + 64 * (NS - 1 - K)];
			if (K < NS - 1 && K < hi) duo_slot<NS, K, false>(S, E, T1, QEp, sel, uin, yin, tag, started, prow);
			else duo_slot<NS, K, true>(S, E, T1, QEp, sel, uin, yin, tag, started, prow);
			DuoSlots<NS, K + 1, LO>::run(S, E, T1, QEp, sp, r, ru, ry, ub, l0, hi, tag, started, prow);
		}
	}
};
template <int NS, int LO>
struct DuoSlots<NS, NS, LO> {
	static __device__ __forceinline__ void run(DuoState<NS> &, const PairEnv &, const unsigned, const unsigned, const unsigned *, const int,
	                                           const unsigned (&)[NS], const unsigned (&)[NS], const unsigned, const bool, const int,
	                                           const unsigned, const unsigned long long, uint8_t *) {}
};

// Diagonals [r0, r1) with the slots below LO left out: one loop per LO, so that every loop is the straight-line chain of its
// slots (a run-time `lo` inside ONE loop cost the kernel 80 spilled registers and a third more instructions).  Behind the first
// diagonal of a run with LO > 0 -- the last to read slot LO - 1's real lane 63 -- that slot's u / y become the floor values.
template <int NS, int LO>
__device__ __forceinline__ void duo_run(DuoState<NS> &S, const PairEnv &E, const unsigned T1, const unsigned QEp, const unsigned *sp, const int r0, const int r1,
                                        const int nsl, const int ncol, const bool l0, const unsigned ub1, unsigned &ub, unsigned &tag, uint8_t *&prow)
{
	for (int r = r0; r < r1; ++r) {
		const int hi = (r >> 6) < nsl - 1 ? (r >> 6) : nsl - 1;
		const int fill = r - 64 * hi;
		const unsigned long long started = fill >= 63 ? ~0ull : ~0ull >> (63 - fill);
		unsigned ru[NS], ry[NS];
#pragma unroll
		for (int k = 0; k < NS; ++k) { ru[k] = duo_ror1(S.U[k]); ry[k] = duo_ror1(S.Y[k]); }
		DuoSlots<NS, 0, LO>::run(S, E, T1, QEp, sp, r, ru, ry, ub, l0, hi, tag, started, prow);
		if (LO > 0 && r == r0) { S.U[LO > 0 ? LO - 1 : 0] = 0; S.Y[LO > 0 ? LO - 1 : 0] = 0; }
		ub = ub1;
		tag -= 1;
		prow += ncol;
	}
}
template <int NS, int LO>
struct DuoPhases {                                                             // the runs with LO, LO + 1, ... slots left behind, one after the other
	static __device__ __forceinline__ void run(DuoState<NS> &S, const PairEnv &E, const unsigned T1, const unsigned QEp, const unsigned *sp, const int tmax, const int total,
	                                           const int nsl, const int ncol, const bool l0, const unsigned ub1, unsigned &ub, unsigned &tag, uint8_t *&prow)
	{
		// slot K is past the longer target's end from diagonal tmax + 64 K + 63 on: LO slots are behind from tmax + 64 (LO - 1) + 63
		const int r0 = tmax + 64 * LO - 1, r1 = tmax + 64 * LO + 63;
		if (r0 >= total) return;
		duo_run<NS, LO>(S, E, T1, QEp, sp, r0, r1 < total ? r1 : total, nsl, ncol, l0, ub1, ub, tag, prow);
		DuoPhases<NS, LO + 1>::run(S, E, T1, QEp, sp, tmax, total, nsl, ncol, l0, ub1, ub, tag, prow);
	}
};
template <int NS>
struct DuoPhases<NS, NS> {
	static __device__ __forceinline__ void run(DuoState<NS> &, const PairEnv &, const unsigned, const unsigned, const unsigned *, const int, const int,
	                                           const int, const int, const bool, const unsigned, unsigned &, unsigned &, uint8_t *&) {}
};

// Returns false when the item is not for this sweep (a code outside the 5-letter alphabet).  Preconditions: ksw_duo_ok(),
// qlen <= 64 NS.  lds: ksw_duo_lds_bytes(max(tl0, tl1)); p: ksw_duo_p_bytes(qlen, max(tl0, tl1)).
template <int NS>
__device__ inline bool ksw_duo_sweep(const uint8_t *query, int qlen, const uint8_t *t0, int tl0, const uint8_t *t1, int tl1,
                                     const KswParams &P, uint8_t *lds, uint8_t *p, DuoResult &R, const bool skip_done = true)
{
	const int lane = lane_id();
	qlen = uni(qlen); tl0 = uni(tl0); tl1 = uni(tl1);
	const int q = uni(P.q), e = uni(P.e), qe = q + e;
	const int tmax = tl0 > tl1 ? tl0 : tl1;
	const int nsl = duo_slots(qlen), ncol = 64 * nsl;
	unsigned *selw = (unsigned *)lds;
	WSYNC();                                                                    // the previous item's LDS reads are done
	bool bad = false;
	const int nsel = tmax + 2 * DUO_PAD;
	for (int j = lane; j < nsel; j += 64) {
		const int t = j - DUO_PAD;
		unsigned c0 = 4, c1 = 4;
		if (t >= 0 && t < tl0) { c0 = t0[t]; if (P.encode_ascii) c0 = enc_base((uint8_t)c0); }
		if (t >= 0 && t < tl1) { c1 = t1[t]; if (P.encode_ascii) c1 = enc_base((uint8_t)c1); }
		bad |= c0 > 4 || c1 > 4;
		selw[j] = 0x000c000cu | (c0 & 7) << 8 | (c1 & 7) << 24;
	}
	const unsigned ZW = (unsigned)(2 * qe) & 0xff, ZM = (unsigned)(2 * qe + P.sc_mch) & 0xff, ZX = (unsigned)(2 * qe + P.sc_mis) & 0xff;
	PairEnv E;
	E.Qp = ((unsigned)q & 0xff) * 0x01000100u; E.Mp = ZM * 0x01000100u;
	E.zx4 = ZX * 0x01010101u; E.zdm = ZM - ZX; E.zw4 = ZW * 0x01010101u;
	const unsigned T1 = E.zw4, QEp = (unsigned)qe * 0x00010001u;
	DuoState<NS> S;
#pragma unroll
	for (int k = 0; k < NS; ++k) {
		const int i = 64 * k + lane;
		unsigned c = 4;
		if (i < qlen) { c = query[i]; if (P.encode_ascii) c = enc_base((uint8_t)c); }
		bad |= c > 4;
		S.T0[k] = pair_table(E, c);
		S.X[k] = 0; S.V[k] = i ? E.Qp : 0u;                                     // x1 = 0, v1 = r ? q : 0 (:211)
		S.U[k] = S.Y[k] = 0;
		S.H[k] = ((unsigned)(-(qe + e * i)) & 0xffffu) * 0x00010001u;         // H(-1, i)
		S.B0[k] = S.B1[k] = 0;                                                  // ez.max starts at 0 (:81): only a positive H reports
	}
	if (ballot(bad)) return false;
	WSYNC();
	const int total = qlen + tmax - 1;
	// lane's selector word of diagonal r: selw[DUO_PAD + r - i]; slot k at a fixed offset below slot 0
	const unsigned *sp = selw + DUO_PAD - 64 * (NS - 1) - lane;
	uint8_t *prow = p + lane;
	unsigned ub = 0;                                                            // u of the cell entering at the top: r ? q : 0 (:212), lane 0 only
	const unsigned ub1 = lane == 0 ? E.Qp : 0u;
	const bool l0 = lane == 0;
	unsigned tag = 0xffffu;
	if (skip_done) {
		// Slot K is past the LONGER target's end from diagonal tmax + 64 K + 63 on (its lane 63 works on t = r - 64 K - 63): those
		// cells are the wildcard continuation, which no maximum and no traceback path can come from (see above), and the last
		// cell of the slot that feeds a real one -- (tmax - 1, 64 K + 63), the neighbour of slot K + 1's lane 0 -- was computed
		// on diagonal tmax + 64 K + 62.  The slots left behind are not computed (round 6: 7 % of a 150 x 460 item's slot-diagonals).
		// What such a slot still hands to the one above -- its lane 63's u / y -- is set to the LOWEST values once their last real
		// use is over (duo_run): the wildcard cells of the slot above are then an alignment matrix over a neighbour column that
		// falls by q + e per cell with no gap open -- never above the true continuation (u, y >= 0 there) --, so their H stays at
		// or below what the full sweep computes for them, which no maximum comes from.  (Frozen values would do as cell INPUTS --
		// nothing real reads them -- but the lanes' running maxima see every cell a lane computes.)
		const int e0 = tmax + 63 < total ? tmax + 63 : total;
		duo_run<NS, 0>(S, E, T1, QEp, sp, 0, e0, nsl, ncol, l0, ub1, ub, tag, prow);
		DuoPhases<NS, 1>::run(S, E, T1, QEp, sp, tmax, total, nsl, ncol, l0, ub1, ub, tag, prow);
	} else duo_run<NS, 0>(S, E, T1, QEp, sp, 0, total, nsl, ncol, l0, ub1, ub, tag, prow);
	WSYNC();
	duo_resolve<NS, 0>(S, qlen, tl0, R);
	duo_resolve<NS, 1>(S, qlen, tl1, R);
	return true;
}

// The CIGAR of alignment K (ksw_backtrack_wave on the shared byte matrix).
template <int K>
__device__ __forceinline__ void ksw_duo_cigar(const DuoResult &R, const uint8_t *p, int qlen, int tlen, int w, int flag,
                                              uint32_t *cig_tmp, int cig_cap, KswOut &out)
{
	out.max = R.max[K]; out.zdropped = 0; out.max_t = R.max_t[K]; out.max_q = R.max_q[K];
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.mqe_t = out.mte_q = -1; out.n_cigar = 0;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	ksw_backtrack_wave<3, K>(p, 64 * duo_slots(qlen), qlen, tlen, w, flag, 0, R.max_t[K], R.max_q[K], cig_tmp, cig_cap, out);
}

}  // namespace ihp
