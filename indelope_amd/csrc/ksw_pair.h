// ksw_pair.h -- TWO alignments per wavefront for the production ksw2 sweep (round 4).  Reference:
// src/ksw2/csrc/ksw2_extz2_sse.c:113-388; the single sweep this is derived from is ksw_narrow.h.
//
// k_ksw is bound by instruction issue, and every VALU instruction of its cell works on ONE int8 value per 32-bit lane
// (the SSE reference: sixteen per instruction, :262-284).  tools/ubench_ksw.hip prices the alternatives on the hardware:
// in a mixed stream every VALU instruction costs the same (~1 per cycle and CU, whatever its class), so what counts is
// instructions per (alignment, diagonal) -- 31 in ksw_narrow.h.  gfx950 has no packed byte arithmetic, but it has packed
// 16-bit adds, subtracts, unsigned and signed max / min (v_pk_*_u16 / _i16): with value << 8 in each half of a register
// they ARE _mm_add_epi8 / _mm_sub_epi8 / _mm_max_epu8 / _mm_min_epu8 / the signed compare against zero for two cells at
// once, exactly as the top byte of a 32-bit register is for one.
//
// Two alignments can share a wavefront when they share the CONTROL of the sweep.  The band of diagonal r is
// st0 = max(0, r-qlen+1, (r-w+1)>>1), en0 = min(tlen-1, r, (r+w)>>1) (:196-203).  With tlen > qlen + w the target never
// limits it: en0 reaches qlen+w-1 at most and the band leaves the matrix on diagonal 2 qlen + w - 1 (:200-203, the
// reference's usual exit for a contig against its window: the window is at least 63 bases longer than the contig,
// indelope.nim:218-220) -- so for such jobs EVERY lane mask, block rotation and loop bound depends on (r, qlen, w) only.
// k_ksw_plan (kernels.h) sorts the jobs of a batch by qlen and pairs equal ones; a pair runs the whole sweep in lock step,
// alignment 0 in bits 15..8 and alignment 1 in bits 31..24 of every u / v / x / y / z register, all scalar work shared.
// Jobs without a partner, with a short window, a wildcard in the contig, a band outside 49..62 or a length whose scores
// could leave 16 bits take the single sweep.
//
//   * score lookup: ONE v_perm_b32 for both alignments -- S1 holds alignment 0's z by query code for this lane's target
//     base, S0 alignment 1's; the selector word (0x0c | c0 << 8 | 0x0c << 16 | (4 + c1) << 24) of a query position is
//     shared by the pair (equal qlen) and comes from LDS.
//   * H is kept as a packed 16-bit G = H + r (q+e) per alignment.  Inside the band G never decreases (H[t] changes by
//     v8 - (q+e), v8 unsigned, :323-329; a cell entering at the top takes H[t-1] + u8, :318) and never exceeds
//     ez.max + r (q+e), so it stays within [-2(q+e), qlen*match + (2 qlen + w)(q+e)]: the plan kernel only pairs jobs for
//     which that is below 32000.  Lanes outside the band hold anything; every test of G is masked by the band's lanes.
//   * exact maximum and z-drop by compares (as ksw_narrow.h): v_cmp_*_i16_sdwa picks a half; the thresholds are scalars.
//   * traceback: the four compare results of a cell (:265/:273/:277-282) are the signs of two saturating differences and
//     "x' != 0", "y' != 0"; they are gathered into a nibble per half with packed shifts and multiply-adds (no v_cmp, no
//     SGPR round trip) and shifted into one accumulator: four diagonals fill the 16 bits of a half, one dword store per
//     lane every FOURTH diagonal into slot (r >> 2) + (st >> 4) of 80 dwords -- the same bytes per alignment as the single
//     sweep's eight diagonals per dword; ksw_backtrack_wave<2> reads its alignment's half.
#pragma once
#include "ksw_narrow.h"

namespace ihp {

#define IHP_PK2(NAME, OP) __device__ __forceinline__ unsigned NAME(unsigned a, unsigned b) { unsigned r; asm(OP " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
IHP_PK2(pk_add, "v_pk_add_u16")
IHP_PK2(pk_sub, "v_pk_sub_u16")
IHP_PK2(pk_maxu, "v_pk_max_u16")
IHP_PK2(pk_minu, "v_pk_min_u16")
#undef IHP_PK2
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b) { unsigned r; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_max0(unsigned a) { unsigned r; asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ unsigned pk_add_s(unsigned a, unsigned s) { unsigned r; asm("v_pk_add_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ unsigned pk_sub_s(unsigned a, unsigned s) { unsigned r; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ unsigned pk_minu_s(unsigned a, unsigned s) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
template <int N> __device__ __forceinline__ unsigned pk_shr(unsigned a) { unsigned r; asm("v_pk_lshrrev_b16 %0, %2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "n"(N)); return r; }
template <int K> __device__ __forceinline__ unsigned pk_minc(unsigned a) { unsigned r; asm("v_pk_min_u16 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "n"(K)); return r; }
// a * K + c per half
template <int K> __device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned c) { unsigned r; asm("v_pk_mad_u16 %0, %1, %3, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c), "n"(K)); return r; }
// both halves shifted left by the wave-uniform n (n in both halves of the scalar)
__device__ __forceinline__ unsigned pk_shl_s(unsigned a, unsigned n2) { unsigned r; asm("v_pk_lshlrev_b16 %0, %2, %1" : "=v"(r) : "v"(a), "s"(n2)); return r; }

// lanes whose half K of v is >  / >= the wave-uniform 16-bit t
template <int K> __device__ __forceinline__ unsigned long long pk_gt(unsigned v, int t)
{
	unsigned long long m;
	if (K == 0) asm("v_cmp_gt_i16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(v), "s"(t));
	else asm("v_cmp_gt_i16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0" : "=s"(m) : "v"(v), "s"(t));
	return m;
}
template <int K> __device__ __forceinline__ unsigned long long pk_ge(unsigned v, int t)
{
	unsigned long long m;
	if (K == 0) asm("v_cmp_ge_i16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(v), "s"(t));
	else asm("v_cmp_ge_i16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_0" : "=s"(m) : "v"(v), "s"(t));
	return m;
}
// half K of a packed word, sign extended (scalar and per lane)
template <int K> __device__ __forceinline__ int pk_half(int s) { return K == 0 ? (int)(short)(s & 0xffff) : s >> 16; }

// bytes of LDS / of traceback scratch a pair needs (qlen shared, the two window lengths)
__host__ __device__ __forceinline__ size_t ksw_pair_lds_bytes(int qlen, int tlen0, int tlen1)
{
	return 64 + (size_t)((tlen0 + 15) / 16) * 16 + 96 + (size_t)((tlen1 + 15) / 16) * 16 + 96 + 4 * ((size_t)((qlen + 15) / 16) * 16 + 96 + 16);
}
__host__ __device__ __forceinline__ size_t ksw_pair_p_bytes(int qlen, int w)
{
	return ((size_t)((2 * qlen + w) >> 2) + (size_t)((qlen + w) >> 4) + 3) * 320;
}
// what a job costs towards a pair's LDS (two such jobs fit when their sum does)
__host__ __device__ __forceinline__ size_t ksw_pair_lds_share(int qlen, int tlen)
{
	return 32 + (size_t)((tlen + 15) / 16) * 16 + 96 + 2 * ((size_t)((qlen + 15) / 16) * 16 + 96 + 16);
}

// Parameters under which the pair sweep stands in for ksw_wave_narrow<false> (the plan kernel adds the per-job tests).
__host__ __device__ __forceinline__ bool ksw_pair_ok(const KswParams &P)
{
	return ksw_narrow_ok(P) && P.w >= 49 && P.w <= 62 && !(P.flag & KSW_EZ_RIGHT) && P.sc_mch > 0 && P.q + P.e > 0 && P.q + P.e < 64;
}
// A job that may be half of a pair: the steady sweep has room (qlen, tlen >= w + 32), the window never cuts the band
// (tlen > qlen + w) and G = H + r (q+e) stays inside 16 bits.
__host__ __device__ __forceinline__ bool ksw_pair_job_ok(const KswParams &P, int qlen, int tlen)
{
	return qlen >= P.w + 32 && tlen >= qlen + P.w + 1 &&
	       (long long)qlen * P.sc_mch + (long long)(2 * qlen + P.w + 2) * (P.q + P.e) < 32000;
}

struct PairEnv {
	const uint8_t *tg0, *tg1;                            // LDS: target codes of the two windows (zero padded)
	const unsigned *qs;                                  // LDS: selector words of the two reversed queries
	unsigned *p;                                         // traceback slots of 80 dwords
	int qlen, w, qe, e;
	int zd;                                              // z-drop, or a value no score difference reaches
	unsigned Qp, Mp, ZWp, QE2p;                          // both halves: q << 8, max_sc << 8, z of a never-refreshed cell << 8; 2(q+e)
	unsigned zx4, zdm;                                   // z(mismatch) in four bytes, z(match) - z(mismatch)
	unsigned zw4;                                        // z(wildcard) in four bytes
};

struct PairState {
	unsigned XA, VA, UA, YA, ZA, GA;                     // slot A: t = st + lane
	unsigned XB, VB, UB, YB, GB;                         // slot B: t = st + 64 + lane, lanes 0..15
	unsigned TA0, TA1, TB0, TB1;                         // z by query code for the lane's target base: alignment 0 / 1, slot A / B
	const unsigned *qptr; int qoffB; int rlB;
	unsigned accA, accB;                                 // traceback nibbles, the latest diagonal lowest in each half
	// wave-uniform
	int st; int edge_g;
	int thr0, thr1;                                      // ez.max + r (q+e) of the coming diagonal
	int max_t0, max_t1, max_q0, max_q1, mqe0, mqe1, mqe_t0, mqe_t1;
	int dead0, dead1;                                    // z-dropped: the values above are final, ez_max* holds ez.max
	int ez_max0, ez_max1;
};

// z by query code (bytes 0..3) for a target base
__device__ __forceinline__ unsigned pair_table(const PairEnv &E, unsigned code)
{
	return code < 4 ? E.zx4 + (E.zdm << (8 * code)) : E.zw4;
}

__device__ __forceinline__ unsigned pair_z(unsigned T0, unsigned T1, unsigned sel) { return __builtin_amdgcn_perm(T1, T0, sel); }

// One cell of both alignments (:116-137 + :262-284, left-aligned); every value is (int8 << 8) per half; z > 0 (ksw_narrow_ok).
// nib: the four compare results of ksw_narrow.h's nibble, first compare highest, in bits 3..0 of each half.
__device__ __forceinline__ void pair_cell(unsigned z, unsigned xp, unsigned vp, unsigned u, unsigned y, const PairEnv &E,
                                          unsigned &xn, unsigned &vn, unsigned &un, unsigned &yn, unsigned &nib)
{
	const unsigned a = pk_add(xp, vp), b = pk_add(y, u);
	const unsigned s1 = pk_sub_sat(z, a);                               // :265  negative <=> a > z (signed)
	const unsigned zz1 = pk_maxu(z, a);                                 // :272  _mm_max_epu8
	const unsigned s2 = pk_sub_sat(zz1, b);                             // :273  negative <=> b > z (signed)
	unsigned zz = pk_maxu(zz1, b);                                      // :131
	zz = pk_minu_s(zz, E.Mp);                                           // :132
	un = pk_sub(zz, vp); vn = pk_sub(zz, u);                            // :133-134
	const unsigned zq = pk_sub_s(zz, E.Qp);
	const unsigned a2 = pk_sub(a, zq), b2 = pk_sub(b, zq);
	xn = pk_max0(a2); yn = pk_max0(b2);                                 // :277-280
	// a2 > 0 <=> x' != 0 (then x' >= 0x100)
	const unsigned c12 = pk_mad<2>(pk_shr<15>(s1), pk_shr<15>(s2));
	const unsigned c34 = pk_add(pk_minc<2>(xn), pk_minc<1>(yn));
	nib = pk_mad<4>(c12, c34);
}

// Close the traceback slot of diagonals ..r_last (band origin st): a full group when (r_last & 3) == 3.
__device__ __forceinline__ void pair_flush(PairState &S, const PairEnv &E, int r_last, int st)
{
	const int lane = lane_id();
	unsigned *row = E.p + (size_t)((r_last >> 2) + (st >> 4)) * 80;
	const unsigned sh = (unsigned)(4 * (3 - (r_last & 3))) * 0x00010001u;
	row[lane] = pk_shl_s(S.accA, sh);
	if (lane < 16) row[64 + lane] = pk_shl_s(S.accB, sh);
}

// The exact maximum (:312-349) and ksw_apply_zdrop (:88-104) of alignment K on diagonal r, decided by lane compares where
// that is enough (see ksw_narrow.h).  inTA / mInB: the lanes of the true band in slot A / B.
template <int K>
__device__ __forceinline__ void pair_ez(PairState &S, const PairEnv &E, const int r, const int st0, const int en0, const bool hasB,
                                        const unsigned long long inTA, const unsigned long long mInB)
{
	const int INTMIN = -0x7fffffff - 1;
	const int st = S.st, loA = st0 - st;
	int &thr = K == 0 ? S.thr0 : S.thr1;
	int &emt = K == 0 ? S.max_t0 : S.max_t1;
	int &emq = K == 0 ? S.max_q0 : S.max_q1;
	const unsigned long long mA = pk_gt<K>(S.GA, thr) & inTA, mB = hasB ? pk_gt<K>(S.GB, thr) & mInB : 0ull;
	if (mA | mB) {
		int gmax, max_t;
		if (popc64(mA) + popc64(mB) == 1) {
			const int i = mA ? ctz64(mA) : ctz64(mB);
			gmax = pk_half<K>(mA ? __builtin_amdgcn_readlane((int)S.GA, i) : __builtin_amdgcn_readlane((int)S.GB, i));
			max_t = st + i + (mA ? 0 : 64);
		} else {
			const int hAm = lane_in(inTA) ? pk_half<K>((int)S.GA) : INTMIN, hBm = (hasB && lane_in(mInB)) ? pk_half<K>((int)S.GB) : INTMIN;
			gmax = wave_max_i32_keep(hAm);
			if (hasB) { const int mb = wave_max_i32_keep(hBm); gmax = mb > gmax ? mb : gmax; }
			max_t = narrow_max_t(hAm, hBm, gmax, hasB, mInB, loA, st, st0, en0);
		}
		thr = gmax; emt = max_t; emq = r - max_t;
		return;
	}
	int thz = thr - E.zd;
	thz = thz < -32768 ? -32768 : thz;
	if ((pk_ge<K>(S.GA, thz) & inTA) | (hasB ? pk_ge<K>(S.GB, thz) & mInB : 0ull)) return;   // ez.max - max_H <= zdrop: :98 cannot hold
	const int hAm = lane_in(inTA) ? pk_half<K>((int)S.GA) : INTMIN, hBm = (hasB && lane_in(mInB)) ? pk_half<K>((int)S.GB) : INTMIN;
	int gmax = wave_max_i32_keep(hAm);
	if (hasB) { const int mb = wave_max_i32_keep(hBm); gmax = mb > gmax ? mb : gmax; }
	const int t = narrow_max_t(hAm, hBm, gmax, hasB, mInB, loA, st, st0, en0), dq = r - t;
	if (t < emt || dq < emq) return;
	const int tl = t - emt, ql = dq - emq;
	const int l = tl > ql ? tl - ql : ql - tl;
	if (thr - gmax > E.zd + l * E.e) {                                  // z-drop: this alignment is done; its partner goes on
		if (K == 0) { S.dead0 = 1; S.ez_max0 = thr - r * E.qe; } else { S.dead1 = 1; S.ez_max1 = thr - r * E.qe; }
	}
}

// The band origin moves one block right in front of diagonal r: close the traceback slot, rotate the registers 16 lanes,
// re-seed slot B.  ex, ev: the block edge x[st-1], v[st-1] lane 0 takes on this diagonal (:207-208).
__device__ __forceinline__ void pair_move(PairState &S, const PairEnv &E, const int r, const int nst, unsigned &ex, unsigned &ev)
{
	const int lane = lane_id();
	if (r & 3) pair_flush(S, E, r - 1, S.st);
	ex = (unsigned)__builtin_amdgcn_readlane((int)S.XA, 15);
	ev = (unsigned)__builtin_amdgcn_readlane((int)S.VA, 15);
	S.edge_g = __builtin_amdgcn_readlane((int)S.GA, 15);
	const unsigned zB = S.rlB < 0 ? E.ZWp : pair_z(S.TB0, S.TB1, E.qs[E.qlen - 1 - S.rlB + S.st + 64 + (lane & 15)]);
	S.XA = rot16(S.XA, S.XB, lane); S.VA = rot16(S.VA, S.VB, lane);
	S.UA = rot16(S.UA, S.UB, lane); S.YA = rot16(S.YA, S.YB, lane);
	S.ZA = rot16(S.ZA, zB, lane);
	S.GA = rot16(S.GA, S.GB, lane);
	S.st = nst;
	S.qptr += 16;
	S.XB = S.VB = S.UB = S.YB = 0; S.GB = 0;
	S.rlB = -1;
	S.TA0 = pair_table(E, E.tg0[nst + lane]); S.TA1 = pair_table(E, E.tg1[nst + lane]);
	S.TB0 = pair_table(E, E.tg0[nst + 64 + (lane & 15)]); S.TB1 = pair_table(E, E.tg1[nst + 64 + (lane & 15)]);
}

// One anti-diagonal of both alignments, any r (the general form: ksw_narrow.h's narrow_diag<ND_ANY> on packed halves, in
// the G = H + r (q+e) form).  st0, en0: the true band of r.  FIRST: r == 0.
template <bool FIRST>
__device__ __forceinline__ void pair_diag(PairState &S, const PairEnv &E, const int r, const int st0, const int en0)
{
	const int lane = lane_id();
	const int nst = st0 & ~15, en = en0 | 15;
	unsigned ex = 0, ev = 0;
	if (!FIRST && nst != S.st) pair_move(S, E, r, nst, ex, ev);
	else if (S.st == 0) ev = r ? E.Qp : 0u;                             // :211
	// neighbours of r-1 (taken before anything is overwritten); lane 0 gets the block edge x1, v1 (:207-211)
	const unsigned xpA = (unsigned)set_lane0((int)ex, 0, dppz_shr1((int)S.XA)), vpA = (unsigned)set_lane0((int)ev, 0, dppz_shr1((int)S.VA));
	const unsigned GpA = (unsigned)set_lane0(S.edge_g, 0, dppz_shr1((int)S.GA));
	const int st = S.st;
	const int loA = st0 - st, hiT = en0 - st, nTop = en - st;
	const int sc = st0 + ((en0 - st0) / 16 + 1) * 16 - 1 - st;           // last refreshed score lane (:215)
	const bool hasB = nTop >= 64;
	const unsigned long long refA = lane_range(loA, sc < 63 ? sc : 63), inTA = lane_range(loA, hiT < 63 ? hiT : 63);
	const unsigned long long spA = (!FIRST && hiT < 64) ? 1ull << hiT : 0ull;   // H[en0] comes from H[en0-1] + u (:318)
	const unsigned long long actA = ~0ull >> (63 - (nTop < 63 ? nTop : 63));
	unsigned long long mInB = 0;
	S.rlB = lane_in(sc >= 64 ? ~0ull >> (127 - sc) : 0ull) ? r : S.rlB;  // :214-228 runs past en; value formed on use
	// ---- slot B (block 4) ------------------------------------------------------------
	if (hasB) {
		const unsigned exB = (unsigned)__builtin_amdgcn_readlane((int)S.XA, 63), evB = (unsigned)__builtin_amdgcn_readlane((int)S.VA, 63);
		const int geB = __builtin_amdgcn_readlane((int)S.GA, 63);
		const unsigned xpB = (unsigned)set_lane0((int)exB, 0, dppz_shr1((int)S.XB)), vpB = (unsigned)set_lane0((int)evB, 0, dppz_shr1((int)S.VB));
		const unsigned GpB = (unsigned)set_lane0(geB, 0, dppz_shr1((int)S.GB));
		const unsigned zB = S.rlB < 0 ? E.ZWp : pair_z(S.TB0, S.TB1, E.qs[E.qlen - 1 - S.rlB + st + 64 + (lane & 15)]);
		mInB = lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15);
		if (lane < 16) {
			unsigned ut = S.UB, yt = S.YB;
			if (st + 64 + lane == r) { yt = 0; ut = r ? E.Qp : 0u; }     // :212
			unsigned xn, vn, un, yn, nib;
			pair_cell(zB, xpB, vpB, ut, yt, E, xn, vn, un, yn, nib);
			S.XB = xn; S.VB = vn; S.UB = un; S.YB = yn;
			S.accB = pk_mad<16>(S.accB, nib);                              // :283
			const bool sp = !FIRST && 64 + lane == hiT;
			S.GB = pk_add(sp ? GpB : S.GB, pk_shr<8>(sp ? un : vn));       // :318, :323-329 (u8, v8 are uint8_t: :193)
		}
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	{
		const unsigned znew = pair_z(S.TA0, S.TA1, *S.qptr);            // qs[qlen-1-r+st+lane]
		S.qptr -= 1;
		S.ZA = lane_in(refA) ? znew : S.ZA;                                 // :214-228
		if (r <= en && r - st < 64) {                                      // :212 (only while the band still touches t == r)
			const bool tr = lane_in(1ull << (r - st));
			S.YA = tr ? 0u : S.YA; S.UA = tr ? (r ? E.Qp : 0u) : S.UA;
		}
		unsigned xn, vn, un, yn, nib;
		pair_cell(S.ZA, xpA, vpA, S.UA, S.YA, E, xn, vn, un, yn, nib);
		S.accA = pk_mad<16>(S.accA, nib);                                  // :283 (lanes past nTop: never read)
		unsigned g;
		if (!FIRST) { const bool sp = lane_in(spA); g = pk_add(sp ? GpA : S.GA, pk_shr<8>(sp ? un : vn)); }   // :318, :323-329
		else g = pk_sub_s(pk_shr<8>(vn), E.QE2p);                          // :349
		const bool act = lane_in(actA);
		S.XA = act ? xn : S.XA; S.VA = act ? vn : S.VA; S.UA = act ? un : S.UA; S.YA = act ? yn : S.YA;
		S.GA = g;
	}
	if ((r & 3) == 3) pair_flush(S, E, r, st);
	// ---- ez updates (:351-357): the window never cuts the band, so only H[st0] at the end of the query -------
	if (r - st0 == E.qlen - 1) {
		const int g = __builtin_amdgcn_readlane((int)S.GA, loA);
		const int h0 = pk_half<0>(g) - r * E.qe, h1 = pk_half<1>(g) - r * E.qe;
		if (!S.dead0 && h0 > S.mqe0) { S.mqe0 = h0; S.mqe_t0 = st0; }
		if (!S.dead1 && h1 > S.mqe1) { S.mqe1 = h1; S.mqe_t1 = st0; }
	}
	if (!S.dead0) { pair_ez<0>(S, E, r, st0, en0, hasB, inTA, mInB); S.thr0 += E.qe; }
	if (!S.dead1) { pair_ez<1>(S, E, r, st0, en0, hasB, inTA, mInB); S.thr1 += E.qe; }
}

// Two jobs with the same qlen and parameters, both ksw_pair_job_ok().  Returns false when a sequence holds a code the
// pair sweep does not take (a wildcard or worse in a query, anything above the wildcard in a target): nothing useful in
// the outputs then -- the plan kernel only pairs jobs whose producer vouches for the codes.  Otherwise both results
// are final (fields and CIGAR of out0 first: cig_tmp is used twice, `emit(k, out)` is called behind each traceback).
template <class Emit>
__device__ inline bool ksw_wave_pair(const uint8_t *q0, const uint8_t *t0, int tlen0, const uint8_t *q1, const uint8_t *t1, int tlen1, int qlen,
                                     const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap, Emit emit, long long *pacc = nullptr)
{
	const long long tc0 = pacc ? (long long)clock64() : 0;
	const int lane = lane_id();
	const int w = P.w, q = P.q, e = P.e, qe = q + e;
	const int TP0 = (tlen0 + 15) / 16 * 16 + 96, TP1 = (tlen1 + 15) / 16 * 16 + 96, QR = (qlen + 15) / 16 * 16 + 96;
	uint8_t *tg0 = lds + 64, *tg1 = tg0 + TP0;
	unsigned *qs = (unsigned *)(tg1 + TP1) + 16;
	bool bad = false;
	for (int i = lane; i < TP0; i += 64) {
		uint8_t b = 0;
		if (i < tlen0) { b = t0[i]; if (P.encode_ascii) b = enc_base(b); }
		bad |= b > 4;
		tg0[i] = b;
	}
	for (int i = lane; i < TP1; i += 64) {
		uint8_t b = 0;
		if (i < tlen1) { b = t1[i]; if (P.encode_ascii) b = enc_base(b); }
		bad |= b > 4;
		tg1[i] = b;
	}
	if (lane < 16) qs[lane - 16] = 0x040c000cu;
	for (int i = lane; i < QR; i += 64) {
		unsigned c0 = 0, c1 = 0;
		if (i < qlen) {
			c0 = q0[qlen - 1 - i]; c1 = q1[qlen - 1 - i];
			if (P.encode_ascii) { c0 = enc_base((uint8_t)c0); c1 = enc_base((uint8_t)c1); }
		}
		bad |= c0 > 3 || c1 > 3;
		qs[i] = 0x000c000cu | c0 << 8 | (4 + c1) << 24;
	}
	if (ballot(bad)) return false;
	WSYNC();
	const long long tc1 = pacc ? (long long)clock64() : 0;

	const unsigned ZW = (unsigned)(2 * qe) & 0xff, ZM = (unsigned)(2 * qe + P.sc_mch) & 0xff, ZX = (unsigned)(2 * qe + P.sc_mis) & 0xff;
	PairEnv E;
	E.tg0 = tg0; E.tg1 = tg1; E.qs = qs; E.p = (unsigned *)p; E.qlen = qlen; E.w = w; E.qe = qe; E.e = e;
	E.zd = P.zdrop < 0 ? 0x3fffffff : P.zdrop;
	E.Qp = ((unsigned)q & 0xff) * 0x01000100u; E.Mp = ZM * 0x01000100u; E.ZWp = ZW * 0x01000100u; E.QE2p = (unsigned)(2 * qe) * 0x00010001u;
	E.zx4 = ZX * 0x01010101u; E.zdm = ZM - ZX; E.zw4 = ZW * 0x01010101u;
	PairState S;
	S.XA = S.VA = S.UA = S.YA = 0; S.ZA = E.ZWp; S.GA = 0;
	S.XB = S.VB = S.UB = S.YB = 0; S.GB = 0;
	S.TA0 = pair_table(E, tg0[lane]); S.TA1 = pair_table(E, tg1[lane]);
	S.TB0 = pair_table(E, tg0[64 + (lane & 15)]); S.TB1 = pair_table(E, tg1[64 + (lane & 15)]);
	S.qptr = qs + (qlen - 1 + lane); S.qoffB = 64 + (lane & 15) - lane; S.rlB = -1;
	S.accA = S.accB = 0; S.st = 0; S.edge_g = 0;
	S.thr0 = S.thr1 = 0; S.max_t0 = S.max_t1 = S.max_q0 = S.max_q1 = -1;
	S.mqe0 = S.mqe1 = KSW_NEG_INF; S.mqe_t0 = S.mqe_t1 = -1;
	S.dead0 = S.dead1 = 0; S.ez_max0 = S.ez_max1 = 0;
	// the band leaves the matrix on diagonal 2 qlen + w - 1 (:200-203): r_end is the first diagonal without a cell
	const int r_end = 2 * qlen + w - 1;
	pair_diag<true>(S, E, 0, 0, 0);
	int r = 1;
	for (; r < r_end; ++r) {
		int st0 = r - qlen + 1, en0 = (r + w) >> 1;
		const int sw = (r - w + 1) >> 1;
		st0 = st0 > sw ? st0 : sw; st0 = st0 > 0 ? st0 : 0;
		en0 = en0 < r ? en0 : r;
		pair_diag<false>(S, E, r, st0, en0);
		if (S.dead0 & S.dead1) { ++r; break; }
	}
	// r - 1 is the last diagonal whose cells were computed; thr* stand at diagonal r
	if (((r - 1) & 3) != 3) pair_flush(S, E, r - 1, S.st);
	WSYNC();
	const long long tc2 = pacc ? (long long)clock64() : 0;
	if (pacc && lane == 0) { pacc[0] += tc1 - tc0; pacc[1] += tc2 - tc1; pacc[3] += 2; }
	for (int k = 0; k < 2; ++k) {
		KswOut out;
		// every exit is a z-drop for the caller (:98-101, :200-203); a sweep that stopped has no score (:355-357) and the window's end is never reached
		out.zdropped = 1; out.mte = out.score = KSW_NEG_INF; out.mte_q = -1; out.n_cigar = 0;
		if (k == 0) { out.max = S.dead0 ? S.ez_max0 : S.thr0 - r * qe; out.max_t = S.max_t0; out.max_q = S.max_q0; out.mqe = S.mqe0; out.mqe_t = S.mqe_t0; }
		else { out.max = S.dead1 ? S.ez_max1 : S.thr1 - r * qe; out.max_t = S.max_t1; out.max_q = S.max_q1; out.mqe = S.mqe1; out.mqe_t = S.mqe_t1; }
		const long long tb0 = pacc ? (long long)clock64() : 0;
		if (k == 0) ksw_backtrack_wave<2, 0>(p, 0, qlen, tlen0, w, P.flag, 1, out.max_t, out.max_q, cig_tmp, cig_cap, out);
		else ksw_backtrack_wave<2, 1>(p, 0, qlen, tlen1, w, P.flag, 1, out.max_t, out.max_q, cig_tmp, cig_cap, out);
		if (pacc && lane == 0) pacc[2] += (long long)clock64() - tb0;
		emit(k, out);
	}
	return true;
}

}  // namespace ihp
