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

// Packed 16-bit arithmetic on two alignments at once.  Written on clang's vector types, not as inline assembly: the compiler
// then knows the instructions (v_pk_add_u16, v_pk_sub_i16 clamp, v_pk_max_u16 ...), schedules them and needs no wait
// state between two of them -- behind an asm statement whose result the next one reads it places an s_nop (nine per diagonal).
typedef unsigned short pk_us2 __attribute__((ext_vector_type(2)));
typedef short pk_ss2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_us2 pk_u(unsigned x) { return __builtin_bit_cast(pk_us2, x); }
__device__ __forceinline__ pk_ss2 pk_s(unsigned x) { return __builtin_bit_cast(pk_ss2, x); }
__device__ __forceinline__ unsigned pk_r(pk_us2 x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ unsigned pk_r(pk_ss2 x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return pk_r((pk_us2)(pk_u(a) + pk_u(b))); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return pk_r((pk_us2)(pk_u(a) - pk_u(b))); }
__device__ __forceinline__ unsigned pk_maxu(unsigned a, unsigned b) { return pk_r(__builtin_elementwise_max(pk_u(a), pk_u(b))); }
__device__ __forceinline__ unsigned pk_minu(unsigned a, unsigned b) { return pk_r(__builtin_elementwise_min(pk_u(a), pk_u(b))); }
// the saturating signed difference: negative <=> a < b as int16 (never wraps)
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b) { return pk_r(__builtin_elementwise_sub_sat(pk_s(a), pk_s(b))); }
__device__ __forceinline__ unsigned pk_max0(unsigned a) { return pk_r(__builtin_elementwise_max(pk_s(a), (pk_ss2){0, 0})); }
__device__ __forceinline__ unsigned pk_add_s(unsigned a, unsigned s) { return pk_add(a, s); }
__device__ __forceinline__ unsigned pk_sub_s(unsigned a, unsigned s) { return pk_sub(a, s); }
__device__ __forceinline__ unsigned pk_minu_s(unsigned a, unsigned s) { return pk_minu(a, s); }
template <int N> __device__ __forceinline__ unsigned pk_shr(unsigned a) { return pk_r((pk_us2)(pk_u(a) >> (pk_us2){N, N})); }
template <int K> __device__ __forceinline__ unsigned pk_minc(unsigned a) { return pk_r(__builtin_elementwise_min(pk_u(a), (pk_us2){K, K})); }
// a * K + c per half
template <int K> __device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned c) { return pk_r((pk_us2)(pk_u(a) * (pk_us2){K, K} + pk_u(c))); }
// both halves shifted left by the wave-uniform n (n in both halves of the scalar)
__device__ __forceinline__ unsigned pk_shl_s(unsigned a, unsigned n2) { return pk_r((pk_us2)(pk_u(a) << pk_u(n2))); }

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
__device__ __forceinline__ int pos_pack(int r, int t) { return (int)(((unsigned)r << 16) | ((unsigned)t & 0xffffu)); }
__device__ __forceinline__ int pos_t(int pos) { return (int)(short)(pos & 0xffff); }
__device__ __forceinline__ int pos_q(int pos) { return (pos >> 16) - (int)(short)(pos & 0xffff); }

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
	int qlm1;                                            // qlen - 1
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
	int pos0, pos1;                                      // where ez.max was found: the diagonal << 16 | max_t (max_q = diagonal - max_t)
	int mqe0, mqe1;
	int cold;                                            // lanes 0..3: mqe_t0, mqe_t1, fin0, fin1 -- written a few times per sweep and read at its end; in a
	                                                     // register of their own they cost the loops four scalar registers and the compiler spilled hotter ones
	int inc0, inc1;                                      // q+e while the alignment is live; 0 once it has z-dropped: the values above are final,
	                                                     // (fin* holds ez.max of an alignment that has z-dropped; thr* then stands at 32767)
};

extern "C" __device__ int ihp_writelane_i32(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");   // (as in ksw_wide.h)
constexpr int PC_MQE_T0 = 0, PC_MQE_T1 = 1, PC_FIN0 = 2, PC_FIN1 = 3;
__device__ __forceinline__ void pair_cold_set(PairState &S, int k, int v) { S.cold = ihp_writelane_i32(v, k, S.cold); }
__device__ __forceinline__ int pair_cold_get(const PairState &S, int k) { return __builtin_amdgcn_readlane(S.cold, k); }

// z by query code (bytes 0..3) for a target base
__device__ __forceinline__ unsigned pair_table(const PairEnv &E, unsigned code)
{
	return code < 4 ? E.zx4 + (E.zdm << (8 * code)) : E.zw4;
}

__device__ __forceinline__ unsigned pair_z(unsigned T0, unsigned T1, unsigned sel) { return __builtin_amdgcn_perm(T1, T0, sel); }

// One cell of both alignments (:116-137 + :262-284, left-aligned); every value is (int8 << 8) per half; z > 0 (ksw_narrow_ok).
// nib: the four compare results behind the reference's traceback byte in bits 3..0 of each half (see below for the order).
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
	// a2 > 0 <=> x' != 0 (then x' >= 0x100): bit 1 = :265, bit 0 = :273, bit 2 = a2 > 0, bit 3 = b2 > 0
	const unsigned c12 = pk_mad<2>(pk_shr<15>(s1), pk_shr<15>(s2));
	nib = c12 + pk_minc<4>(xn) + pk_minc<8>(yn);                        // (each half stays below 16: one 32-bit v_add3 adds both)
}

// Close the traceback slot of diagonals ..r_last (band origin st): a full group when (r_last & 3) == 3.
__device__ __forceinline__ void pair_flush(PairState &S, const PairEnv &E, int r_last, int st)
{
	const int lane = lane_id();
	unsigned *row = E.p + (size_t)((r_last >> 2) + (st >> 4)) * 80;
	const unsigned sh = (unsigned)(4 * (3 - (r_last & 3))) * 0x00010001u;
	row[lane] = pk_shl_s(S.accA, sh);
	row[64 + lane] = pk_shl_s(S.accB, sh);              // (lanes 16..63 land in the next slot's first dwords, which its own store rewrites later)
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
	int &pos = K == 0 ? S.pos0 : S.pos1;
	const int emt = pos_t(pos), emq = pos_q(pos);
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
		thr = gmax; pos = pos_pack(r, max_t);
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
		if (K == 0) S.inc0 = 0; else S.inc1 = 0;
		pair_cold_set(S, K == 0 ? PC_FIN0 : PC_FIN1, thr - r * E.qe);
		thr = 0x7fff - E.qe;                                                // (the caller adds q+e once more)
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
	// (the selector is read whether or not it is needed: a load behind a per-lane test is a divergent branch)
	const unsigned zBf = pair_z(S.TB0, S.TB1, E.qs[E.qlen - 1 - (S.rlB < 0 ? 0 : S.rlB) + S.st + 64 + (lane & 15)]);
	const unsigned zB = S.rlB < 0 ? E.ZWp : zBf;
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
		const unsigned zBf = pair_z(S.TB0, S.TB1, E.qs[E.qlen - 1 - (S.rlB < 0 ? 0 : S.rlB) + st + 64 + (lane & 15)]);
		const unsigned zB = S.rlB < 0 ? E.ZWp : zBf;
		mInB = lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15);
		{   // (all 64 lanes; lanes 16..63 of the slot-B registers are never read)
			unsigned ut = S.UB, yt = S.YB;
			{ const bool tr = st + 64 + lane == r; yt = tr ? 0u : yt; ut = tr ? (r ? E.Qp : 0u) : ut; }   // :212
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
		if (S.inc0 && h0 > S.mqe0) { S.mqe0 = h0; pair_cold_set(S, PC_MQE_T0, st0); }
		if (S.inc1 && h1 > S.mqe1) { S.mqe1 = h1; pair_cold_set(S, PC_MQE_T1, st0); }
	}
	if (S.inc0) { pair_ez<0>(S, E, r, st0, en0, hasB, inTA, mInB); S.thr0 += E.qe; }
	if (S.inc1) { pair_ez<1>(S, E, r, st0, en0, hasB, inTA, mInB); S.thr1 += E.qe; }
}

// Tell the compiler again that the wave-uniform part of the state is uniform.  pair_diag has divergent branches (slot B's
// sixteen lanes, a lazily formed score behind a per-lane test); behind a loop over it the compiler's uniformity analysis
// gives up on the loop's exit and with it on every scalar that leaves the loop -- they would live in vector registers,
// and every branch on them would be an exec-mask branch.
__device__ __forceinline__ void pair_uniform(PairState &S)
{
	S.st = uni(S.st); S.edge_g = uni(S.edge_g); S.thr0 = uni(S.thr0); S.thr1 = uni(S.thr1); S.pos0 = uni(S.pos0); S.pos1 = uni(S.pos1);
	S.mqe0 = uni(S.mqe0); S.mqe1 = uni(S.mqe1);
	S.inc0 = uni(S.inc0); S.inc1 = uni(S.inc1);
}

// ---------------------------------------------------------------- the steady diagonals, laid out as ksw_narrow.h's
// What the lean loops carry from diagonal to diagonal besides PairState (wave-uniform).
struct PairCtl {
	int r, st0, en0;                                     // the coming diagonal and its true band
	int lim;                                             // the run's end; pulled below r once both alignments have z-dropped
	unsigned long long geLoM, spM, hiM;                  // lanes >= st0 - st; the lane of en0 (slot A); lanes <= en0 - st
	int zsafe0, zsafe1;                                  // the last diagonal on which the cell of the running maximum alone rules a z-drop out
};

// The exact maximum (:312-349) and ksw_apply_zdrop (:88-104) of alignment K on a diagonal of the lean loops, and the step of
// its threshold to the next diagonal.  As pair_ez, with two differences that keep the scalar unit's share small (the kernel
// is bound by SALU issue as much as by VALU issue: tools/r4 profiles):
//  * a finished alignment needs no test of its own: its threshold stands at 32767, where no G reaches, its `inc` is 0 and
//    its zsafe is INT_MAX;
//  * a z-drop is ruled out without looking while the cell t* of the running maximum vouches for it.  t* held G = thr when
//    it was found; while it stays inside the band below the top cell its G cannot fall (H[t] changes by v8 - (q+e), v8
//    unsigned, :323-329), and the z-drop threshold is thr + (r - r*) (q+e) - zdrop: for (r - r*) <= zq = zdrop / (q+e)
//    diagonals -- and while t* is in the band: r <= 2 t* + w and r <= t* + qlen - 1 -- that cell alone keeps max_H within
//    zdrop of ez.max and :98 cannot hold.  (Not when t* was the top cell en0: H[en0] is formed anew from H[en0-1] + u8 on
//    every diagonal, :318.)
// inTA / mInB: the lanes of the true band in slot A / B; topA: the lane of en0 when it is in slot A, else 0.
template <int K, bool HASB>
__device__ __forceinline__ void pair_ez_lean(PairState &S, const PairEnv &E, PairCtl &C, const int r, const unsigned long long inTA, const unsigned long long mInB,
                                             const unsigned long long topA, const int zq)
{
	const int INTMIN = -0x7fffffff - 1;
	const int st = S.st;
	int &thr = K == 0 ? S.thr0 : S.thr1;
	int &pos = K == 0 ? S.pos0 : S.pos1;
	int &zsafe = K == 0 ? C.zsafe0 : C.zsafe1;
	const int inc = K == 0 ? S.inc0 : S.inc1;
	const unsigned long long mA = pk_gt<K>(S.GA, thr) & inTA, mB = HASB ? pk_gt<K>(S.GB, thr) & mInB : 0ull;
	if (mA | mB) {
		int gmax, max_t; bool top;
		if (!HASB ? !(mA & (mA - 1)) : popc64(mA) + popc64(mB) == 1) {
			if (!HASB || mA) {
				const int i = ctz64(mA);
				gmax = pk_half<K>(__builtin_amdgcn_readlane((int)S.GA, i)); max_t = st + i; top = (mA & topA) != 0;
			} else {
				const int i = ctz64(mB);
				gmax = pk_half<K>(__builtin_amdgcn_readlane((int)S.GB, i)); max_t = st + 64 + i; top = max_t == C.en0;
			}
		} else {
			const int st0 = C.st0, en0 = C.en0;
			const int hAm = lane_in(inTA) ? pk_half<K>((int)S.GA) : INTMIN, hBm = (HASB && lane_in(mInB)) ? pk_half<K>((int)S.GB) : INTMIN;
			gmax = wave_max_i32_keep(hAm);
			if (HASB) { const int mb = wave_max_i32_keep(hBm); gmax = mb > gmax ? mb : gmax; }
			max_t = narrow_max_t(hAm, hBm, gmax, HASB, mInB, st0 - st, st, st0, en0);
			top = max_t == en0;
		}
		pos = pos_pack(r, max_t);
		int zs = r + zq;
		const int b1 = 2 * max_t + E.w, b2 = max_t + E.qlen - 1;
		zs = zs < b1 ? zs : b1; zs = zs < b2 ? zs : b2;
		zsafe = uni(top ? r : zs);                                           // (or the compiler forms the minimum on the vector unit and keeps zsafe there)
		thr = uni(gmax + inc); pos = uni(pos);
		return;
	}
	if (r <= zsafe) { thr += inc; return; }
	{
		const int st0 = C.st0, en0 = C.en0;
		int thz = thr - E.zd;
		thz = thz < -32768 ? -32768 : thz;
		if (!((pk_ge<K>(S.GA, thz) & inTA) | (HASB ? pk_ge<K>(S.GB, thz) & mInB : 0ull))) {   // else ez.max - max_H <= zdrop: :98 cannot hold
			const int hAm = lane_in(inTA) ? pk_half<K>((int)S.GA) : INTMIN, hBm = (HASB && lane_in(mInB)) ? pk_half<K>((int)S.GB) : INTMIN;
			int gmax = wave_max_i32_keep(hAm);
			if (HASB) { const int mb = wave_max_i32_keep(hBm); gmax = mb > gmax ? mb : gmax; }
			const int t = narrow_max_t(hAm, hBm, gmax, HASB, mInB, st0 - st, st, st0, en0), dq = r - t;
			const int emt = pos_t(pos), emq = pos_q(pos);
			if (t >= emt && dq >= emq) {
				const int tl = t - emt, ql = dq - emq;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (thr - gmax > E.zd + l * E.e) {                          // z-drop: this alignment is done; its partner goes on
					const int fin = thr - r * E.qe;
					thr = 0x7fff; zsafe = 0x7fffffff;
					if (K == 0) S.inc0 = 0; else S.inc1 = 0;
					pair_cold_set(S, K == 0 ? PC_FIN0 : PC_FIN1, fin);
					if (!(S.inc0 | S.inc1)) C.lim = INTMIN;
					return;
				}
			}
		}
	}
	thr += inc;
}

// pair_ez_lean's everyday cases for a diagonal without block 4, on the scalar unit by hand: no lane of alignment K above its
// threshold and the cell of the running maximum still vouching against a z-drop (seven instructions), or exactly one lane
// above it (the new maximum: twenty).  Returns 0 when that was all; 1: several lanes improve, 2: the vouching has run out --
// pair_ez_rest does those, nothing has been changed then.  Written as one asm statement because the compiler kept max_t /
// max_q in spill lanes of a vector register (v_writelane / v_readlane around every new maximum) and formed the minimum of
// three scalars on the vector unit; everything here stays in the SGPRs the operands name.
#define IHP_PAIR_EZ_ASM(WORD, SEXT, ENTRY, RARE2, RARE1, STATC)                                                                                    \
	asm("v_cmp_gt_i16_sdwa %[m], %[G], %[thr] src0_sel:" WORD " src1_sel:WORD_0\n\t"                               \
	    ENTRY                                                                                                      \
	    "s_and_b64 %[m], %[m], %[inT]\n\t"                                                                         \
	    "s_cbranch_scc1 1f\n\t"                                                                                    \
	    "s_cmp_le_i32 %[r], %[zs]\n\t"                                                                             \
	    "s_cbranch_scc1 2f\n\t"                                                                                    \
	    RARE2 "\n\t"                                                                                 \
	    "s_branch 9f\n"                                                                                            \
	    "2:\n\t"                                                                                                   \
	    "s_add_i32 %[thr], %[thr], %[inc]\n\t"                                                                     \
	    "s_branch 9f\n"                                                                                            \
	    "1:\n\t"                                                                                                   \
	    "s_bcnt1_i32_b64 %[i], %[m]\n\t"                                                                           \
	    "s_cmp_lg_u32 %[i], 1\n\t"                                                                                 \
	    "s_cbranch_scc0 3f\n\t"                                                                                    \
	    RARE1 "\n\t"                                                                                 \
	    "s_branch 9f\n"                                                                                            \
	    "3:\n\t"                                                                                                   \
	    "s_ff1_i32_b64 %[i], %[m]\n\t"                                                                             \
	    "v_readlane_b32 %[g], %[G], %[i]\n\t"                                                                      \
	    "s_add_i32 %[i], %[i], %[st]\n\t"                                                                          \
	    "s_add_i32 %[zs], %[r], %[zq]\n\t"                                                                         \
	    "s_lshl1_add_u32 %[b], %[i], %[w]\n\t"                                                                     \
	    "s_min_i32 %[zs], %[zs], %[b]\n\t"                                                                         \
	    "s_add_i32 %[b], %[i], %[qlm1]\n\t"                                                                        \
	    "s_min_i32 %[zs], %[zs], %[b]\n\t"                                                                         \
	    "s_pack_ll_b32_b16 %[pos], %[i], %[r]\n\t"                                                                 \
	    SEXT "\n\t"                                                                                                \
	    "s_and_b64 %[m], %[m], %[sp]\n\t"                                                                          \
	    "s_cselect_b32 %[zs], %[r], %[zs]\n\t"                                                                     \
	    "s_add_i32 %[thr], %[g], %[inc]\n"                                                                         \
	    "9:"                                                                                                       \
	    : [m] "=&s"(m), [stat] STATC(stat), [i] "=&s"(i), [g] "=&s"(g), [b] "=&s"(b), [thr] "+s"(thr), [pos] "+s"(pos), [zs] "+s"(zsafe) \
	    : [G] "v"(G), [inT] "s"(inT), [sp] "s"(spM), [r] "s"(r), [st] "s"(st), [zq] "s"(zq), [w] "s"(w), [qlm1] "s"(qlm1), [inc] "s"(inc) \
	    : "scc")
// `stat` is shared by the two alignments of a diagonal: alignment 0 clears it and reports in bits 1..0, alignment 1 in bits 3..2
// (one test per diagonal instead of a clear and a test per alignment).
template <int K>
__device__ __forceinline__ void pair_ez_asm(int &stat, const unsigned G, int &thr, int &pos, int &zsafe, const unsigned long long inT_, const unsigned long long spM_,
                                            const int r_, const int st_, const int zq_, const int w_, const int qlm1_, const int inc_)
{
	const unsigned long long inT = (unsigned long long)uni((long long)inT_), spM = (unsigned long long)uni((long long)spM_);
	const int r = uni(r_), st = uni(st_), zq = uni(zq_), w = uni(w_), qlm1 = uni(qlm1_), inc = uni(inc_);
	unsigned long long m; int i, g, b;
	if (K == 0) IHP_PAIR_EZ_ASM("WORD_0", "s_sext_i32_i16 %[g], %[g]", "s_mov_b32 %[stat], 0\n\t", "s_mov_b32 %[stat], 2", "s_mov_b32 %[stat], 1", "=&s");
	else IHP_PAIR_EZ_ASM("WORD_1", "s_ashr_i32 %[g], %[g], 16", "", "s_or_b32 %[stat], %[stat], 8", "s_or_b32 %[stat], %[stat], 4", "+s");
}
#undef IHP_PAIR_EZ_ASM

// One steady diagonal of both alignments and the step to the next (narrow_steady_step of ksw_narrow.h on packed halves).
// HASB = block 4 is computed (hiT >= 64); EDGE = 1: the band origin moved on this diagonal and lane 0 takes the block edge
// ex, ev (:207-208); EDGE = 2: a diagonal 1 <= r <= w+30 -- the band still starts in block 0 (lane 0 takes x1 = 0, v1 = q,
// :211), grows from the single cell t = r, computes blocks 0 .. en0/16 only and, while r <= en, holds the boundary cell
// t = r (:212).  PAR = parity of r + w when the caller knows it (0: st0 grows on the step to r+1, 1: en0 does), -1 otherwise.
// FLUSH: 0 = r & 3 is known not to be 3 (no traceback store on this diagonal), 1 = it is 3 when r & 2, -1 = look.
template <bool HASB, int EDGE, int PAR, int FLUSH>
__device__ __forceinline__ void pair_steady_step(PairState &S, const PairEnv &E, PairCtl &C, const int zq, const unsigned ex = 0, const unsigned ev = 0)
{
	const int lane = lane_id();
	const int st = S.st, r = C.r;
	if (PAR < 0) {
		C.geLoM = ~0ull << (C.st0 - st);
		C.spM = HASB ? 0ull : 1ull << (C.en0 - st); C.hiM = HASB ? ~0ull : C.spM | (C.spM - 1);
	}
	unsigned xpA = (unsigned)dppz_shr1((int)S.XA), vpA = (unsigned)dppz_shr1((int)S.VA);    // neighbours of r-1
	if (EDGE == 1) { xpA = (unsigned)set_lane0((int)ex, 0, (int)xpA); vpA = (unsigned)set_lane0((int)ev, 0, (int)vpA); }
	if (EDGE == 2) vpA = (unsigned)set_lane0((int)E.Qp, 0, (int)vpA);
	const int en = C.en0 | 15;                           // EDGE == 2: last computed cell; t = r is computed while r <= en
	const unsigned GpA = (unsigned)dppz_shr1((int)S.GA);                                   // en0 is never on lane 0 here
	const unsigned long long geLoM = C.geLoM;
	unsigned long long mInB = 0;
	// ---- slot B (block 4) ------------------------------------------------------------
	if (HASB) {
		const unsigned exB = (unsigned)__builtin_amdgcn_readlane((int)S.XA, 63), evB = (unsigned)__builtin_amdgcn_readlane((int)S.VA, 63);
		const int geB = __builtin_amdgcn_readlane((int)S.GA, 63);
		const unsigned xpB = (unsigned)set_lane0((int)exB, 0, dppz_shr1((int)S.XB)), vpB = (unsigned)set_lane0((int)evB, 0, dppz_shr1((int)S.VB));
		const unsigned GpB = (unsigned)set_lane0(geB, 0, dppz_shr1((int)S.GB));
		const unsigned zf = pair_z(S.TB0, S.TB1, S.qptr[S.qoffB]);       // qs[qlen-1-r+st+64+lane]
		const unsigned zB = lane_in(~geLoM) ? zf : E.ZWp;                  // refreshed up to lane loA + 63: lanes 0..loA-1 here
		const int hiB = C.en0 - st - 64;
		mInB = lane_range(0, hiB < 15 ? hiB : 15);
		{   // (all 64 lanes: a wave instruction costs the same with 16 lanes active, and a divergent branch here costs the loop
			// its scalar control; lanes 16..63 of the slot-B registers are never read)
			unsigned ut = S.UB, yt = S.YB;
			if (EDGE == 2) { const bool tr = 64 + lane == r; yt = tr ? 0u : yt; ut = tr ? E.Qp : ut; }   // :212
			unsigned xn, vn, un, yn, nib;
			pair_cell(zB, xpB, vpB, ut, yt, E, xn, vn, un, yn, nib);
			S.XB = xn; S.VB = vn; S.UB = un; S.YB = yn;
			S.accB = pk_mad<16>(S.accB, nib);                              // :283
			const bool sp = lane == hiB;
			S.GB = pk_add(sp ? GpB : S.GB, pk_shr<8>(sp ? un : vn));       // :318, :323-329
		}
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	// (EDGE == 2 commits under a lane mask instead of branching around the cell: a divergent branch in the loop body made
	// the compiler treat the loop's whole control as divergent and keep every scalar of it in vector registers)
	S.qptr -= 1;
	{
		const unsigned long long actM = EDGE == 2 ? ~0ull >> (63 - (en < 63 ? en : 63)) : ~0ull;
		const unsigned znew = pair_z(S.TA0, S.TA1, S.qptr[1]);          // qs[qlen-1-r+st+lane]
		S.ZA = lane_in(geLoM & actM) ? znew : S.ZA;                        // :214-228: refreshed from st0 to st0 + 63 (or to en)
		if (EDGE == 2 && r <= en && r < 64) {                             // :212
			const bool tr = lane_in(1ull << r);
			S.YA = tr ? 0u : S.YA; S.UA = tr ? E.Qp : S.UA;
		}
		unsigned xn, vn, un, yn, nib;
		pair_cell(S.ZA, xpA, vpA, S.UA, S.YA, E, xn, vn, un, yn, nib);
		if (EDGE == 2) {
			const bool act = lane_in(actM);
			S.XA = act ? xn : S.XA; S.VA = act ? vn : S.VA; S.UA = act ? un : S.UA; S.YA = act ? yn : S.YA;
		} else { S.XA = xn; S.VA = vn; S.UA = un; S.YA = yn; }
		S.accA = pk_mad<16>(S.accA, nib);                                  // :283
		if (!HASB) {
			const bool sp = lane_in(C.spM);
			S.GA = pk_add(sp ? GpA : S.GA, pk_shr<8>(sp ? un : vn));       // :318, :323-329
		} else S.GA = pk_add(S.GA, pk_shr<8>(vn));                         // en0 is in block 4
	}
	if (FLUSH < 0 ? (r & 3) == 3 : FLUSH > 0 && (r & 2)) {                // pair_flush of a full group
		unsigned *row = E.p + (size_t)((r >> 2) + (st >> 4)) * 80;
		row[lane] = S.accA;
		row[64 + lane] = S.accB;                        // (lanes 16..63 land in the next slot's first dwords, which its own store rewrites later)
	}
	// ---- exact max (:312-349) and ksw_apply_zdrop (:88-104) ----------------------------
	const unsigned long long inTA = geLoM & C.hiM;
	if (HASB) {
		pair_ez_lean<0, true>(S, E, C, r, inTA, mInB, C.spM, zq);
		pair_ez_lean<1, true>(S, E, C, r, inTA, mInB, C.spM, zq);
	} else {
		int stat;
		pair_ez_asm<0>(stat, S.GA, S.thr0, S.pos0, C.zsafe0, inTA, C.spM, r, st, zq, E.w, E.qlm1, S.inc0);
		pair_ez_asm<1>(stat, S.GA, S.thr1, S.pos1, C.zsafe1, inTA, C.spM, r, st, zq, E.w, E.qlm1, S.inc1);
		if (stat) {                                                        // (an everyday diagonal: one test for both alignments)
			if (stat & 3) pair_ez_lean<0, false>(S, E, C, r, inTA, 0ull, C.spM, zq);
			if (stat >> 2) pair_ez_lean<1, false>(S, E, C, r, inTA, 0ull, C.spM, zq);
		}
	}
	// ---- the step to r + 1: (r+w)>>1 grows from an odd r+w, (r-w+1)>>1 otherwise ----
	if (EDGE == 2) {
		const int s1 = (r + 2 - E.w) >> 1, e1 = (r + 1 + E.w) >> 1;
		C.st0 = s1 > 0 ? s1 : 0; C.en0 = e1 < r + 1 ? e1 : r + 1;
	} else if (PAR == 0) { C.st0 += 1; C.geLoM <<= 1; }
	else if (PAR == 1) { C.en0 += 1; C.spM <<= 1; C.hiM = (C.hiM << 1) | 1ull; }
	else { const int up = (r + E.w) & 1; C.en0 += up; C.st0 += 1 - up; }
	C.r = r + 1;
}

// ---------------------------------------------------------------- round 5: the everyday steady run as ONE asm loop
// pair_steady_step<false, 0, 0, 0> + <false, 0, 1, 1> (no block 4, even w: a pair of diagonals starts on an even r, only the
// second can close a traceback group) until C.lim or until a diagonal needs the general ez code, written out by hand:
//  * the vector side is the 30 instructions of the cell + 2 compares (the compiler's: 37 + reloads of scalars it had spilled
//    into lanes of a vector register -- a v_readlane and its wait states inside the loop);
//  * the scalar side tests BOTH alignments with one branch: no lane of either above its threshold and both running maxima
//    still vouching (r <= min(zsafe0, zsafe1)) is seven instructions for the two of them; an alignment with exactly one lane
//    above its threshold (the new maximum) takes fourteen; one whose vouching has run out is cleared by a single compare
//    against the z-drop threshold where that is enough.  Several lanes at once or a possible z-drop leave the loop: `stat`
//    says which alignment needs pair_ez_lean (bits 1..0 / 3..2 as in pair_ez_asm) and whether the diagonal was the second of
//    its pair (bit 4); everything else of that diagonal (cells, traceback store, the other alignment) is done.
//  * inside the steady run t* leaves the band only through (r - w + 1) >> 1 (the end of the query is not in reach), so the
//    vouching limit is min(r + zq, 2 t* + w): pair_ez_asm's third term is gone; the caller's tail loop starts afresh (-1).
//  * nothing of the loop lives in a spilled scalar: what it needs are operands, the rest of the kernel's state is the
//    compiler's to park where it likes across the statement.
// Hazards (the assembler pads nothing inside an asm string): a DPP source is never written in the two instructions in front
// of it (every one of XA / VA / GA is written a dozen instructions earlier; nine scalar / LDS instructions open the
// statement), no VALU reads an SGPR that a VALU wrote (the compare masks go to the scalar unit, v_readlane's lane select
// comes from s_ff1), lgkmcnt counts only this loop's two LDS reads (the statement opens with lgkmcnt(0)) and is 0 again at
// every exit, the statement ends with two wait states before the compiler's code may read its outputs through DPP.
#define IHP_PS_VEC(ZW, WAIT, RM, LM, PRE)                                                                         \
	"v_mov_b32_dpp %[t0], %[XA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t1], %[VA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t2], %[GA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	PRE                                                                                                            \
	WAIT "\n\t"                                                                                                    \
	"v_perm_b32 %[t3], %[TA1], %[TA0], " ZW "\n\t"                                                                 \
	"v_cndmask_b32_e64 %[ZA], %[ZA], %[t3], " RM "\n\t"  /* the refreshed scores (:214-228) */                     \
	"v_pk_add_u16 %[t0], %[t0], %[t1]\n\t"               /* a = x' + v' */                                         \
	"v_pk_add_u16 %[t3], %[YA], %[UA]\n\t"               /* b = y + u */                                           \
	"v_pk_sub_i16 %[t4], %[ZA], %[t0] clamp\n\t"         /* s1: negative <=> a > z */                              \
	"v_pk_max_u16 %[t5], %[ZA], %[t0]\n\t"                                                                         \
	"v_pk_sub_i16 %[t8], %[t5], %[t3] clamp\n\t"         /* s2: negative <=> b > max(z, a) */                      \
	"v_pk_max_u16 %[t5], %[t5], %[t3]\n\t"                                                                         \
	"v_pk_min_u16 %[t5], %[t5], %[Mp]\n\t"               /* z */                                                   \
	"v_pk_sub_i16 %[VA], %[t5], %[UA]\n\t"               /* v = z - u */                                           \
	"v_pk_sub_i16 %[UA], %[t5], %[t1]\n\t"               /* u = z - v' */                                          \
	"v_pk_sub_i16 %[t5], %[t5], %[Qp]\n\t"                                                                         \
	"v_pk_sub_i16 %[t0], %[t0], %[t5]\n\t"                                                                         \
	"v_pk_sub_i16 %[t3], %[t3], %[t5]\n\t"                                                                         \
	"v_pk_max_i16 %[XA], %[t0], 0\n\t"                                                                             \
	"v_pk_max_i16 %[YA], %[t3], 0\n\t"                                                                             \
	"v_pk_lshrrev_b16 %[t1], 15, %[t8] op_sel_hi:[0,1]\n\t"                                                        \
	"v_pk_lshrrev_b16 %[t4], 14, %[t4] op_sel_hi:[0,1]\n\t"                                                        \
	"v_and_or_b32 %[t1], %[t4], %[k22], %[t1]\n\t"                                                                 \
	"v_pk_min_i16 %[t4], %[XA], 4 op_sel_hi:[1,0]\n\t"                                                             \
	"v_pk_min_i16 %[t8], %[YA], 8 op_sel_hi:[1,0]\n\t"                                                             \
	"v_add3_u32 %[t1], %[t4], %[t1], %[t8]\n\t"                                                                    \
	"v_pk_mad_u16 %[acA], %[acA], 16, %[t1] op_sel_hi:[1,0,1]\n\t"  /* the traceback nibble behind the group's earlier ones */ \
	"v_cndmask_b32_e64 %[t2], %[GA], %[t2], %[sp]\n\t"   /* H[en0] = H[en0-1] + u (:318), the others += v */       \
	"v_cndmask_b32_e64 %[t4], %[VA], %[UA], %[sp]\n\t"                                                             \
	"v_pk_lshrrev_b16 %[t4], 8, %[t4] op_sel_hi:[0,1]\n\t"                                                         \
	"v_pk_add_u16 %[GA], %[t4], %[t2]\n\t"                                                                         \
	"s_and_b64 %[inT], " LM ", %[hi]\n\t"
// The same for a diagonal whose en0 is the one of the diagonal before (the second of a steady pair): the cell en0 had an H of its own
// on that diagonal, u and v are exact differences of H, so H[en0-1] + u (:318) and H[en0] + v are one number -- three instructions
// less (the neighbour's H and the two selects).
#define IHP_PS_VEC_SAME_EN(ZW, WAIT, RM, LM, PRE)                                                                         \
	"v_mov_b32_dpp %[t0], %[XA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t1], %[VA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	PRE                                                                                                            \
	WAIT "\n\t"                                                                                                    \
	"v_perm_b32 %[t3], %[TA1], %[TA0], " ZW "\n\t"                                                                 \
	"v_cndmask_b32_e64 %[ZA], %[ZA], %[t3], " RM "\n\t"  /* the refreshed scores (:214-228) */                     \
	"v_pk_add_u16 %[t0], %[t0], %[t1]\n\t"               /* a = x' + v' */                                         \
	"v_pk_add_u16 %[t3], %[YA], %[UA]\n\t"               /* b = y + u */                                           \
	"v_pk_sub_i16 %[t4], %[ZA], %[t0] clamp\n\t"         /* s1: negative <=> a > z */                              \
	"v_pk_max_u16 %[t5], %[ZA], %[t0]\n\t"                                                                         \
	"v_pk_sub_i16 %[t8], %[t5], %[t3] clamp\n\t"         /* s2: negative <=> b > max(z, a) */                      \
	"v_pk_max_u16 %[t5], %[t5], %[t3]\n\t"                                                                         \
	"v_pk_min_u16 %[t5], %[t5], %[Mp]\n\t"               /* z */                                                   \
	"v_pk_sub_i16 %[VA], %[t5], %[UA]\n\t"               /* v = z - u */                                           \
	"v_pk_sub_i16 %[UA], %[t5], %[t1]\n\t"               /* u = z - v' */                                          \
	"v_pk_sub_i16 %[t5], %[t5], %[Qp]\n\t"                                                                         \
	"v_pk_sub_i16 %[t0], %[t0], %[t5]\n\t"                                                                         \
	"v_pk_sub_i16 %[t3], %[t3], %[t5]\n\t"                                                                         \
	"v_pk_max_i16 %[XA], %[t0], 0\n\t"                                                                             \
	"v_pk_max_i16 %[YA], %[t3], 0\n\t"                                                                             \
	"v_pk_lshrrev_b16 %[t1], 15, %[t8] op_sel_hi:[0,1]\n\t"                                                        \
	"v_pk_lshrrev_b16 %[t4], 14, %[t4] op_sel_hi:[0,1]\n\t"                                                        \
	"v_and_or_b32 %[t1], %[t4], %[k22], %[t1]\n\t"                                                                 \
	"v_pk_min_i16 %[t4], %[XA], 4 op_sel_hi:[1,0]\n\t"                                                             \
	"v_pk_min_i16 %[t8], %[YA], 8 op_sel_hi:[1,0]\n\t"                                                             \
	"v_add3_u32 %[t1], %[t4], %[t1], %[t8]\n\t"                                                                    \
	"v_pk_mad_u16 %[acA], %[acA], 16, %[t1] op_sel_hi:[1,0,1]\n\t"  /* the traceback nibble behind the group's earlier ones */ \
	"v_pk_lshrrev_b16 %[t4], 8, %[VA] op_sel_hi:[0,1]\n\t"  /* every H += v, H[en0] too: en0 is the lane it was on the diagonal before, */ \
	"v_pk_add_u16 %[GA], %[t4], %[GA]\n\t"             /* whose H it has, and H[en0-1] + u (:318) is the same number */ \
	"s_and_b64 %[inT], " LM ", %[hi]\n\t"
// one alignment behind the joint test (labels 3..7 are local to the instance: every branch is forward).  ZS2: more terms of
// the vouching limit (the tail's t* + qlen - 1), with the lane of the new maximum in %[i].
#define IHP_PS_PER(M, THR, POS, ZS, INC, WORD, SEXT, BIT1, BIT2, ZS2)                                           \
	"s_and_b64 " M ", " M ", %[inT]\n\t"                                                                           \
	"s_cbranch_scc0 3f\n\t"                                                                                        \
	"s_bcnt1_i32_b64 %[i], " M "\n\t"                                                                              \
	"s_cmp_eq_u32 %[i], 1\n\t"                                                                                     \
	"s_cbranch_scc0 5f\n\t"                                                                                        \
	"s_ff1_i32_b64 %[i], " M "\n\t"                                                                                \
	"v_readlane_b32 %[g], %[GA], %[i]\n\t"                                                                         \
	"s_lshl1_add_u32 %[b], %[i], %[c1]\n\t"                                                                        \
	"s_add_i32 " ZS ", %[r], %[zq]\n\t"                                                                            \
	"s_min_i32 " ZS ", " ZS ", %[b]\n\t"                                                                           \
	ZS2                                                                                                            \
	"s_add_i32 %[i], %[i], %[st]\n\t"                                                                              \
	"s_pack_ll_b32_b16 " POS ", %[i], %[r]\n\t"                                                                    \
	SEXT "\n\t"                                                                                                    \
	"s_and_b64 " M ", " M ", %[sp]\n\t"                                                                            \
	"s_cselect_b32 " ZS ", %[r], " ZS "\n\t"                                                                       \
	"s_add_i32 " THR ", %[g], " INC "\n\t"                                                                         \
	"s_branch 7f\n"                                                                                                \
	"3:\n\t"                                                                                                       \
	"s_cmp_le_i32 %[r], " ZS "\n\t"                                                                                \
	"s_cbranch_scc1 4f\n\t"                                                                                        \
	"s_sub_i32 %[b], " THR ", %[zd]\n\t"                                                                           \
	"s_max_i32 %[b], %[b], 0xffff8000\n\t"                                                                         \
	"v_cmp_ge_i16_sdwa " M ", %[GA], %[b] src0_sel:" WORD " src1_sel:WORD_0\n\t"                                   \
	"s_and_b64 " M ", " M ", %[inT]\n\t"                                                                           \
	"s_cbranch_scc0 6f\n"                                                                                          \
	"4:\n\t"                                                                                                       \
	"s_add_i32 " THR ", " THR ", " INC "\n\t"                                                                      \
	"s_branch 7f\n"                                                                                                \
	"5:\n\t"                                                                                                       \
	"s_or_b32 %[stat], %[stat], " BIT1 "\n\t"                                                                      \
	"s_branch 7f\n"                                                                                                \
	"6:\n\t"                                                                                                       \
	"s_or_b32 %[stat], %[stat], " BIT2 "\n"                                                                        \
	"7:\n\t"
#define IHP_PS_EZ(EXIT, ZS2)                                                                                    \
	"v_cmp_gt_i16_sdwa %[m0], %[GA], %[thr0] src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                  \
	"v_cmp_gt_i16_sdwa %[m1], %[GA], %[thr1] src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                  \
	"s_or_b64 vcc, %[m0], %[m1]\n\t"                                                                               \
	"s_and_b64 vcc, vcc, %[inT]\n\t"                                                                               \
	"s_cbranch_scc1 1f\n\t"                                                                                        \
	"s_cmp_le_i32 %[r], %[zsm]\n\t"                                                                                \
	"s_cbranch_scc0 1f\n\t"                                                                                        \
	"s_add_i32 %[thr0], %[thr0], %[inc0]\n\t"                                                                      \
	"s_add_i32 %[thr1], %[thr1], %[inc1]\n\t"                                                                      \
	"s_branch 9f\n"                                                                                                \
	"1:\n\t"                                                                                                       \
	IHP_PS_PER("%[m0]", "%[thr0]", "%[pos0]", "%[zs0]", "%[inc0]", "WORD_0", "s_sext_i32_i16 %[g], %[g]", "1", "2", ZS2("%[zs0]"))  \
	IHP_PS_PER("%[m1]", "%[thr1]", "%[pos1]", "%[zs1]", "%[inc1]", "WORD_1", "s_ashr_i32 %[g], %[g], 16", "4", "8", ZS2("%[zs1]"))  \
	"s_min_i32 %[zsm], %[zs0], %[zs1]\n\t"                                                                         \
	"s_cmp_lg_u32 %[stat], 0\n\t"                                                                                  \
	"s_cbranch_scc1 " EXIT "\n"                                                                                    \
	"9:\n\t"
#define IHP_PS_ZS2_NONE(ZS) ""
#define IHP_PS_ZS2_TAIL(ZS) "s_add_i32 %[b], %[i], %[c2]\n\t" "s_min_i32 " ZS ", " ZS ", %[b]\n\t"

// Runs pairs of steady diagonals from C.r ((C.r + w) even, w even, no block 4, C.geLoM / spM / hiM current) while
// C.r + 1 < C.lim.  Returns 0 when that bound is reached; otherwise diagonal C.r is done except for pair_ez_lean of the
// alignments named in bits 3..0 and the step to C.r + 1, and bit 4 says that it was the second diagonal of its pair.
__device__ __forceinline__ int pair_steady_pairs_asm(PairState &S, const PairEnv &E, PairCtl &C, const int zq_)
{
	typedef const __attribute__((address_space(3))) unsigned *lds_cu32;
	unsigned qp = (unsigned)(unsigned long long)(lds_cu32)S.qptr;
	const unsigned qp0 = qp;
	unsigned vof = (unsigned)lane_id() * 4u + (unsigned)((C.r >> 2) + (S.st >> 4)) * 320u;   // the traceback slot the next store goes to
	unsigned long long ge = (unsigned long long)uni((long long)C.geLoM), sp = (unsigned long long)uni((long long)C.spM), hi = (unsigned long long)uni((long long)C.hiM);
	int r = uni(C.r), stat;
	const int lim = uni(C.lim), st = uni(S.st), zq = uni(zq_), c1 = uni(2 * S.st + E.w), zd = uni(E.zd), inc0 = uni(S.inc0), inc1 = uni(S.inc1);
	const unsigned Mp = (unsigned)uni((int)E.Mp), Qp = (unsigned)uni((int)E.Qp), k22 = 0x00020002u;
	const unsigned long long pbase = (unsigned long long)uni((long long)(unsigned long long)E.p);
	unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8;
	unsigned long long m0, m1, inT;
	int i, g, b, zsm;
	asm volatile(
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_min_i32 %[zsm], %[zs0], %[zs1]\n\t"
		"s_mov_b32 %[stat], 0\n"
		"10:\n\t"
		"s_add_i32 %[b], %[r], 1\n\t"
		"s_cmp_lt_i32 %[b], %[lim]\n\t"
		"s_cbranch_scc0 90f\n\t"
		"v_add_u32_e32 %[qp], -8, %[qp]\n\t"
		"ds_read_b32 %[t6], %[qp] offset:8\n\t"
		"ds_read_b32 %[t7], %[qp] offset:4\n\t"
		// ---- the first diagonal of the pair: st0 grows on the step behind it
		IHP_PS_VEC("%[t6]", "s_waitcnt lgkmcnt(1)", "%[ge]", "%[ge]", "")
		IHP_PS_EZ("80f", IHP_PS_ZS2_NONE)
		"s_lshl_b64 %[ge], %[ge], 1\n\t"
		"s_add_i32 %[r], %[r], 1\n\t"
		// ---- the second: it closes a traceback group when r & 2; en0 grows behind it
		IHP_PS_VEC_SAME_EN("%[t7]", "s_waitcnt lgkmcnt(0)", "%[ge]", "%[ge]", "")
		"s_bitcmp1_b32 %[r], 1\n\t"
		"s_cbranch_scc0 2f\n\t"
		"global_store_dword %[vof], %[acA], %[pb]\n\t"
		"global_store_dword %[vof], %[acB], %[pb] offset:256\n\t"
		"v_add_u32_e32 %[vof], 0x140, %[vof]\n"
		"2:\n\t"
		IHP_PS_EZ("81f", IHP_PS_ZS2_NONE)
		"s_lshl_b64 %[sp], %[sp], 1\n\t"
		"s_lshl_b64 %[hi], %[hi], 1\n\t"
		"s_or_b64 %[hi], %[hi], 1\n\t"
		"s_add_i32 %[r], %[r], 1\n\t"
		"s_branch 10b\n"
		"80:\n\t"                                            // left on the first diagonal: its partner's score word is not used
		"s_waitcnt lgkmcnt(0)\n\t"
		"v_add_u32_e32 %[qp], 4, %[qp]\n\t"
		"s_branch 90f\n"
		"81:\n\t"
		"s_or_b32 %[stat], %[stat], 16\n"
		"90:\n\t"
		"s_nop 1"
		: [XA] "+v"(S.XA), [VA] "+v"(S.VA), [UA] "+v"(S.UA), [YA] "+v"(S.YA), [ZA] "+v"(S.ZA), [GA] "+v"(S.GA), [acA] "+v"(S.accA),
		  [qp] "+v"(qp), [vof] "+v"(vof), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
		  [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8),
		  [ge] "+s"(ge), [sp] "+s"(sp), [hi] "+s"(hi), [thr0] "+s"(S.thr0), [thr1] "+s"(S.thr1), [pos0] "+s"(S.pos0), [pos1] "+s"(S.pos1),
		  [zs0] "+s"(C.zsafe0), [zs1] "+s"(C.zsafe1), [r] "+s"(r), [stat] "=&s"(stat), [m0] "=&s"(m0), [m1] "=&s"(m1), [inT] "=&s"(inT),
		  [i] "=&s"(i), [g] "=&s"(g), [b] "=&s"(b), [zsm] "=&s"(zsm)
		: [acB] "v"(S.accB), [TA0] "v"(S.TA0), [TA1] "v"(S.TA1), [Mp] "s"(Mp), [Qp] "s"(Qp), [k22] "s"(k22), [inc0] "s"(inc0), [inc1] "s"(inc1),
		  [lim] "s"(lim), [st] "s"(st), [zq] "s"(zq), [c1] "s"(c1), [zd] "s"(zd), [pb] "s"(pbase)
		: "vcc", "scc", "memory");
	S.qptr += (int)(qp - qp0) / 4;
	C.geLoM = ge; C.spM = sp; C.hiM = hi; C.r = r;
	C.st0 = (r - E.w + 1) >> 1; C.en0 = (r + E.w) >> 1;
	return stat;
}
// The same for a run of TAIL diagonals (pair_tail_qrun: the band cut by the end of the query; no move, no change of the
// computed blocks or of the refreshed score groups inside the run).  What differs from the steady run: st0 grows on EVERY
// diagonal (lo: the lanes >= st0 - st; rf: the refreshed lanes st0 - st .. sc, both shift by one per diagonal), en0 behind an
// odd one; H[st0] is the end-of-query score of every diagonal (:353-354), kept here as mq = mqe + r (q+e) beside the
// thresholds; the vouching limit has its third term back (t* + qlen - 1); a diagonal loads the NEXT one's score word, so a
// run may start and end on either parity (the runs of the tail are 5-15 diagonals long).  All 64 lanes are computed and
// committed: the lanes above the last computed block hold zeros when they enter the band (nothing has touched them since
// the sweep began or since a move cleared block 4), and the caller puts the zeros back.
// Both alignments must be live.  Returns 0 at C.lim; otherwise diagonal C.r lacks pair_ez_lean of the alignments in bits 3..0
// and the step to C.r + 1 (mq is then already that of C.r + 1).
__device__ __forceinline__ int pair_tail_run_asm(PairState &S, const PairEnv &E, PairCtl &C, const int zq_, unsigned long long lo_, unsigned long long rf_,
                                                 unsigned long long sp_, unsigned long long hi_, int la_)
{
	typedef const __attribute__((address_space(3))) unsigned *lds_cu32;
	const unsigned qp0 = (unsigned)(unsigned long long)(lds_cu32)S.qptr;
	unsigned qp = qp0 - 4u;                                  // the next diagonal's score word
	unsigned vof = (unsigned)lane_id() * 4u + (unsigned)((C.r >> 2) + (S.st >> 4)) * 320u;
	unsigned long long lo = (unsigned long long)uni((long long)lo_), rf = (unsigned long long)uni((long long)rf_), sp = (unsigned long long)uni((long long)sp_), hi = (unsigned long long)uni((long long)hi_);
	int r = uni(C.r), la = uni(la_), stat;
	const int r_in = r;
	const int lim = uni(C.lim), st = uni(S.st), zq = uni(zq_), c1 = uni(2 * S.st + E.w), c2 = uni(S.st + E.qlm1), zd = uni(E.zd), inc0 = uni(S.inc0), inc1 = uni(S.inc1);
	const unsigned Mp = (unsigned)uni((int)E.Mp), Qp = (unsigned)uni((int)E.Qp), k22 = 0x00020002u;
	const unsigned long long pbase = (unsigned long long)uni((long long)(unsigned long long)E.p);
	int mq0 = uni(S.mqe0 + r * E.qe), mq1 = uni(S.mqe1 + r * E.qe);
	const int mt0_in = pair_cold_get(S, PC_MQE_T0), mt1_in = pair_cold_get(S, PC_MQE_T1);
	int mt0 = mt0_in, mt1 = mt1_in;
	unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8;
	unsigned long long m0, m1, inT;
	int i, g, b, zsm;
#define IHP_PT_MQE                                                                                              \
	"v_readlane_b32 %[g], %[GA], %[la]\n\t"              /* H[st0] + r (q+e) of both */                            \
	"s_add_i32 %[b], %[la], %[st]\n\t"                                                                             \
	"s_sext_i32_i16 %[i], %[g]\n\t"                                                                                \
	"s_cmp_gt_i32 %[i], %[mq0]\n\t"                                                                                \
	"s_cselect_b32 %[mq0], %[i], %[mq0]\n\t"                                                                       \
	"s_cselect_b32 %[mt0], %[b], %[mt0]\n\t"                                                                       \
	"s_ashr_i32 %[i], %[g], 16\n\t"                                                                                \
	"s_cmp_gt_i32 %[i], %[mq1]\n\t"                                                                                \
	"s_cselect_b32 %[mq1], %[i], %[mq1]\n\t"                                                                       \
	"s_cselect_b32 %[mt1], %[b], %[mt1]\n\t"                                                                       \
	"s_add_i32 %[mq0], %[mq0], %[inc0]\n\t"                                                                        \
	"s_add_i32 %[mq1], %[mq1], %[inc1]\n\t"
#define IHP_PT_NEXT(EXIT)                                                                                       \
	"s_lshl_b64 %[lo], %[lo], 1\n\t"                                                                               \
	"s_lshl_b64 %[rf], %[rf], 1\n\t"                                                                               \
	"s_add_i32 %[la], %[la], 1\n\t"                                                                                \
	"v_add_u32_e32 %[qp], -4, %[qp]\n\t"                                                                           \
	"s_add_i32 %[r], %[r], 1\n\t"                                                                                  \
	"s_cmp_lt_i32 %[r], %[lim]\n\t"                                                                                \
	"s_cbranch_scc0 " EXIT "\n\t"
	asm volatile(
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_min_i32 %[zsm], %[zs0], %[zs1]\n\t"
		"s_mov_b32 %[stat], 0\n\t"
		"s_bitcmp1_b32 %[r], 0\n\t"
		"s_cbranch_scc1 11f\n\t"
		"ds_read_b32 %[t6], %[qp] offset:4\n"
		"10:\n\t"
		// ---- an even diagonal: en0 stays
		"ds_read_b32 %[t7], %[qp]\n\t"
		IHP_PS_VEC("%[t6]", "s_waitcnt lgkmcnt(1)", "%[rf]", "%[lo]", "")
		IHP_PT_MQE
		IHP_PS_EZ("90f", IHP_PS_ZS2_TAIL)
		IHP_PT_NEXT("90f")
		"s_branch 12f\n"
		"11:\n\t"
		"ds_read_b32 %[t7], %[qp] offset:4\n"
		"12:\n\t"
		// ---- an odd one: it closes a traceback group when r & 2; en0 grows behind it
		"ds_read_b32 %[t6], %[qp]\n\t"
		IHP_PS_VEC_SAME_EN("%[t7]", "s_waitcnt lgkmcnt(1)", "%[rf]", "%[lo]", "")
		"s_bitcmp1_b32 %[r], 1\n\t"
		"s_cbranch_scc0 2f\n\t"
		"global_store_dword %[vof], %[acA], %[pb]\n\t"
		"global_store_dword %[vof], %[acB], %[pb] offset:256\n\t"
		"v_add_u32_e32 %[vof], 0x140, %[vof]\n"
		"2:\n\t"
		IHP_PT_MQE
		IHP_PS_EZ("90f", IHP_PS_ZS2_TAIL)
		"s_lshl_b64 %[sp], %[sp], 1\n\t"
		"s_lshl_b64 %[hi], %[hi], 1\n\t"
		"s_or_b64 %[hi], %[hi], 1\n\t"
		IHP_PT_NEXT("90f")
		"s_branch 10b\n"
		"90:\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_nop 1"
		: [XA] "+v"(S.XA), [VA] "+v"(S.VA), [UA] "+v"(S.UA), [YA] "+v"(S.YA), [ZA] "+v"(S.ZA), [GA] "+v"(S.GA), [acA] "+v"(S.accA),
		  [qp] "+v"(qp), [vof] "+v"(vof), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
		  [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8),
		  [lo] "+s"(lo), [rf] "+s"(rf), [sp] "+s"(sp), [hi] "+s"(hi), [la] "+s"(la), [thr0] "+s"(S.thr0), [thr1] "+s"(S.thr1), [pos0] "+s"(S.pos0), [pos1] "+s"(S.pos1),
		  [zs0] "+s"(C.zsafe0), [zs1] "+s"(C.zsafe1), [mq0] "+s"(mq0), [mq1] "+s"(mq1), [mt0] "+s"(mt0), [mt1] "+s"(mt1),
		  [r] "+s"(r), [stat] "=&s"(stat), [m0] "=&s"(m0), [m1] "=&s"(m1), [inT] "=&s"(inT),
		  [i] "=&s"(i), [g] "=&s"(g), [b] "=&s"(b), [zsm] "=&s"(zsm)
		: [acB] "v"(S.accB), [TA0] "v"(S.TA0), [TA1] "v"(S.TA1), [Mp] "s"(Mp), [Qp] "s"(Qp), [k22] "s"(k22), [inc0] "s"(inc0), [inc1] "s"(inc1),
		  [lim] "s"(lim), [st] "s"(st), [zq] "s"(zq), [c1] "s"(c1), [c2] "s"(c2), [zd] "s"(zd), [pb] "s"(pbase)
		: "vcc", "scc", "memory");
#undef IHP_PT_MQE
#undef IHP_PT_NEXT
	// qp points at the score word of the diagonal after the last one computed: r when the run ended, r + 1 behind a pending one
	const int done = stat ? r + 1 - r_in : r - r_in;
	(void)qp;
	S.qptr -= done;
	const int r_mq = stat ? r + 1 : r;
	S.mqe0 = mq0 - r_mq * E.qe; S.mqe1 = mq1 - r_mq * E.qe;
	if (mt0 != mt0_in) pair_cold_set(S, PC_MQE_T0, mt0);
	if (mt1 != mt1_in) pair_cold_set(S, PC_MQE_T1, mt1);
	C.r = r;
	return stat;
}
// The same for the EARLY diagonals 1 .. w + 30 while block 4 is not computed (st = 0; pair_steady_step<false, 2, ...>): lane 0
// takes v1 = q (:211), the cell t = r is the boundary y = 0, u = q while the computed blocks reach it (:212), the band grows
// from one cell.  GROW = 1: r <= w -- st0 = 0, en0 = r: the top cell IS the boundary cell and moves up every diagonal, the
// scores of all computed blocks are refreshed (rf = the computed lanes, constant: the caller cuts the runs where a block
// enters).  GROW = 0: w < r -- the steady pattern (st0 behind an even r, en0 behind an odd one; rf = the lanes >= st0).
// All 64 lanes are computed and committed; the caller puts the zeros back above the computed blocks.  Runs start and end on
// either parity.  Returns as pair_tail_run_asm (without the end-of-query part: the query's end is far away).
template <int GROW>
__device__ __forceinline__ int pair_early_run_asm(PairState &S, const PairEnv &E, PairCtl &C, const int zq_, unsigned long long rf_, unsigned long long sp_,
                                                  unsigned long long hi_, unsigned long long tr_)
{
	typedef const __attribute__((address_space(3))) unsigned *lds_cu32;
	const unsigned qp0 = (unsigned)(unsigned long long)(lds_cu32)S.qptr;
	unsigned qp = qp0 - 4u;                                  // the next diagonal's score word
	unsigned vof = (unsigned)lane_id() * 4u + (unsigned)(C.r >> 2) * 320u;
	unsigned long long rf = (unsigned long long)uni((long long)rf_), sp = (unsigned long long)uni((long long)sp_), hi = (unsigned long long)uni((long long)hi_), tr = (unsigned long long)uni((long long)tr_);
	int r = uni(C.r), stat;
	const int r_in = r;
	const int lim = uni(C.lim), st = 0, zq = uni(zq_), c1 = uni(E.w), zd = uni(E.zd), inc0 = uni(S.inc0), inc1 = uni(S.inc1);
	const unsigned Mp = (unsigned)uni((int)E.Mp), Qp = (unsigned)uni((int)E.Qp), k22 = 0x00020002u;
	const unsigned Qv = Qp;                                  // (a vector copy: a select cannot take its mask and a value from the scalar file)
	// (the early diagonals sit in front of a loop whose control the compiler takes for divergent: say again what is uniform)
	int thr0 = uni(S.thr0), thr1 = uni(S.thr1), pos0 = uni(S.pos0), pos1 = uni(S.pos1), zs0 = uni(C.zsafe0), zs1 = uni(C.zsafe1);
	const unsigned long long pbase = (unsigned long long)uni((long long)(unsigned long long)E.p);
	unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8;
	unsigned long long m0, m1, inT;
	int i, g, b, zsm;
#define IHP_PE_PRE                                                                                              \
	"v_writelane_b32 %[t1], %[Qp], 0\n\t"                /* v1 = q (:211) */                                       \
	"v_cndmask_b32_e64 %[UA], %[UA], %[Qv], %[tr]\n\t"   /* u[r] = q, y[r] = 0 (:212) */                           \
	"v_cndmask_b32_e64 %[YA], %[YA], 0, %[tr]\n\t"
#define IHP_PE_NEXT(EXIT)                                                                                       \
	"v_add_u32_e32 %[qp], -4, %[qp]\n\t"                                                                           \
	"s_add_i32 %[r], %[r], 1\n\t"                                                                                  \
	"s_cmp_lt_i32 %[r], %[lim]\n\t"                                                                                \
	"s_cbranch_scc0 " EXIT "\n\t"
#define IHP_PE_UP                                                                                               \
	"s_lshl_b64 %[sp], %[sp], 1\n\t"                                                                               \
	"s_lshl_b64 %[hi], %[hi], 1\n\t"                                                                               \
	"s_or_b64 %[hi], %[hi], 1\n\t"
#define IHP_PE_ASM(EVEN_EXTRA, VEC_ODD) \
	asm volatile( \
		"s_waitcnt lgkmcnt(0)\n\t" \
		"s_min_i32 %[zsm], %[zs0], %[zs1]\n\t" \
		"s_mov_b32 %[stat], 0\n\t" \
		"s_bitcmp1_b32 %[r], 0\n\t" \
		"s_cbranch_scc1 11f\n\t" \
		"ds_read_b32 %[t6], %[qp] offset:4\n" \
		"10:\n\t" \
		"ds_read_b32 %[t7], %[qp]\n\t" \
		IHP_PS_VEC("%[t6]", "s_waitcnt lgkmcnt(1)", "%[rf]", "%[rf]", IHP_PE_PRE) \
		IHP_PS_EZ("90f", IHP_PS_ZS2_NONE) \
		"s_lshl_b64 %[tr], %[tr], 1\n\t" \
		EVEN_EXTRA \
		IHP_PE_NEXT("90f") \
		"s_branch 12f\n" \
		"11:\n\t" \
		"ds_read_b32 %[t7], %[qp] offset:4\n" \
		"12:\n\t" \
		"ds_read_b32 %[t6], %[qp]\n\t" \
		VEC_ODD("%[t7]", "s_waitcnt lgkmcnt(1)", "%[rf]", "%[rf]", IHP_PE_PRE) \
		"s_bitcmp1_b32 %[r], 1\n\t" \
		"s_cbranch_scc0 2f\n\t" \
		"global_store_dword %[vof], %[acA], %[pb]\n\t" \
		"global_store_dword %[vof], %[acB], %[pb] offset:256\n\t" \
		"v_add_u32_e32 %[vof], 0x140, %[vof]\n" \
		"2:\n\t" \
		IHP_PS_EZ("90f", IHP_PS_ZS2_NONE) \
		"s_lshl_b64 %[tr], %[tr], 1\n\t" \
		IHP_PE_UP \
		IHP_PE_NEXT("90f") \
		"s_branch 10b\n" \
		"90:\n\t" \
		"s_waitcnt lgkmcnt(0)\n\t" \
		"s_nop 1" \
		: [XA] "+v"(S.XA), [VA] "+v"(S.VA), [UA] "+v"(S.UA), [YA] "+v"(S.YA), [ZA] "+v"(S.ZA), [GA] "+v"(S.GA), [acA] "+v"(S.accA), \
		  [qp] "+v"(qp), [vof] "+v"(vof), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), \
		  [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), \
		  [rf] "+s"(rf), [sp] "+s"(sp), [hi] "+s"(hi), [tr] "+s"(tr), [thr0] "+s"(thr0), [thr1] "+s"(thr1), [pos0] "+s"(pos0), [pos1] "+s"(pos1), \
		  [zs0] "+s"(zs0), [zs1] "+s"(zs1), \
		  [r] "+s"(r), [stat] "=&s"(stat), [m0] "=&s"(m0), [m1] "=&s"(m1), [inT] "=&s"(inT), \
		  [i] "=&s"(i), [g] "=&s"(g), [b] "=&s"(b), [zsm] "=&s"(zsm) \
		: [acB] "v"(S.accB), [TA0] "v"(S.TA0), [TA1] "v"(S.TA1), [Qv] "v"(Qv), [Mp] "s"(Mp), [Qp] "s"(Qp), [k22] "s"(k22), [inc0] "s"(inc0), [inc1] "s"(inc1), \
		  [lim] "s"(lim), [st] "s"(st), [zq] "s"(zq), [c1] "s"(c1), [zd] "s"(zd), [pb] "s"(pbase) \
		: "vcc", "scc", "memory");
	// (GROW: en0 = r is a new cell on every diagonal; otherwise the odd diagonals keep the en0 of the even one before them)
	if (GROW) { IHP_PE_ASM(IHP_PE_UP, IHP_PS_VEC) } else { IHP_PE_ASM("s_lshl_b64 %[rf], %[rf], 1\n\t", IHP_PS_VEC_SAME_EN) }
#undef IHP_PE_ASM
	S.qptr -= stat ? r + 1 - r_in : r - r_in;
	(void)qp;
	S.thr0 = thr0; S.thr1 = thr1; S.pos0 = pos0; S.pos1 = pos1; C.zsafe0 = zs0; C.zsafe1 = zs1;
	C.r = r;
	return stat;
}
#undef IHP_PE_PRE
#undef IHP_PE_NEXT
#undef IHP_PE_UP

// The diagonals C.r .. bound-1, all with or all without block 4: single steps until r + w is even, then pairs.  With an even
// w the first diagonal of a pair is an even one: only the second can close a group of four (FLUSH).
// The diagonals that compute block 4 (en0 >= st + 64: the last 2 w - 97 diagonals in front of a move) by hand, any parity,
// even w.  Slot B's sixteen cells are a second pass of the cell over its own registers (lane 0 takes the block edge from lane
// 63 of slot A as it stood on the diagonal before, the special cell en0 is lane hiB of slot B, the scores of its lanes below
// st0 - st are refreshed, the others are the never-refreshed z); slot A has no special cell.  The scalar side is the steady
// loop's with one more test in front: a lane of slot B above a threshold sends both alignments to pair_ez_lean<K, true>
// (stat = 5) -- the running maximum sits near the main diagonal, in slot A.  (The compiler's version runs pair_ez_lean for both
// alignments on every such diagonal.)  Measured on 50 000 pairs of 247 x 327: -1.9 % VALU, -4.0 % SALU, -3 % time for the
// kernel.  The diagonal of a move written the same way -- one statement for one diagonal -- executed MORE instructions than
// the general step (the statement's set-up and the fences around it) and is not kept.
// Returns as pair_tail_run_asm: 0 at C.lim; otherwise diagonal C.r lacks pair_ez_lean of the alignments in bits 3..0 and the
// step to C.r + 1.
#define IHP_PH_CELL(X, V, U, Y, Z, AC)                                                                            \
	"v_pk_add_u16 %[t0], %[t0], %[t1]\n\t"               /* a = x' + v' */                                         \
	"v_pk_add_u16 %[t3], " Y ", " U "\n\t"               /* b = y + u */                                           \
	"v_pk_sub_i16 %[t4], " Z ", %[t0] clamp\n\t"                                                                   \
	"v_pk_max_u16 %[t5], " Z ", %[t0]\n\t"                                                                         \
	"v_pk_sub_i16 %[t8], %[t5], %[t3] clamp\n\t"                                                                   \
	"v_pk_max_u16 %[t5], %[t5], %[t3]\n\t"                                                                         \
	"v_pk_min_u16 %[t5], %[t5], %[Mp]\n\t"               /* z */                                                   \
	"v_pk_sub_i16 " V ", %[t5], " U "\n\t"               /* v = z - u */                                           \
	"v_pk_sub_i16 " U ", %[t5], %[t1]\n\t"               /* u = z - v' */                                          \
	"v_pk_sub_i16 %[t5], %[t5], %[Qp]\n\t"                                                                         \
	"v_pk_sub_i16 %[t0], %[t0], %[t5]\n\t"                                                                         \
	"v_pk_sub_i16 %[t3], %[t3], %[t5]\n\t"                                                                         \
	"v_pk_max_i16 " X ", %[t0], 0\n\t"                                                                             \
	"v_pk_max_i16 " Y ", %[t3], 0\n\t"                                                                             \
	"v_pk_lshrrev_b16 %[t1], 15, %[t8] op_sel_hi:[0,1]\n\t"                                                        \
	"v_pk_lshrrev_b16 %[t4], 14, %[t4] op_sel_hi:[0,1]\n\t"                                                        \
	"v_and_or_b32 %[t1], %[t4], %[k22], %[t1]\n\t"                                                                 \
	"v_pk_min_i16 %[t4], " X ", 4 op_sel_hi:[1,0]\n\t"                                                             \
	"v_pk_min_i16 %[t8], " Y ", 8 op_sel_hi:[1,0]\n\t"                                                             \
	"v_add3_u32 %[t1], %[t4], %[t1], %[t8]\n\t"                                                                    \
	"v_pk_mad_u16 " AC ", " AC ", 16, %[t1] op_sel_hi:[1,0,1]\n\t"
#define IHP_PH_DIAG(ZWA, ZWB)                                                                                     \
	"v_readlane_b32 %[ex], %[XA], 63\n\t"                /* the block edge of slot B: slot A's lane 63 of diagonal r - 1 */ \
	"v_readlane_b32 %[ev], %[VA], 63\n\t"                                                                          \
	"v_readlane_b32 %[eg], %[GA], 63\n\t"                                                                          \
	"v_mov_b32_dpp %[t0], %[XB] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t1], %[VB] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t2], %[GB] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_writelane_b32 %[t0], %[ex], 0\n\t"                                                                          \
	"v_writelane_b32 %[t1], %[ev], 0\n\t"                                                                          \
	"v_writelane_b32 %[t2], %[eg], 0\n\t"                                                                          \
	"s_waitcnt lgkmcnt(2)\n\t"                                                                                     \
	"v_perm_b32 %[t3], %[TB1], %[TB0], " ZWB "\n\t"                                                                \
	"v_cndmask_b32_e64 %[t9], %[ZWv], %[t3], %[nge]\n\t" /* refreshed below st0 - st, never refreshed above */     \
	IHP_PH_CELL("%[XB]", "%[VB]", "%[UB]", "%[YB]", "%[t9]", "%[acB]")                                             \
	"v_cndmask_b32_e64 %[t2], %[GB], %[t2], %[spB]\n\t"  /* H[en0] = H[en0-1] + u (:318), the others += v */       \
	"v_cndmask_b32_e64 %[t4], %[VB], %[UB], %[spB]\n\t"                                                            \
	"v_pk_lshrrev_b16 %[t4], 8, %[t4] op_sel_hi:[0,1]\n\t"                                                         \
	"v_pk_add_u16 %[GB], %[t4], %[t2]\n\t"                                                                         \
	"v_mov_b32_dpp %[t0], %[XA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_mov_b32_dpp %[t1], %[VA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                            \
	"v_perm_b32 %[t3], %[TA1], %[TA0], " ZWA "\n\t"                                                                \
	"v_cndmask_b32_e64 %[ZA], %[ZA], %[t3], %[ge]\n\t"   /* the refreshed scores (:214-228) */                     \
	IHP_PH_CELL("%[XA]", "%[VA]", "%[UA]", "%[YA]", "%[ZA]", "%[acA]")                                             \
	"v_pk_lshrrev_b16 %[t4], 8, %[VA] op_sel_hi:[0,1]\n\t"                                                         \
	"v_pk_add_u16 %[GA], %[t4], %[GA]\n\t"               /* en0 is in block 4: every H of slot A += v */           \
	"s_mov_b64 %[inT], %[ge]\n\t"
#define IHP_PH_EZ(EXITB, EXIT)                                                                                    \
	"v_cmp_gt_i16_sdwa %[m0], %[GB], %[thr0] src0_sel:WORD_0 src1_sel:WORD_0\n\t"                                  \
	"v_cmp_gt_i16_sdwa %[m1], %[GB], %[thr1] src0_sel:WORD_1 src1_sel:WORD_0\n\t"                                  \
	"s_or_b64 vcc, %[m0], %[m1]\n\t"                                                                               \
	"s_and_b64 vcc, vcc, %[inB]\n\t"                                                                               \
	"s_cbranch_scc1 " EXITB "\n\t"                                                                                 \
	IHP_PS_EZ(EXIT, IHP_PS_ZS2_NONE)
#define IHP_PH_NEXT(EXIT)                                                                                         \
	"v_add_u32_e32 %[qp], -4, %[qp]\n\t"                                                                           \
	"v_add_u32_e32 %[qpb], -4, %[qpb]\n\t"                                                                         \
	"s_add_i32 %[r], %[r], 1\n\t"                                                                                  \
	"s_cmp_lt_i32 %[r], %[lim]\n\t"                                                                                \
	"s_cbranch_scc0 " EXIT "\n\t"
__device__ __forceinline__ int pair_hasb_run_asm(PairState &S, const PairEnv &E, PairCtl &C, const int zq_)
{
	typedef const __attribute__((address_space(3))) unsigned *lds_cu32;
	const unsigned qp0 = (unsigned)(unsigned long long)(lds_cu32)S.qptr;
	unsigned qp = qp0 - 4u, qpb = qp0 - 4u + 4u * (unsigned)S.qoffB;   // the next diagonal's score words
	unsigned vof = (unsigned)lane_id() * 4u + (unsigned)((C.r >> 2) + (S.st >> 4)) * 320u;
	int r = uni(C.r), stat;
	const int r_in = r;
	const int hiB = uni(C.en0 - S.st - 64);
	unsigned long long ge = (unsigned long long)uni((long long)C.geLoM), nge = ~ge;
	unsigned long long spB = 1ull << hiB, inB = (2ull << hiB) - 1ull, sp0 = 0ull;
	const int lim = uni(C.lim), st = uni(S.st), zq = uni(zq_), c1 = uni(2 * S.st + E.w), zd = uni(E.zd), inc0 = uni(S.inc0), inc1 = uni(S.inc1);
	const unsigned Mp = (unsigned)uni((int)E.Mp), Qp = (unsigned)uni((int)E.Qp), k22 = 0x00020002u;
	const unsigned ZWv = E.ZWp;
	const unsigned long long pbase = (unsigned long long)uni((long long)(unsigned long long)E.p);
	unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11;
	unsigned long long m0, m1, inT;
	int i, g, b, zsm, ex, ev, eg;
	asm volatile(
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_min_i32 %[zsm], %[zs0], %[zs1]\n\t"
		"s_mov_b32 %[stat], 0\n\t"
		"s_bitcmp1_b32 %[r], 0\n\t"
		"s_cbranch_scc1 11f\n\t"
		"ds_read_b32 %[t6], %[qp] offset:4\n\t"
		"ds_read_b32 %[t10], %[qpb] offset:4\n"
		"10:\n\t"
		// ---- an even diagonal: st0 grows behind it
		"ds_read_b32 %[t7], %[qp]\n\t"
		"ds_read_b32 %[t11], %[qpb]\n\t"
		IHP_PH_DIAG("%[t6]", "%[t10]")
		IHP_PH_EZ("91f", "90f")
		"s_lshl_b64 %[ge], %[ge], 1\n\t"
		"s_not_b64 %[nge], %[ge]\n\t"
		IHP_PH_NEXT("90f")
		"s_branch 12f\n"
		"11:\n\t"
		"ds_read_b32 %[t7], %[qp] offset:4\n\t"
		"ds_read_b32 %[t11], %[qpb] offset:4\n"
		"12:\n\t"
		// ---- an odd one: it closes a traceback group when r & 2; en0 grows behind it
		"ds_read_b32 %[t6], %[qp]\n\t"
		"ds_read_b32 %[t10], %[qpb]\n\t"
		IHP_PH_DIAG("%[t7]", "%[t11]")
		"s_bitcmp1_b32 %[r], 1\n\t"
		"s_cbranch_scc0 2f\n\t"
		"global_store_dword %[vof], %[acA], %[pb]\n\t"
		"global_store_dword %[vof], %[acB], %[pb] offset:256\n\t"
		"v_add_u32_e32 %[vof], 0x140, %[vof]\n"
		"2:\n\t"
		IHP_PH_EZ("91f", "90f")
		"s_lshl_b64 %[spB], %[spB], 1\n\t"
		"s_lshl_b64 %[inB], %[inB], 1\n\t"
		"s_or_b64 %[inB], %[inB], 1\n\t"
		IHP_PH_NEXT("90f")
		"s_branch 10b\n"
		"91:\n\t"
		"s_mov_b32 %[stat], 5\n"
		"90:\n\t"
		"s_waitcnt lgkmcnt(0)\n\t"
		"s_nop 1"
		: [XA] "+v"(S.XA), [VA] "+v"(S.VA), [UA] "+v"(S.UA), [YA] "+v"(S.YA), [ZA] "+v"(S.ZA), [GA] "+v"(S.GA), [acA] "+v"(S.accA),
		  [XB] "+v"(S.XB), [VB] "+v"(S.VB), [UB] "+v"(S.UB), [YB] "+v"(S.YB), [GB] "+v"(S.GB), [acB] "+v"(S.accB),
		  [qp] "+v"(qp), [qpb] "+v"(qpb), [vof] "+v"(vof), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5),
		  [t6] "=&v"(t6), [t7] "=&v"(t7), [t8] "=&v"(t8), [t9] "=&v"(t9), [t10] "=&v"(t10), [t11] "=&v"(t11),
		  [ge] "+s"(ge), [nge] "+s"(nge), [spB] "+s"(spB), [inB] "+s"(inB), [thr0] "+s"(S.thr0), [thr1] "+s"(S.thr1), [pos0] "+s"(S.pos0), [pos1] "+s"(S.pos1),
		  [zs0] "+s"(C.zsafe0), [zs1] "+s"(C.zsafe1), [r] "+s"(r), [stat] "=&s"(stat), [m0] "=&s"(m0), [m1] "=&s"(m1), [inT] "=&s"(inT),
		  [i] "=&s"(i), [g] "=&s"(g), [b] "=&s"(b), [zsm] "=&s"(zsm), [ex] "=&s"(ex), [ev] "=&s"(ev), [eg] "=&s"(eg)
		: [TA0] "v"(S.TA0), [TA1] "v"(S.TA1), [TB0] "v"(S.TB0), [TB1] "v"(S.TB1), [ZWv] "v"(ZWv), [Mp] "s"(Mp), [Qp] "s"(Qp), [k22] "s"(k22),
		  [inc0] "s"(inc0), [inc1] "s"(inc1), [sp] "s"(sp0), [lim] "s"(lim), [st] "s"(st), [zq] "s"(zq), [c1] "s"(c1), [zd] "s"(zd), [pb] "s"(pbase)
		: "vcc", "scc", "memory");
	const int done = stat ? r + 1 - r_in : r - r_in;
	(void)qp; (void)qpb; (void)spB; (void)inB; (void)nge;
	S.qptr -= done;
	C.geLoM = ge; C.r = r;
	C.st0 = (r - E.w + 1) >> 1; C.en0 = (r + E.w) >> 1;
	return stat;
}

#undef IHP_PH_CELL
#undef IHP_PH_DIAG
#undef IHP_PH_EZ
#undef IHP_PH_NEXT
#undef IHP_PS_VEC
#undef IHP_PS_VEC_SAME_EN
#undef IHP_PS_PER
#undef IHP_PS_EZ
#undef IHP_PS_ZS2_NONE
#undef IHP_PS_ZS2_TAIL

template <bool HASB>
__device__ __forceinline__ void pair_steady_run(PairState &S, const PairEnv &E, PairCtl &C, const int zq, const int bound)
{
	C.lim = (S.inc0 | S.inc1) ? bound : -0x7fffffff - 1;
	if (HASB && !(E.w & 1)) {
		// the diagonals that compute block 4, by hand (pair_hasb_run_asm); what it hands back is finished here with the general code
		C.geLoM = ~0ull << (C.st0 - S.st); C.spM = 0ull; C.hiM = ~0ull;
		while (C.r < C.lim) {
			const int stat = pair_hasb_run_asm(S, E, C, zq);
			if (!stat) break;
			const int r = C.r, hiB = C.en0 - S.st - 64;
			const unsigned long long mInB = lane_range(0, hiB < 15 ? hiB : 15);
			if (stat & 3) pair_ez_lean<0, true>(S, E, C, r, C.geLoM, mInB, 0ull, zq);
			if (stat & 12) pair_ez_lean<1, true>(S, E, C, r, C.geLoM, mInB, 0ull, zq);
			if ((r + E.w) & 1) C.en0 += 1; else { C.st0 += 1; C.geLoM <<= 1; }
			C.r = r + 1;
			C.r = uni(C.r); C.st0 = uni(C.st0); C.en0 = uni(C.en0); C.lim = uni(C.lim); C.zsafe0 = uni(C.zsafe0); C.zsafe1 = uni(C.zsafe1);
			pair_uniform(S);
		}
		return;
	}
	if (C.r < C.lim && ((C.r + E.w) & 1)) pair_steady_step<HASB, 0, -1, -1>(S, E, C, zq);
	C.geLoM = ~0ull << (C.st0 - S.st);
	C.spM = HASB ? 0ull : 1ull << ((C.en0 - S.st) & 63); C.hiM = HASB ? ~0ull : C.spM | (C.spM - 1);
	if (!(E.w & 1) && !HASB) {
		// the everyday case by hand (pair_steady_pairs_asm); what it hands back is finished here with the general code
		while (C.r + 1 < C.lim) {
			const int stat = pair_steady_pairs_asm(S, E, C, zq);
			if (!stat) break;
			const int r = C.r;
			const unsigned long long inTA = C.geLoM & C.hiM;
			if (stat & 3) pair_ez_lean<0, false>(S, E, C, r, inTA, 0ull, C.spM, zq);
			if (stat & 12) pair_ez_lean<1, false>(S, E, C, r, inTA, 0ull, C.spM, zq);
			if (stat & 16) { C.en0 += 1; C.spM <<= 1; C.hiM = (C.hiM << 1) | 1ull; C.r = r + 1; }
			else {
				C.st0 += 1; C.geLoM <<= 1; C.r = r + 1;
				if (C.r < C.lim) pair_steady_step<false, 0, 1, 1>(S, E, C, zq);
			}
		}
	} else if (!(E.w & 1)) {
		while (C.r + 1 < C.lim) {
			pair_steady_step<HASB, 0, 0, 0>(S, E, C, zq);
			pair_steady_step<HASB, 0, 1, 1>(S, E, C, zq);
		}
	} else {
		while (C.r + 1 < C.lim) {
			pair_steady_step<HASB, 0, 0, -1>(S, E, C, zq);
			pair_steady_step<HASB, 0, 1, -1>(S, E, C, zq);
		}
	}
	if (C.r < C.lim) pair_steady_step<HASB, 0, -1, -1>(S, E, C, zq);
}

// The early and steady diagonals r .. r_hi-1 (see narrow_steady_loop): on return r is the next diagonal (or, when both
// alignments have z-dropped, one behind the last computed).
__device__ __forceinline__ void pair_steady_loop(PairState &S, const PairEnv &E, int &r, const int r_hi, long long *pacc = nullptr)
{
	const int lane = lane_id();
	const int w = E.w;
	const int zq = uni(E.zd >= 0x3fffffff ? 0x3fffffff : E.zd / E.qe);   // (the division runs on the vector unit)
	PairCtl C;
	C.r = r; C.st0 = (r - w + 1) >> 1; C.en0 = (r + w) >> 1;
	C.geLoM = C.spM = C.hiM = 0; C.lim = r_hi;
	C.zsafe0 = S.inc0 ? -1 : 0x7fffffff; C.zsafe1 = S.inc1 ? -1 : 0x7fffffff;
	if (r < w + 31) {                                    // the band grows from one cell (EDGE = 2)
		const long long te0 = pacc ? (long long)clock64() : 0;   // (diagnostics: the cycles of the early diagonals, as ksw_narrow.h's)
		C.st0 = C.st0 > 0 ? C.st0 : 0; C.en0 = C.en0 < r ? C.en0 : r;
		C.lim = r_hi < w + 31 ? r_hi : w + 31;
		if (!(w & 1)) {
			// by hand while block 4 is not computed and both alignments are live (pair_early_run_asm): r <= w in runs that end where a
			// block of sixteen cells enters (r = 16, 32, 48: the lanes above the computed blocks are cleared again behind a run), then
			// w < r up to the diagonal on which en0 reaches lane 64; what a run hands back is finished with the general code
			const int lim_all = C.lim;
			while (C.r < C.lim && S.inc0 && S.inc1) {
				const int r0 = uni(C.r), grow = r0 <= w;
				int e = grow ? (r0 | 15) + 1 : 128 - w;                     // (r + w) >> 1 >= 64 from r = 128 - w
				if (grow && e > w + 1) e = w + 1;
				e = e < lim_all ? e : lim_all;
				if (e - r0 < 2) break;
				const int st0 = grow ? 0 : (r0 - w + 1) >> 1, en0 = grow ? r0 : (r0 + w) >> 1, en = en0 | 15;
				C.lim = e;
				const unsigned long long rf = grow ? ~0ull >> (63 - en) : ~0ull << st0, tr = r0 < 64 ? 1ull << r0 : 0ull;
				const int stat = grow ? pair_early_run_asm<1>(S, E, C, zq, rf, 1ull << en0, (2ull << en0) - 1ull, tr)
				                      : pair_early_run_asm<0>(S, E, C, zq, rf, 1ull << en0, (2ull << en0) - 1ull, tr);
				if (en < 63) {
					const bool act = lane <= en;
					S.XA = act ? S.XA : 0u; S.VA = act ? S.VA : 0u; S.UA = act ? S.UA : 0u; S.YA = act ? S.YA : 0u;
				}
				const int rr = C.r;
				{ const int a = (rr - w + 1) >> 1, b = (rr + w) >> 1; C.st0 = a > 0 ? a : 0; C.en0 = b < rr ? b : rr; }
				C.lim = lim_all;
				if (stat) {
					const unsigned long long inTA = lane_range(C.st0, C.en0), spM = 1ull << C.en0;
					if (stat & 3) pair_ez_lean<0, false>(S, E, C, rr, inTA, 0ull, spM, zq);
					if (stat & 12) pair_ez_lean<1, false>(S, E, C, rr, inTA, 0ull, spM, zq);
					const int a = (rr - w + 2) >> 1, b = (rr + 1 + w) >> 1;
					C.r = rr + 1; C.st0 = a > 0 ? a : 0; C.en0 = b < rr + 1 ? b : rr + 1;
				}
				C.r = uni(C.r); C.st0 = uni(C.st0); C.en0 = uni(C.en0); C.lim = uni(C.lim); C.zsafe0 = uni(C.zsafe0); C.zsafe1 = uni(C.zsafe1);
				pair_uniform(S);
			}
		}
		while (uni(C.r) < uni(C.lim)) {
			if (uni(C.en0) < 64) pair_steady_step<false, 2, -1, -1>(S, E, C, zq);
			else pair_steady_step<true, 2, -1, -1>(S, E, C, zq);
		}
		C.r = uni(C.r); C.st0 = uni(C.st0); C.en0 = uni(C.en0); C.zsafe0 = uni(C.zsafe0); C.zsafe1 = uni(C.zsafe1);
		pair_uniform(S);
		if (pacc && lane == 0) pacc[4] += (long long)clock64() - te0;
	}
	while (C.r < r_hi && (S.inc0 | S.inc1)) {
		C.lim = r_hi;
		const bool moved = (C.st0 & ~15) != S.st;
		unsigned ex = 0, ev = 0;
		if (moved) {
			// the band origin moved one block right: close the traceback slot, rotate the registers 16 lanes, re-seed slot B
			if (C.r & 3) pair_flush(S, E, C.r - 1, S.st);
			ex = (unsigned)__builtin_amdgcn_readlane((int)S.XA, 15); ev = (unsigned)__builtin_amdgcn_readlane((int)S.VA, 15);
			S.edge_g = __builtin_amdgcn_readlane((int)S.GA, 15);
			const unsigned zf = pair_z(S.TB0, S.TB1, S.qptr[S.qoffB + 1]);  // qs[qlen-r+st+64+lane]: scores of diagonal r-1
			const unsigned zB = lane_in(0x7fffull) ? zf : E.ZWp;            // st0 was 16k+15, so last_sc = 78
			S.XA = rot16(S.XA, S.XB, lane); S.VA = rot16(S.VA, S.VB, lane);
			S.UA = rot16(S.UA, S.UB, lane); S.YA = rot16(S.YA, S.YB, lane);
			S.ZA = rot16(S.ZA, zB, lane);
			S.GA = rot16(S.GA, S.GB, lane);
			S.st = C.st0 & ~15;
			S.qptr += 16;
			S.XB = S.VB = S.UB = S.YB = 0; S.GB = 0;
			S.TA0 = pair_table(E, E.tg0[S.st + lane]); S.TA1 = pair_table(E, E.tg1[S.st + lane]);
			S.TB0 = pair_table(E, E.tg0[S.st + 64 + (lane & 15)]); S.TB1 = pair_table(E, E.tg1[S.st + 64 + (lane & 15)]);
		}
		int r_end = 2 * (S.st + 16) + w - 1;             // the next move
		r_end = r_end < r_hi ? r_end : r_hi;
		int r_b = 2 * (S.st + 64) - w;                   // block 4 from here on
		r_b = r_b > C.r ? r_b : C.r; r_b = r_b < r_end ? r_b : r_end;
		if (moved) {                                     // the diagonal of the move: lane 0 takes the block edge
			if (C.r < r_b) pair_steady_step<false, 1, -1, -1>(S, E, C, zq, ex, ev);
			else pair_steady_step<true, 1, -1, -1>(S, E, C, zq, ex, ev);
		}
		pair_steady_run<false>(S, E, C, zq, r_b);
		pair_steady_run<true>(S, E, C, zq, r_end);
	}
	r = C.r;
	// what these diagonals did not track: a slot-B lane was refreshed on the last one (st0 - st + 63 of diagonal r-1) or never
	S.rlB = lane <= ((r - w) >> 1) - S.st + 63 - 64 ? r - 1 : -1;
}

// ---------------------------------------------------------------- the tail: the band cut by the end of the query
// One diagonal behind the steady ones, once the band is narrower than 48 cells (narrow_tail_step of ksw_narrow.h on packed
// halves): st0 = r - qlen + 1 grows on every diagonal (the window never cuts the band of a pair), the band shrinks, never
// reaches block 4 (hiT <= 62) and refreshes no score past lane 62, so slot B only waits for the next move; only the computed
// lanes (blocks 0 .. (en0|15)-st) are committed.  H[st0] is the end-of-query result of every diagonal (:353-354).
// C.lim is pulled below r when the band leaves the matrix (:200-203) or both alignments have z-dropped.
template <bool EDGE>
__device__ __forceinline__ void pair_tail_step(PairState &S, const PairEnv &E, PairCtl &C, const int zq, const unsigned ex = 0, const unsigned ev = 0)
{
	const int INTMIN = -0x7fffffff - 1;
	const int st = S.st, r = C.r, st0 = C.st0, en0 = C.en0;
	const int loA = st0 - st, hiT = en0 - st;
	const int sc = loA + (((en0 - st0) >> 4) + 1) * 16 - 1;      // last refreshed score lane (:215): <= 62
	const int nTop = (en0 | 15) - st;                    // last computed lane: 15, 31, 47 or 63
	unsigned xpA = (unsigned)dppz_shr1((int)S.XA), vpA = (unsigned)dppz_shr1((int)S.VA), GpA = (unsigned)dppz_shr1((int)S.GA);   // neighbours of r-1
	if (EDGE) { xpA = (unsigned)set_lane0((int)ex, 0, (int)xpA); vpA = (unsigned)set_lane0((int)ev, 0, (int)vpA); }
	if (hiT == 0) GpA = (unsigned)set_lane0(S.edge_g, 0, (int)GpA);      // en0 on lane 0: H[en0-1] is the block edge
	const unsigned long long inTM = lane_span(loA, hiT);
	S.qptr -= 1;
	{
		const unsigned znew = pair_z(S.TA0, S.TA1, S.qptr[1]);          // qs[qlen-1-r+st+lane]
		S.ZA = lane_in(lane_span(loA, sc)) ? znew : S.ZA;                  // :214-228 (the 16-byte stores run past en)
	}
	{
		unsigned xn, vn, un, yn, nib;
		pair_cell(S.ZA, xpA, vpA, S.UA, S.YA, E, xn, vn, un, yn, nib);
		const bool act = lane_in(lane_span(0, nTop));
		S.XA = act ? xn : S.XA; S.VA = act ? vn : S.VA; S.UA = act ? un : S.UA; S.YA = act ? yn : S.YA;
		S.accA = pk_mad<16>(S.accA, nib);                                  // :283
		const bool sp = lane_in(1ull << hiT);
		S.GA = pk_add(sp ? GpA : S.GA, pk_shr<8>(sp ? un : vn));           // :318, :323-329
	}
	if ((r & 3) == 3) pair_flush(S, E, r, st);
	{                                                                    // :353-354 (r - st0 == qlen - 1 on every diagonal here)
		const int g = __builtin_amdgcn_readlane((int)S.GA, loA);
		const int h0 = pk_half<0>(g) - r * E.qe, h1 = pk_half<1>(g) - r * E.qe;
		if (S.inc0 && h0 > S.mqe0) { S.mqe0 = h0; pair_cold_set(S, PC_MQE_T0, st0); }
		if (S.inc1 && h1 > S.mqe1) { S.mqe1 = h1; pair_cold_set(S, PC_MQE_T1, st0); }
	}
	{
		int stat;
		pair_ez_asm<0>(stat, S.GA, S.thr0, S.pos0, C.zsafe0, inTM, 1ull << hiT, r, st, zq, E.w, E.qlm1, S.inc0);
		pair_ez_asm<1>(stat, S.GA, S.thr1, S.pos1, C.zsafe1, inTM, 1ull << hiT, r, st, zq, E.w, E.qlm1, S.inc1);
		if (stat) {
			if (stat & 3) pair_ez_lean<0, false>(S, E, C, r, inTM, 0ull, 1ull << hiT, zq);
			if (stat >> 2) pair_ez_lean<1, false>(S, E, C, r, inTM, 0ull, 1ull << hiT, zq);
		}
	}
	// ---- the band of r + 1 (:196-205) ----
	{
		const int a = r + 2 - E.qlen, b = (r + 2 - E.w) >> 1;
		C.st0 = a > b ? a : b; C.en0 = (r + 1 + E.w) >> 1; C.r = r + 1;
		if (C.st0 > C.en0) C.lim = INTMIN;                                 // :200-203
	}
}

// A run of tail diagonals C.r .. C.lim-1 within which the band origin, the number of computed blocks and the number of
// refreshed 16-byte score groups do not change (narrow_tail_qrun): every lane set moves by a bit or two per diagonal.
// FULL: blocks 0..3 are all computed (nothing to hold back).  The caller has done the diagonal of a move.
template <bool FULL>
__device__ __forceinline__ void pair_tail_qrun(PairState &S, const PairEnv &E, PairCtl &C, const int zq)
{
	const int st = S.st;
	const int nTop = (C.en0 | 15) - st;                  // last computed lane
	const unsigned long long actM = ~0ull >> (63 - nTop);
	// the run by hand while both alignments are live (pair_tail_run_asm); what it hands back is finished with the general code
	while (!(E.w & 1) && S.inc0 && S.inc1 && C.r + 1 < C.lim) {
		const int loA = C.st0 - st, hiT = C.en0 - st;
		const int sc = loA + (((hiT - loA) >> 4) + 1) * 16 - 1;
		const int stat = pair_tail_run_asm(S, E, C, zq, ~0ull << loA, lane_span(loA, sc), 1ull << hiT, lane_span(0, hiT), loA);
		if (!FULL) {                                                     // the lanes above the computed blocks: zeros, as before the run
			const bool act = lane_in(actM);
			S.XA = act ? S.XA : 0u; S.VA = act ? S.VA : 0u; S.UA = act ? S.UA : 0u; S.YA = act ? S.YA : 0u;
		}
		const int r = C.r;
		C.st0 = r - E.qlen + 1; C.en0 = (r + E.w) >> 1;
		if (!stat) break;
		{
			const unsigned long long inTM = lane_span(C.st0 - st, C.en0 - st), spM = 1ull << (C.en0 - st);
			if (stat & 3) pair_ez_lean<0, false>(S, E, C, r, inTM, 0ull, spM, zq);
			if (stat & 12) pair_ez_lean<1, false>(S, E, C, r, inTM, 0ull, spM, zq);
			C.r = r + 1; C.st0 = r + 2 - E.qlen; C.en0 = (r + 1 + E.w) >> 1;
		}
	}
	if (C.r >= C.lim) return;
	int loA = C.st0 - st, hiT = C.en0 - st;
	int sc = loA + (((hiT - loA) >> 4) + 1) * 16 - 1;    // last refreshed score lane (:215): <= 62
	unsigned long long inTM = lane_span(loA, hiT), refM = lane_span(loA, sc), spM = 1ull << hiT;
	do {
		const int r = C.r;
		const unsigned xpA = (unsigned)dppz_shr1((int)S.XA), vpA = (unsigned)dppz_shr1((int)S.VA), GpA = (unsigned)dppz_shr1((int)S.GA);   // (no move: edge 0, :210)
		S.qptr -= 1;
		{
			const unsigned znew = pair_z(S.TA0, S.TA1, S.qptr[1]);      // qs[qlen-1-r+st+lane]
			S.ZA = lane_in(refM) ? znew : S.ZA;                            // :214-228
		}
		{
			unsigned xn, vn, un, yn, nib;
			pair_cell(S.ZA, xpA, vpA, S.UA, S.YA, E, xn, vn, un, yn, nib);
			if (FULL) { S.XA = xn; S.VA = vn; S.UA = un; S.YA = yn; }
			else { const bool act = lane_in(actM); S.XA = act ? xn : S.XA; S.VA = act ? vn : S.VA; S.UA = act ? un : S.UA; S.YA = act ? yn : S.YA; }
			S.accA = pk_mad<16>(S.accA, nib);                              // :283
			const bool sp = lane_in(spM);
			S.GA = pk_add(sp ? GpA : S.GA, pk_shr<8>(sp ? un : vn));       // :318, :323-329
		}
		if ((r & 3) == 3) pair_flush(S, E, r, st);
		{                                                                // :353-354
			const int g = __builtin_amdgcn_readlane((int)S.GA, loA);
			const int h0 = pk_half<0>(g) - r * E.qe, h1 = pk_half<1>(g) - r * E.qe;
			if (S.inc0 && h0 > S.mqe0) { S.mqe0 = h0; pair_cold_set(S, PC_MQE_T0, st + loA); }
			if (S.inc1 && h1 > S.mqe1) { S.mqe1 = h1; pair_cold_set(S, PC_MQE_T1, st + loA); }
		}
		{
			int stat;
			pair_ez_asm<0>(stat, S.GA, S.thr0, S.pos0, C.zsafe0, inTM, spM, r, st, zq, E.w, E.qlm1, S.inc0);
			pair_ez_asm<1>(stat, S.GA, S.thr1, S.pos1, C.zsafe1, inTM, spM, r, st, zq, E.w, E.qlm1, S.inc1);
			if (stat) {
				if (stat & 3) pair_ez_lean<0, false>(S, E, C, r, inTM, 0ull, spM, zq);
				if (stat >> 2) pair_ez_lean<1, false>(S, E, C, r, inTM, 0ull, spM, zq);
			}
		}
		// ---- the band of r + 1: st0 + 1, en0 + 1 from an odd r + w ----
		const int p = (r + E.w) & 1;
		inTM = bit_clear(inTM, loA); refM = bit_clear(refM, loA);
		loA += 1; sc += 1; refM = bit_set(refM, sc);
		hiT += p; spM <<= p; inTM |= spM;
		C.r = r + 1; C.st0 = st + loA; C.en0 = st + hiT;
	} while (C.r < C.lim);
}

// The diagonals behind the steady ones, r .. r_end-1 (r_end: the first diagonal without a cell).  While the band is still 48
// or more cells wide (a handful of diagonals) the general pair_diag does them; from then on pair_tail_step / pair_tail_qrun,
// run by run between two moves of the band origin.  On return r is one behind the last diagonal computed.
__device__ __forceinline__ void pair_tail_loop(PairState &S, const PairEnv &E, int &r, const int r_end)
{
	const int INTMIN = -0x7fffffff - 1;
	const int qlen = E.qlen, w = E.w;
	const int zq = uni(E.zd >= 0x3fffffff ? 0x3fffffff : E.zd / E.qe);   // (the division runs on the vector unit)
	for (; r < r_end; ++r) {
		int st0 = r - qlen + 1, en0 = (r + w) >> 1;
		const int sw = (r - w + 1) >> 1;
		st0 = st0 > sw ? st0 : sw; st0 = st0 > 0 ? st0 : 0;
		en0 = en0 < r ? en0 : r;
		// the lean steps need a band that is narrower than 48 cells and stays so: cut by the end of the query from here on
		if (en0 - st0 < 48 && r > w + 32 && (st0 & ~15) > 0 && r - qlen + 1 >= sw) break;
		pair_diag<false>(S, E, r, st0, en0);
		if (!(S.inc0 | S.inc1)) { ++r; return; }
	}
	r = uni(r);
	pair_uniform(S);
	if (r >= r_end) return;
	PairCtl C;
	C.r = r; C.geLoM = C.spM = C.hiM = 0;
	C.zsafe0 = S.inc0 ? -1 : 0x7fffffff; C.zsafe1 = S.inc1 ? -1 : 0x7fffffff;
	{
		const int a = r + 1 - qlen, b = (r + 1 - w) >> 1;
		C.st0 = a > b ? a : b; C.en0 = (r + w) >> 1;
	}
	bool out = false;
	while (C.r < r_end && !out) {
		const bool moved = (C.st0 & ~15) != S.st;
		unsigned ex = 0, ev = 0;
		if (moved) pair_move(S, E, C.r, C.st0 & ~15, ex, ev);
		// the next move: st0 = r - qlen + 1 reaches st + 16 (the other bound, (r-w+1)>>1, lags behind it in the tail)
		const int m1 = S.st + 15 + qlen, m2 = 2 * (S.st + 16) + w - 1;
		int r_move = m1 < m2 ? m1 : m2;
		r_move = r_move < r_end ? r_move : r_end;
		C.lim = r_move;
		if (moved) pair_tail_step<true>(S, E, C, zq, ex, ev);
		while (C.r < C.lim) {
			// a run up to the next diagonal on which the band origin, the computed blocks (en0 crosses a multiple of 16) or the
			// refreshed score groups ((en0-st0)>>4 drops) change
			const int rr = C.r, wd = C.en0 - C.st0, p = (rr + w) & 1;
			const int r_k = rr + 2 * ((wd & 15) + 1) - 1 + p;            // width falls to 16k - 1
			const int r_e = rr + 2 * (16 - (C.en0 & 15)) - p;             // en0 reaches the next multiple of 16
			int e = r_k < r_e ? r_k : r_e; e = e < C.lim ? e : C.lim;
			if (e > rr) {
				const int keep = C.lim;
				C.lim = e;
				if (((C.en0 | 15) - S.st) == 63) pair_tail_qrun<true>(S, E, C, zq); else pair_tail_qrun<false>(S, E, C, zq);
				if (C.lim != INTMIN) C.lim = keep;
				continue;
			}
			pair_tail_step<false>(S, E, C, zq);
		}
		if (C.lim == INTMIN) out = true;                 // both z-dropped, or the band has left the matrix
	}
	r = C.r;
}

// What the sweep of a pair leaves for the traceback (every field of ksw_extz_t the sweep decides, per alignment).
struct PairResult { int max0, max1, pos0, pos1, mqe0, mqe1, mqe_t0, mqe_t1; };

// The sweep of two jobs with the same qlen and parameters, both ksw_pair_job_ok(): the traceback slots in p, the results
// in R.  Returns false when a sequence holds a code the pair sweep does not take (a wildcard or worse in a query, anything
// above the wildcard in a target): the plan kernel only pairs jobs whose producer vouches for the codes.
// Nothing but (lds, p) and what is in R is needed afterwards: the caller reads the jobs again for the tracebacks, so that
// none of their fields occupies a scalar register across the sweep (the kernel is short of them: 100 per wavefront).
__device__ inline bool ksw_pair_sweep(const uint8_t *q0, const uint8_t *t0, int tlen0, const uint8_t *q1, const uint8_t *t1, int tlen1, int qlen,
                                      int w, int q, int e, int sc_mch, int sc_mis, int zdrop, int encode_ascii, uint8_t *lds, uint8_t *p, PairResult &R, long long *pacc = nullptr)
{
	const long long tc0 = pacc ? (long long)clock64() : 0;
	const int lane = lane_id();
	// every scalar a register of its own: the launch arguments arrive four to a load, and a register tuple is kept -- and
	// spilled, and reloaded inside the loops -- as long as any part of it is needed
	w = uni(w); q = uni(q); e = uni(e); sc_mch = uni(sc_mch); sc_mis = uni(sc_mis); zdrop = uni(zdrop); qlen = uni(qlen); tlen0 = uni(tlen0); tlen1 = uni(tlen1);
	asm volatile("" : "+s"(w), "+s"(q), "+s"(e), "+s"(sc_mch), "+s"(sc_mis), "+s"(zdrop), "+s"(qlen), "+s"(tlen0), "+s"(tlen1));
	const int qe = q + e;
	const int TP0 = (tlen0 + 15) / 16 * 16 + 96, TP1 = (tlen1 + 15) / 16 * 16 + 96, QR = (qlen + 15) / 16 * 16 + 96;
	uint8_t *tg0 = lds + 64, *tg1 = tg0 + TP0;
	unsigned *qs = (unsigned *)(tg1 + TP1) + 16;
	bool bad = false;
	for (int i = lane; i < TP0; i += 64) {
		uint8_t b = 0;
		if (i < tlen0) { b = t0[i]; if (encode_ascii) b = enc_base(b); }
		bad |= b > 4;
		tg0[i] = b;
	}
	for (int i = lane; i < TP1; i += 64) {
		uint8_t b = 0;
		if (i < tlen1) { b = t1[i]; if (encode_ascii) b = enc_base(b); }
		bad |= b > 4;
		tg1[i] = b;
	}
	if (lane < 16) qs[lane - 16] = 0x040c000cu;
	for (int i = lane; i < QR; i += 64) {
		unsigned c0 = 0, c1 = 0;
		if (i < qlen) {
			c0 = q0[qlen - 1 - i]; c1 = q1[qlen - 1 - i];
			if (encode_ascii) { c0 = enc_base((uint8_t)c0); c1 = enc_base((uint8_t)c1); }
		}
		bad |= c0 > 3 || c1 > 3;
		qs[i] = 0x000c000cu | c0 << 8 | (4 + c1) << 24;
	}
	if (ballot(bad)) return false;
	WSYNC();
	if (pacc && lane == 0) { const long long tc1 = (long long)clock64(); pacc[0] += tc1 - tc0; pacc[1] -= tc1; }

	const unsigned ZW = (unsigned)(2 * qe) & 0xff, ZM = (unsigned)(2 * qe + sc_mch) & 0xff, ZX = (unsigned)(2 * qe + sc_mis) & 0xff;
	PairEnv E;
	E.tg0 = tg0; E.tg1 = tg1; E.qs = qs; E.p = (unsigned *)p; E.qlen = qlen; E.qlm1 = qlen - 1; E.w = w; E.qe = qe; E.e = e;
	E.zd = zdrop < 0 ? 0x3fffffff : zdrop;
	E.Qp = ((unsigned)q & 0xff) * 0x01000100u; E.Mp = ZM * 0x01000100u; E.ZWp = ZW * 0x01000100u; E.QE2p = (unsigned)(2 * qe) * 0x00010001u;
	E.zx4 = ZX * 0x01010101u; E.zdm = ZM - ZX; E.zw4 = ZW * 0x01010101u;
	PairState S;
	S.XA = S.VA = S.UA = S.YA = 0; S.ZA = E.ZWp; S.GA = 0;
	S.XB = S.VB = S.UB = S.YB = 0; S.GB = 0;
	S.TA0 = pair_table(E, tg0[lane]); S.TA1 = pair_table(E, tg1[lane]);
	S.TB0 = pair_table(E, tg0[64 + (lane & 15)]); S.TB1 = pair_table(E, tg1[64 + (lane & 15)]);
	S.qptr = qs + (qlen - 1 + lane); S.qoffB = 64 + (lane & 15) - lane; S.rlB = -1;
	S.accA = S.accB = 0; S.st = 0; S.edge_g = 0;
	S.thr0 = S.thr1 = 0; S.pos0 = S.pos1 = pos_pack(-2, -1);             // max_t = max_q = -1 (:81-86)
	S.mqe0 = S.mqe1 = KSW_NEG_INF; S.cold = lane < 2 ? -1 : 0;           // mqe_t = -1, fin = 0
	S.inc0 = S.inc1 = qe;
	// the band leaves the matrix on diagonal 2 qlen + w - 1 (:200-203): r_end is the first diagonal without a cell
	const int r_end = 2 * qlen + w - 1;
	pair_diag<true>(S, E, 0, 0, 0);
	pair_uniform(S);
	int r = 1;
	// steady diagonals: st0 = (r-w+1)>>1 > r-qlen+1 (the window never cuts the band): up to 2 qlen - w - 3
	pair_steady_loop(S, E, r, 2 * qlen - w - 2, pacc);
	{
		const long long tt0 = pacc ? (long long)clock64() : 0;
		if (S.inc0 | S.inc1) pair_tail_loop(S, E, r, r_end);
		if (pacc && lane == 0) pacc[5] += (long long)clock64() - tt0;
	}
	// r - 1 is the last diagonal whose cells were computed; thr* stand at diagonal r
	if (((r - 1) & 3) != 3) pair_flush(S, E, r - 1, S.st);
	WSYNC();
	if (pacc && lane == 0) { pacc[1] += (long long)clock64(); pacc[3] += 2; }
	R.max0 = S.inc0 ? S.thr0 - r * qe : pair_cold_get(S, PC_FIN0); R.max1 = S.inc1 ? S.thr1 - r * qe : pair_cold_get(S, PC_FIN1);
	R.pos0 = S.pos0; R.pos1 = S.pos1; R.mqe0 = S.mqe0; R.mqe1 = S.mqe1; R.mqe_t0 = pair_cold_get(S, PC_MQE_T0); R.mqe_t1 = pair_cold_get(S, PC_MQE_T1);
	return true;
}

// The record and the CIGAR of alignment K of a pair from its sweep's results (ksw_backtrack_wave on the shared slots).
template <int K>
__device__ __forceinline__ void ksw_pair_finish(const PairResult &R, const uint8_t *p, int qlen, int tlen, int w, int flag, uint32_t *cig_tmp, int cig_cap, KswOut &out, long long *pacc = nullptr)
{
	// every exit is a z-drop for the caller (:98-101, :200-203); a sweep that stopped has no score (:355-357) and the window's end is never reached
	out.zdropped = 1; out.mte = out.score = KSW_NEG_INF; out.mte_q = -1; out.n_cigar = 0;
	out.max = K == 0 ? R.max0 : R.max1;
	const int pos = K == 0 ? R.pos0 : R.pos1;
	out.max_t = pos_t(pos); out.max_q = pos_q(pos);
	out.mqe = K == 0 ? R.mqe0 : R.mqe1; out.mqe_t = K == 0 ? R.mqe_t0 : R.mqe_t1;
	const long long tb0 = pacc ? (long long)clock64() : 0;
	ksw_backtrack_wave<2, K>(p, 0, qlen, tlen, w, flag, 1, out.max_t, out.max_q, cig_tmp, cig_cap, out);
	if (pacc && lane_id() == 0) pacc[2] += (long long)clock64() - tb0;
}

}  // namespace ihp
