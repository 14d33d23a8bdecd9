// ksw_narrow.h -- the production ksw2 sweep: register-resident like ksw_fast.h, with the int8 work values kept
// in the TOP BYTE of a 32-bit register and everything wave-uniform (band limits, lane masks, the running
// maximum) computed on the scalar unit.  Reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388.
//
// ksw2's u, v, x, y, s are int8 in the reference and wrap (the padding cells of the 16-rounded band do wrap:
// their y climbs by u-(max_sc-q) per diagonal); ksw_fast.h masks and sign-extends around every operation.
// With value << 24 in a register, 32-bit add/sub wrap exactly like _mm_add_epi8/_mm_sub_epi8, signed 32-bit
// compares are the int8 compares (_mm_cmpgt_epi8) and unsigned 32-bit max/min are _mm_max_epu8/_mm_min_epu8
// (:131-132, :271-272): the cell needs no masks at all.  Only the H update reads a value back (>> 24, as
// the reference's uint8_t u8/v8 of :193).
//
// The score byte s[t] (stale outside the refreshed 16-byte groups, :214-228) is carried as z = s + 2(q+e);
// a fresh z is one v_perm_b32 into the lane's 8-byte table {0, z(A), z(C), z(G) | z(T), z(N)} for its
// target base, with the selector word ((query code + 1) << 24 | 0x0c0c0c: table byte into the top byte,
// zeros below) read from LDS.  Lane masks for "refreshed", "inside the true band", "computed" come from
// scalar shifts and are consumed directly as v_cndmask conditions.
//
// Traceback matrix.  The reference stores one byte per cell (:283); here the four compares that make that byte are
// shifted into a per-lane accumulator (one v_addc each, the compare's lane mask as carry-in), eight diagonals to a
// dword, and a dword per lane is stored every eighth diagonal: slot (r >> 3) + (st >> 4) of 80 dwords (lanes of slot
// A, then the 16 of slot B) holds diagonal r's nibble at bits 4*(7 - (r & 7)).  A move of the band origin closes the
// slot early (st changes, so the slot index does too); moves are at least 16 diagonals apart, so a group of eight
// diagonals spans at most two slots.  The walk back decodes a nibble (c1 c2 c3 c4) into the reference's byte.
//
// Exact maximum.  ksw2 needs max H of every diagonal (:312-349) only to update ez.max and to test the z-drop (:88-104).
// Both are decided by lane compares against two wave-uniform thresholds: no lane above ez.max and some lane at or above
// ez.max - zdrop means nothing happens on this diagonal; exactly one lane above ez.max is the new maximum (one
// v_readlane); only ties, several improving lanes or a possible z-drop run the DPP reduction and the tie order.
#pragma once
#include "ksw_fast.h"

namespace ihp {

__host__ __device__ __forceinline__ size_t ksw_narrow_lds_bytes(int qlen, int tlen)
{   // z table + target codes + one selector word per (padded) query position
	return 64 + (size_t)((tlen + 15) / 16) * 16 + 96 + 16 + 4 * ((size_t)((qlen + 15) / 16) * 16 + 96 + 16);
}

// What this sweep covers: the register layout's band, a 5-letter alphabet with the wildcard last, and
// z = s + 2(q+e) > 0 as int8 for every s (then :271's clamp of z is the identity and d can use the same z).
__host__ __device__ __forceinline__ bool ksw_narrow_ok(const KswParams &P)
{
	const int qe2 = 2 * (P.q + P.e);
	const int zm = (int)(signed char)((qe2 + P.sc_mch) & 0xff), zx = (int)(signed char)((qe2 + P.sc_mis) & 0xff);
	const int zw = (int)(signed char)(qe2 & 0xff);
	return P.w >= 0 && P.w <= 62 && P.m == 5 && zm > 0 && zx > 0 && zw > 0;
}

__device__ __forceinline__ int dppz_shr1(int v)
{   // lane l gets v[l-1]; lane 0 gets 0
	return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true);
}

__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// v with lane 0 replaced by the wave-uniform s (v_writelane_b32: one VALU slot; a v_cndmask would need the mask
// and the value on the constant bus at once)
__device__ __forceinline__ int set_lane0(int s, int /*lane*/, int v)
{
	asm("v_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(s)));
	return v;
}

// lanes [lo, hi] of the wave, 0 <= lo, hi <= 63; empty when hi < lo
__device__ __forceinline__ unsigned long long lane_range(int lo, int hi)
{
	return hi < lo ? 0ull : ((~0ull << lo) & (~0ull >> (63 - hi)));
}

// max over the 64 lanes without clobbering the input: xor 1, xor 2, 8-lane mirror and 16-lane mirror leave each
// row's max in all its lanes; row_bcast:15 / row_bcast:31 then fold rows 0->1, 2->3 and 1->2,3: lane 63 has it.
__device__ __forceinline__ int wave_max_i32_keep(int v)
{
	int t;
	asm("s_nop 4\n\t"         // covers VALU-writes-EXEC -> DPP (5 wait states) as well as VGPR -> DPP (2)
	    "v_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 1\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	    "s_nop 1"
	    : "=&v"(t) : "v"(v));
	return __builtin_amdgcn_readlane(t, 63);
}

// bytes of traceback scratch for one job (see "Traceback matrix" above)
__host__ __device__ __forceinline__ size_t ksw_narrow_p_bytes(int qlen, int tlen)
{
	const int total = qlen + tlen - 1;
	return total <= 0 ? 16 : ((size_t)(total >> 3) + (size_t)(tlen >> 4) + 3) * 320;
}

// acc * 2 + (this lane's bit of m): one VALU instruction
__device__ __forceinline__ unsigned shl1_in(unsigned acc, unsigned long long m)
{
	asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(acc) : "s"(m) : "vcc");
	return acc;
}

// One cell (:116-137 + :262-310); every value is (int8 << 24).  z > 0, so :271 is the identity.  The four compares
// behind the traceback byte (:265/:291, :273/:299, :277-282/:303-308) are appended to `acc`, first compare highest.
template <bool RIGHT>
__device__ __forceinline__ void narrow_cell(int z, int xp, int vp, int u, int y, unsigned M, int q,
                                            int &xn, int &vn, int &un, int &yn, unsigned &acc)
{
	const int a = xp + vp, b = y + u;
	acc = shl1_in(acc, ballot(RIGHT ? !(z > a) : a > z));                // :265, :291
	unsigned zz = (unsigned)z > (unsigned)a ? (unsigned)z : (unsigned)a;    // :272 _mm_max_epu8
	acc = shl1_in(acc, ballot(RIGHT ? !((int)zz > b) : b > (int)zz));    // :273-274, :299-300
	zz = zz > (unsigned)b ? zz : (unsigned)b;                           // :131
	zz = zz < M ? zz : M;                                               // :132
	un = (int)zz - vp; vn = (int)zz - u;                                // :133-134
	const int zq = (int)zz - q;
	const int a2 = a - zq, b2 = b - zq;
	xn = a2 > 0 ? a2 : 0; yn = b2 > 0 ? b2 : 0;                         // :277-282, :303-308
	acc = shl1_in(acc, ballot(RIGHT ? !(a2 < 0) : a2 > 0));
	acc = shl1_in(acc, ballot(RIGHT ? !(b2 < 0) : b2 > 0));
}

// the reference's byte (:283) from a nibble of the accumulator
__device__ __forceinline__ unsigned narrow_p_byte(unsigned nib)
{
	return ((nib & 4) ? 2u : (nib >> 3)) | ((nib & 2) << 2) | ((nib & 1) << 4);
}

struct NarrowState {
	// per lane: slot A (t = st + lane), slot B (t = st + 64 + lane, lanes 0..15)
	int XA, VA, UA, YA, ZA, HA;
	unsigned T1A, T0A;
	int XB, VB, UB, YB, HB, rlB;
	const unsigned *qptr;                                // LDS: this lane's selector word for the coming diagonal
	int qoffB;                                           // 64 + (lane & 15) - lane: from qptr to the lane's slot-B word
	unsigned T1B, T0B;
	unsigned accA, accB;                                 // traceback nibbles: A shifted in, B placed at 4*(7 - (r & 7))
	// wave-uniform
	int st, edge_h, last_sc, band_exit;
	int ez_max, ez_max_t, ez_max_q, mqe, mqe_t, mte, mte_q, score;
};

struct NarrowEnv {
	const uint8_t *tg;                                   // LDS: target codes
	const unsigned *qs;                                  // LDS: selector words of the reversed query
	const uint2 *tbl;                                    // LDS: z table per target code
	unsigned *p;                                         // traceback slots of 80 dwords
	uint8_t *pb; int ncol;                               // ksw_wide.h: the reference's byte matrix, ncol bytes per diagonal
	int qlen, tlen, w, qe, e, zdrop;
	int q24, ZW24;                                       // q << 24, z of a never-refreshed cell << 24
	unsigned M24;                                        // max_sc << 24
	long long *pacc;                                     // diagnostics: per-wave cycle counters or null ([4] early, [5] tail diagonals)
};

__device__ __forceinline__ int narrow_z(unsigned T0, unsigned T1, unsigned sel) { return (int)__builtin_amdgcn_perm(T0, T1, sel); }

// Close the traceback slot of diagonals ..r_last (band origin st): r_last & 7 == 7 for a full group.
__device__ __forceinline__ void narrow_flush(NarrowState &F, const NarrowEnv &E, int r_last, int st)
{
	const int lane = lane_id();
	unsigned *row = E.p + (size_t)((r_last >> 3) + (st >> 4)) * 80;
	row[lane] = F.accA << (4 * (7 - (r_last & 7)));
	if (lane < 16) row[64 + lane] = F.accB;
	F.accB = 0;
}

// max_t of the exact maximum with the reference's tie order (:320-348): en0 first, then the stride classes of the
// vector part, then the scalar tail.  hA / hB hold INT_MIN outside the true band.
__device__ __forceinline__ int narrow_max_t(int hA, int hB, int max_H, bool hasB, unsigned long long mInB, int loA, int st, int st0, int en0)
{
	const unsigned long long mA = ballot(hA == max_H);
	const unsigned long long mB = hasB ? ballot(hB == max_H) & mInB : 0ull;
	if (popc64(mA) + popc64(mB) == 1) return mA ? st + ctz64(mA) : st + 64 + ctz64(mB);
	const unsigned long long m = loA ? ((mA >> loA) | (mB << (64 - loA))) : mA;   // bit i <-> t = st0 + i
	const int ie = en0 - st0, nv = (en0 - st0) / 4 * 4;
	if ((m >> ie) & 1) return en0;
	const unsigned long long mv = nv ? (m & ((1ull << nv) - 1)) : 0ull;
	int max_t = en0;
	if (mv) {
		for (int j = 0; j < 4; ++j) {
			const unsigned long long cm = mv & (0x1111111111111111ull << j);
			if (cm) { max_t = st0 + ctz64(cm); break; }
		}
	} else {
		const unsigned long long mt = m & ~mv;
		if (mt) max_t = st0 + ctz64(mt);
	}
	return max_t;
}

// One anti-diagonal outside the steady loop (narrow_steady_loop below); same contract as fast_diag() in ksw_fast.h
// (returns true when the sweep must stop; every such exit is a z-drop for the caller, :98-101 and :200-203).  KIND:
//   ND_ANY    any r >= 1;
//   ND_FIRST  r == 0 (the same code with r folded);
//   ND_EARLY  1 <= r <= w+30 of a job with qlen, tlen >= w+32: the band starts in block 0 (st == 0, no move), is
//             not yet cut by the sequence ends, and grows: st0 = max(0,(r-w+1)>>1), en0 = min(r,(r+w)>>1).
// A block edge x[st-1], v[st-1] is taken (:207-208) exactly on the diagonal where st moves: then st-1 = old st+15
// lies in [last_st, last_en]; without a move st-1 < last_st and the edge is 0 (:210).
// Slot B's score bytes are not tracked per diagonal while the refreshed range only grows (st0, en0 and en0-st0
// non-decreasing: ND_EARLY and the steady loop): then a B lane is "refreshed on the previous diagonal" up to
// last_sc-64 and "never" above, which is all its lazily formed value needs.
enum { ND_ANY = 0, ND_EARLY = 2, ND_FIRST = 3 };

template <bool RIGHT, int KIND>
__device__ __forceinline__ bool narrow_diag(NarrowState &F, const NarrowEnv &E, const int r)
{
	constexpr bool EARLY = KIND == ND_EARLY;
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	int st0, en0, nst, en;
	if (EARLY) {
		st0 = (r - E.w + 1) >> 1; st0 = st0 > 0 ? st0 : 0;
		en0 = (r + E.w) >> 1; en0 = en0 < r ? en0 : r;
		nst = 0; en = en0 | 15;
	} else if (!ksw_band(r, E.qlen, E.tlen, E.w, st0, en0, nst, en)) { F.band_exit = 1; return true; }   // :200-203
	int ex = 0, ev = 0;
	if (!EARLY && nst != F.st) {
		// the band origin moved one block right: close the traceback slot, rotate the registers 16 lanes, re-seed slot B
		if (r & 7) narrow_flush(F, E, r - 1, F.st);
		ex = __builtin_amdgcn_readlane(F.XA, 15);
		ev = __builtin_amdgcn_readlane(F.VA, 15);
		F.edge_h = __builtin_amdgcn_readlane(F.HA, 15);
		const int zB = F.rlB < 0 ? E.ZW24 : narrow_z(F.T0B, F.T1B, E.qs[E.qlen - 1 - F.rlB + F.st + 64 + (lane & 15)]);
		F.XA = (int)rot16((unsigned)F.XA, (unsigned)F.XB, lane); F.VA = (int)rot16((unsigned)F.VA, (unsigned)F.VB, lane);
		F.UA = (int)rot16((unsigned)F.UA, (unsigned)F.UB, lane); F.YA = (int)rot16((unsigned)F.YA, (unsigned)F.YB, lane);
		F.ZA = (int)rot16((unsigned)F.ZA, (unsigned)zB, lane);
		F.HA = (int)rot16((unsigned)F.HA, (unsigned)F.HB, lane);
		F.st = nst;
		F.qptr += 16;
		F.XB = F.VB = F.UB = F.YB = 0; F.HB = KSW_NEG_INF;
		F.rlB = -1;
		const uint2 ta = E.tbl[E.tg[nst + lane]], tb = E.tbl[E.tg[nst + 64 + (lane & 15)]];
		F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
	} else if (EARLY || F.st == 0) { ev = r ? E.q24 : 0; }               // :211
	// neighbours of r-1 (taken before anything is overwritten); lane 0 gets the block edge x1, v1 (:207-211)
	const int xpA = EARLY ? dppz_shr1(F.XA) : set_lane0(ex, 0, dppz_shr1(F.XA)), vpA = set_lane0(ev, 0, dppz_shr1(F.VA));
	int HpA = dppz_shr1(F.HA);
	if (!EARLY) HpA = set_lane0(F.edge_h, 0, HpA);      // an early band never has en0 on lane 0 (r >= 1)
	const int st = EARLY ? 0 : F.st;
	const int loA = st0 - st;                            // first true-band lane (<= 15)
	const int hiT = en0 - st;                            // last true-band lane (may be >= 64: slot B)
	const int nTop = en - st;                            // last computed lane: 15, 31, 47, 63 or 79
	const int sc = st0 + ((en0 - st0) / 16 + 1) * 16 - 1 - st;   // last refreshed score lane (:215), > loA
	const int hasB = (nTop >> 6) & 1;                    // block 4 is computed (an integer: a bool carried across the
	                                                     // DPP asm below is materialised per lane and tested again)
	const bool has_spec = EARLY || (r > 0 && en0 > 0);   // H[en0] comes from H[en0-1] + u (:318)
	const bool geLo = lane >= loA;
	const bool refA = geLo && lane <= sc;                               // refreshed score lanes (:214-228); sc > loA
	const bool inTA = geLo && lane <= hiT;                              // inside the true band (hiT >= loA)
	const bool spA = has_spec && lane == hiT;                           // in slot B if hiT >= 64
	int hB = INTMIN, hA;
	unsigned long long mInB = 0;
	// ---- slot B (block 4) ------------------------------------------------------------
	if (!EARLY) F.rlB = lane_in(sc >= 64 ? ~0ull >> (127 - sc) : 0ull) ? r : F.rlB;  // :214-228 runs past en; value formed on use
	if (hasB) {                                                        // nTop == 79: the whole block
		const int exB = __builtin_amdgcn_readlane(F.XA, 63), evB = __builtin_amdgcn_readlane(F.VA, 63);
		const int HeB = __builtin_amdgcn_readlane(F.HA, 63);
		int xpB = dppz_shr1(F.XB), vpB = dppz_shr1(F.VB), HpB = dppz_shr1(F.HB);
		xpB = set_lane0(exB, 0, xpB); vpB = set_lane0(evB, 0, vpB);
		HpB = set_lane0(HeB, 0, HpB);
		int zB;
		if (EARLY) {
			const int zf = narrow_z(F.T0B, F.T1B, F.qptr[F.qoffB]);        // qs[qlen-1-r+st+64+lane]
			zB = lane_in(lane_range(0, sc - 64)) ? zf : E.ZW24;
		} else zB = F.rlB < 0 ? E.ZW24 : narrow_z(F.T0B, F.T1B, E.qs[E.qlen - 1 - F.rlB + st + 64 + (lane & 15)]);
		mInB = lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15);           // loA <= 15, so block 4 is never below the band
		if (lane < 16) {
			int ut = F.UB, yt = F.YB;
			if (st + 64 + lane == r) { yt = 0; ut = r ? E.q24 : 0; }   // :212
			int xn, vn, un, yn; unsigned nib = 0;
			narrow_cell<RIGHT>(zB, xpB, vpB, ut, yt, E.M24, E.q24, xn, vn, un, yn, nib);
			F.XB = xn; F.VB = vn; F.UB = un; F.YB = yn;
			F.accB |= nib << (4 * (7 - (r & 7)));                      // :283
			const bool sp = has_spec && 64 + lane == hiT;
			const int h = (sp ? HpB : F.HB) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329 (u8, v8 are uint8_t: :193)
			const bool inT = lane_in(mInB);
			hB = inT ? h : INTMIN;
			F.HB = inT ? h : F.HB;
		}
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	{
		const int znew = narrow_z(F.T0A, F.T1A, *F.qptr);            // qs[qlen-1-r+st+lane]
		F.qptr -= 1;
		F.ZA = refA ? znew : F.ZA;                                      // :214-228
		if (r <= en && r - st < 64) {                                  // :212 (only while the band still touches t == r)
			const bool tr = lane_in(1ull << (r - st));
			F.YA = tr ? 0 : F.YA; F.UA = tr ? (r ? E.q24 : 0) : F.UA;
		}
		int xn, vn, un, yn;
		narrow_cell<RIGHT>(F.ZA, xpA, vpA, F.UA, F.YA, E.M24, E.q24, xn, vn, un, yn, F.accA);   // :283 (lanes past nTop: never read)
		int h;
		const bool sp = spA;
		if (KIND != ND_FIRST) h = (sp ? HpA : F.HA) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329
		else h = (int)((unsigned)vn >> 24) - E.qe - E.qe;              // :349
		const bool act = lane_in(~0ull >> (63 - (nTop < 63 ? nTop : 63)));
		F.XA = act ? xn : F.XA; F.VA = act ? vn : F.VA; F.UA = act ? un : F.UA; F.YA = act ? yn : F.YA;
		hA = inTA ? h : INTMIN;
		F.HA = inTA ? h : F.HA;
	}
	if ((r & 7) == 7) narrow_flush(F, E, r, st);
	F.last_sc = sc;
	// ---- ez updates (:351-357) -----------------------------------------------------------
	if (!EARLY) {
		int Hen0 = 0;
		if (en0 == E.tlen - 1 || r - st0 == E.qlen - 1) {
			Hen0 = hiT < 64 ? __builtin_amdgcn_readlane(hA, hiT & 63) : __builtin_amdgcn_readlane(hB, (hiT - 64) & 63);
			const int Hst0 = __builtin_amdgcn_readlane(hA, loA);
			if (en0 == E.tlen - 1 && Hen0 > F.mte) { F.mte = Hen0; F.mte_q = r - en; }        // rounded en (:352)
			if (r - st0 == E.qlen - 1 && Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st0; }
		}
		if (r == E.qlen + E.tlen - 2 && en0 == E.tlen - 1) F.score = Hen0;                  // :356-357
	}
	// ---- exact max (:312-349) and ksw_apply_zdrop (:88-104), decided by lane compares where that is enough ----
	const unsigned long long mIA = ballot(hA > F.ez_max), mIB = hasB ? ballot(hB > F.ez_max) : 0ull;
	if (mIA | mIB) {
		int max_H, max_t;
		if (popc64(mIA) + popc64(mIB) == 1) {
			const int i = mIA ? ctz64(mIA) : ctz64(mIB);
			max_H = mIA ? __builtin_amdgcn_readlane(hA, i) : __builtin_amdgcn_readlane(hB, i);
			max_t = st + i + (mIA ? 0 : 64);
		} else {
			max_H = wave_max_i32_keep(hA);
			if (hasB) { const int mb = wave_max_i32_keep(hB); max_H = mb > max_H ? mb : max_H; }
			max_t = narrow_max_t(hA, hB, max_H, hasB, mInB, loA, st, st0, en0);
		}
		F.ez_max = max_H; F.ez_max_t = max_t; F.ez_max_q = r - max_t;
		return false;
	}
	if (E.zdrop < 0) return false;
	{
		const int thz = F.ez_max - E.zdrop;
		if (ballot(hA >= thz) | (hasB ? ballot(hB >= thz) : 0ull)) return false;   // ez.max - max_H <= zdrop: :98 cannot hold
	}
	int max_H = wave_max_i32_keep(hA);
	if (hasB) { const int mb = wave_max_i32_keep(hB); max_H = mb > max_H ? mb : max_H; }
	const int t = narrow_max_t(hA, hB, max_H, hasB, mInB, loA, st, st0, en0), dq = r - t;
	if (t < F.ez_max_t || dq < F.ez_max_q) return false;
	const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
	const int l = tl > ql ? tl - ql : ql - tl;
	return F.ez_max - max_H > E.zdrop + l * E.e;
}

// What the steady loop carries from diagonal to diagonal besides NarrowState (all wave-uniform, in SGPRs).
struct SteadyCtl {
	int r, st0, en0;                                     // the coming diagonal and its true band
	int thrI;                                            // ez.max + r (q+e): the H' a cell must beat to improve the maximum
	int lim;                                             // the run's end; a z-drop pulls it below r so that every loop winds down
	int stop_r;                                          // the diagonal a z-drop stopped on, or -1
	unsigned long long geLoM, spM;                       // lanes >= st0 - st; the lane of en0 (slot A)
	int zd;                                              // zdrop, or a value no score difference reaches when there is none
};

// One steady diagonal (see narrow_steady_loop) and the step to the next.  HASB = block 4 is computed (hiT >= 64);
// EDGE = 1: the band origin moved on this diagonal and lane 0 takes the block edge ex, ev (:207-208; without a move the
// edge is 0, :210).  EDGE = 2: a diagonal 1 <= r <= w+30 of a job with qlen, tlen >= w+32 -- the band still starts in
// block 0 (lane 0 takes x1 = 0, v1 = q, :211), grows from the single cell t = r (st0 = max(0,(r-w+1)>>1),
// en0 = min(r,(r+w)>>1)), computes blocks 0 .. en0/16 only (the execution mask keeps the cells above untouched) and,
// while r <= en, holds the boundary cell t = r (:212).  PAR = parity of r + w when the caller knows it (0: st0 grows
// on the step to r+1, 1: en0 does; the lane masks then move by one shift), -1 to work both out here.
template <bool RIGHT, bool HASB, int EDGE, int PAR>
__device__ __forceinline__ void narrow_steady_step(NarrowState &F, const NarrowEnv &E, SteadyCtl &C, const int ex = 0, const int ev = 0)
{
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	const int st = F.st, r = C.r;
	if (PAR < 0) { C.geLoM = ~0ull << (C.st0 - st); C.spM = HASB ? 0ull : 1ull << (C.en0 - st); }
	int xpA = dppz_shr1(F.XA), vpA = dppz_shr1(F.VA);    // neighbours of r-1
	if (EDGE == 1) { xpA = set_lane0(ex, 0, xpA); vpA = set_lane0(ev, 0, vpA); }
	if (EDGE == 2) vpA = set_lane0(E.q24, 0, vpA);
	const int en = C.en0 | 15;                           // EDGE == 2: last computed cell; t = r is computed while r <= en
	const int HpA = dppz_shr1(F.HA);
	const unsigned long long geLoM = C.geLoM;
	unsigned long long mIB = 0;
	// ---- slot B (block 4) ------------------------------------------------------------
	if (HASB) {
		const int exB = __builtin_amdgcn_readlane(F.XA, 63), evB = __builtin_amdgcn_readlane(F.VA, 63);
		const int HeB = __builtin_amdgcn_readlane(F.HA, 63);
		int xpB = dppz_shr1(F.XB), vpB = dppz_shr1(F.VB), HpB = dppz_shr1(F.HB);
		xpB = set_lane0(exB, 0, xpB); vpB = set_lane0(evB, 0, vpB);
		HpB = set_lane0(HeB, 0, HpB);
		const int zf = narrow_z(F.T0B, F.T1B, F.qptr[F.qoffB]);            // qs[qlen-1-r+st+64+lane]
		const int zB = lane_in(~geLoM) ? zf : E.ZW24;                      // refreshed up to lane loA + 63: lanes 0..loA-1 here
		if (lane < 16) {
			int ut = F.UB, yt = F.YB;
			if (EDGE == 2 && 64 + lane == r) { yt = 0; ut = E.q24; }       // :212
			int xn, vn, un, yn;
			narrow_cell<RIGHT>(zB, xpB, vpB, ut, yt, E.M24, E.q24, xn, vn, un, yn, F.accB);   // :283
			F.XB = xn; F.VB = vn; F.UB = un; F.YB = yn;
			const bool sp = 64 + lane == C.en0 - st;
			F.HB = (sp ? HpB : F.HB) + (int)((unsigned)(sp ? un : vn) >> 24);   // :318, :323-329 in H' form
		}
		mIB = ballot(F.HB > C.thrI) & 0xffffull;
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	F.qptr -= 1;
	if (EDGE != 2 || lane_in(~0ull >> (63 - (en < 63 ? en : 63)))) {
		const bool geLo = lane_in(geLoM);
		const int znew = narrow_z(F.T0A, F.T1A, F.qptr[1]);          // qs[qlen-1-r+st+lane]
		F.ZA = geLo ? znew : F.ZA;                                      // :214-228: refreshed from st0 to st0 + 63 (or to en)
		if (EDGE == 2 && r <= en && r < 64) {                           // :212
			const bool tr = lane_in(1ull << r);
			F.YA = tr ? 0 : F.YA; F.UA = tr ? E.q24 : F.UA;
		}
		int xn, vn, un, yn;
		narrow_cell<RIGHT>(F.ZA, xpA, vpA, F.UA, F.YA, E.M24, E.q24, xn, vn, un, yn, F.accA);   // :283
		F.XA = xn; F.VA = vn; F.UA = un; F.YA = yn;
		int h;
		if (!HASB) {
			const bool sp = lane_in(C.spM);
			h = (sp ? HpA : F.HA) + (int)((unsigned)(sp ? un : vn) >> 24);   // :318, :323-329 in H' form
		} else h = F.HA + (int)((unsigned)vn >> 24);                     // en0 is in block 4
		F.HA = geLo ? h : F.HA;
	}
	if ((r & 7) == 7) {                                  // narrow_flush with both accumulators in shifted form
		unsigned *row = E.p + (size_t)((r >> 3) + (st >> 4)) * 80;
		row[lane] = F.accA;
		if (lane < 16) row[64 + lane] = F.accB;
	}
	// ---- exact max (:312-349) and ksw_apply_zdrop (:88-104) ----------------------------
	// a lane that left the band holds an H' that was <= thrI when it left, and thrI only grows
	const unsigned long long mIA = ballot(F.HA > C.thrI);
	if (mIA | mIB) {
		int max_H, max_t;
		if (popc64(mIA) + popc64(mIB) == 1) {
			const int i = mIA ? ctz64(mIA) : ctz64(mIB);
			max_H = mIA ? __builtin_amdgcn_readlane(F.HA, i) : __builtin_amdgcn_readlane(F.HB, i);
			max_t = st + i + (mIA ? 0 : 64);
		} else {
			const int hiT = C.en0 - st;
			const unsigned long long mInB = HASB ? lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15) : 0ull;
			const int hAm = lane_in(geLoM) ? F.HA : INTMIN, hBm = (HASB && lane_in(mInB)) ? F.HB : INTMIN;
			max_H = wave_max_i32_keep(hAm);
			if (HASB) { const int mb = wave_max_i32_keep(hBm); max_H = mb > max_H ? mb : max_H; }
			max_t = narrow_max_t(hAm, hBm, max_H, HASB, mInB, C.st0 - st, st, C.st0, C.en0);
		}
		C.thrI = max_H; F.ez_max_t = max_t; F.ez_max_q = r - max_t;
	} else {
		const int thrZ = C.thrI - C.zd;
		if (!((ballot(F.HA >= thrZ) & geLoM) | (HASB ? ballot(F.HB >= thrZ) & 0xffffull : 0ull))) {
			// ez.max - max_H > zdrop: the full test of :98-101
			const int hiT = C.en0 - st;
			const unsigned long long mInB = HASB ? lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15) : 0ull;
			const int hAm = lane_in(geLoM) ? F.HA : INTMIN, hBm = (HASB && lane_in(mInB)) ? F.HB : INTMIN;
			int max_H = wave_max_i32_keep(hAm);
			if (HASB) { const int mb = wave_max_i32_keep(hBm); max_H = mb > max_H ? mb : max_H; }
			const int t = narrow_max_t(hAm, hBm, max_H, HASB, mInB, C.st0 - st, st, C.st0, C.en0), dq = r - t;
			if (t >= F.ez_max_t && dq >= F.ez_max_q) {
				const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (C.thrI - max_H > C.zd + l * E.e) {
					// z-drop: the loops wind down at their next check; a diagonal computed until then only adds cells to the
					// traceback slots (no cell can beat thrI any more, none can be zdrop below it)
					C.stop_r = r; C.lim = INTMIN; F.ez_max = C.thrI - r * E.qe;
					C.thrI = 0x3fffffff; C.zd = 0x7ffffff0;
				}
			}
		}
	}
	// ---- the step to r + 1: (r+w)>>1 grows from an odd r+w, (r-w+1)>>1 otherwise ----
	if (EDGE == 2) {
		const int s1 = (r + 2 - E.w) >> 1, e1 = (r + 1 + E.w) >> 1;
		C.st0 = s1 > 0 ? s1 : 0; C.en0 = e1 < r + 1 ? e1 : r + 1;
	} else if (PAR == 0) { C.st0 += 1; C.geLoM <<= 1; }
	else if (PAR == 1) { C.en0 += 1; C.spM <<= 1; }
	else { const int up = (r + E.w) & 1; C.en0 += up; C.st0 += 1 - up; }
	C.thrI += E.qe; C.r = r + 1;
}

// The diagonals C.r .. bound-1, all with or all without block 4: single steps until r + w is even, then pairs.
template <bool RIGHT, bool HASB>
__device__ __forceinline__ void narrow_steady_run(NarrowState &F, const NarrowEnv &E, SteadyCtl &C, const int bound)
{
	C.lim = C.stop_r < 0 ? bound : -0x7fffffff - 1;
	if (C.r < C.lim && ((C.r + E.w) & 1)) narrow_steady_step<RIGHT, HASB, 0, -1>(F, E, C);
	C.geLoM = ~0ull << (C.st0 - F.st); C.spM = HASB ? 0ull : 1ull << ((C.en0 - F.st) & 63);
	while (C.r + 1 < C.lim) {
		narrow_steady_step<RIGHT, HASB, 0, 0>(F, E, C);
		narrow_steady_step<RIGHT, HASB, 0, 1>(F, E, C);
	}
	if (C.r < C.lim) narrow_steady_step<RIGHT, HASB, 0, -1>(F, E, C);
}

// The steady diagonals r .. r_hi-1: w in [49, 62], the band limited by w on both sides (st0 = (r-w+1)>>1,
// en0 = (r+w)>>1, en < r), so en0-st0 is in [48, 62]: the refreshed scores end exactly 63 cells after st0, blocks 0..3
// are always computed, en0 is never on lane 0 and none of the sequence-end cases (:212, :349, :351-357) can apply.
// The loop is laid out by what the band does: the origin moves every 32 diagonals (r = 2(st+16) + w - 1), block 4
// is computed on the last few diagonals before a move (from r = 2(st+64) - w on), so a period is one move, a run of
// diagonals without block 4 and a run with it; within a run, diagonals go in pairs (st0 grows after the first, en0 after
// the second) whose lane masks move by one scalar shift -- straight-line bodies with almost nothing to decide.
// Inside the loop H is kept as H' = H + r (q+e): the per-diagonal "- (q+e)" of :318/:323 disappears and the
// thresholds move instead (thrI = ez.max + r (q+e), one scalar add per diagonal).  Lanes above the band accumulate
// harmless values on top of KSW_NEG_INF (a cell entering the band takes H of t-1, :318); lanes that left the band keep
// their last H' and are converted back exactly on exit (a cell t leaves after diagonal 2t + w).  The slot-B nibbles
// are shifted in like slot A's (its diagonals are consecutive up to the move that closes the slot).
// Returns true when the sweep must stop (z-drop); r is then the last diagonal whose cells were computed.
template <bool RIGHT>
__device__ __forceinline__ bool narrow_steady_loop(NarrowState &F, const NarrowEnv &E, int &r, const int r_hi)
{
	const int lane = lane_id();
	const int w = E.w, qe = E.qe;
	const int r0 = r;
	SteadyCtl C;
	C.r = r; C.st0 = (r - w + 1) >> 1; C.en0 = (r + w) >> 1;
	C.thrI = F.ez_max + r * qe; C.stop_r = -1; C.zd = E.zdrop < 0 ? 0x3fffffff : E.zdrop;
	C.geLoM = C.spM = 0;
	F.HA += (r - 1) * qe; F.HB += (r - 1) * qe;
	if (r & 7) F.accB >>= 4 * (8 - (r & 7));            // placed -> shifted form (nibble of r-1 lowest)
	if (r < w + 31) {                                    // the caller starts this early only when the job has room (EDGE = 2)
		C.st0 = C.st0 > 0 ? C.st0 : 0; C.en0 = C.en0 < r ? C.en0 : r;
		C.lim = r_hi < w + 31 ? r_hi : w + 31;
		const long long te0 = E.pacc ? (long long)clock64() : 0;
		while (C.r < C.lim) {
			if (C.en0 < 64) narrow_steady_step<RIGHT, false, 2, -1>(F, E, C);
			else narrow_steady_step<RIGHT, true, 2, -1>(F, E, C);
		}
		if (E.pacc && lane == 0) E.pacc[4] += (long long)clock64() - te0;
	}
	while (C.r < r_hi && C.stop_r < 0) {
		C.lim = r_hi;
		const bool moved = (C.st0 & ~15) != F.st;
		int ex = 0, ev = 0;
		if (moved) {
			// the band origin moved one block right: close the traceback slot, rotate the registers 16 lanes, re-seed slot B
			if (C.r & 7) {
				unsigned *row = E.p + (size_t)(((C.r - 1) >> 3) + (F.st >> 4)) * 80;
				const int sh = 4 * (8 - (C.r & 7));
				row[lane] = F.accA << sh;
				if (lane < 16) row[64 + lane] = F.accB << sh;
			}
			ex = __builtin_amdgcn_readlane(F.XA, 15); ev = __builtin_amdgcn_readlane(F.VA, 15);
			F.edge_h = __builtin_amdgcn_readlane(F.HA, 15) - (C.r - 1) * qe;   // t = st+15 was st0 of diagonal r-1: in the band there
			const int zf = narrow_z(F.T0B, F.T1B, F.qptr[F.qoffB + 1]);    // qs[qlen-r+st+64+lane]: scores of diagonal r-1
			const int zB = lane_in(0x7fffull) ? zf : E.ZW24;               // st0 was 16k+15, so last_sc = 78
			F.XA = (int)rot16((unsigned)F.XA, (unsigned)F.XB, lane); F.VA = (int)rot16((unsigned)F.VA, (unsigned)F.VB, lane);
			F.UA = (int)rot16((unsigned)F.UA, (unsigned)F.UB, lane); F.YA = (int)rot16((unsigned)F.YA, (unsigned)F.YB, lane);
			F.ZA = (int)rot16((unsigned)F.ZA, (unsigned)zB, lane);
			F.HA = (int)rot16((unsigned)F.HA, (unsigned)F.HB, lane);
			F.st = C.st0 & ~15;
			F.qptr += 16;
			F.XB = F.VB = F.UB = F.YB = 0; F.HB = KSW_NEG_INF;
			const uint2 ta = E.tbl[E.tg[F.st + lane]], tb = E.tbl[E.tg[F.st + 64 + (lane & 15)]];
			F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
		}
		int r_end = 2 * (F.st + 16) + w - 1;             // the next move
		r_end = r_end < r_hi ? r_end : r_hi;
		int r_b = 2 * (F.st + 64) - w;                   // block 4 from here on
		r_b = r_b > C.r ? r_b : C.r; r_b = r_b < r_end ? r_b : r_end;
		if (moved) {                                     // the diagonal of the move: lane 0 takes the block edge
			if (C.r < r_b) narrow_steady_step<RIGHT, false, 1, -1>(F, E, C, ex, ev);
			else narrow_steady_step<RIGHT, true, 1, -1>(F, E, C, ex, ev);
		}
		narrow_steady_run<RIGHT, false>(F, E, C, r_b);
		narrow_steady_run<RIGHT, true>(F, E, C, r_end);
	}
	const bool stop = C.stop_r >= 0;
	r = stop ? C.r - 1 : C.r;                            // stopped: the last diagonal computed (the z-drop's, or the one after it)
	if (!stop) {
		F.ez_max = C.thrI - r * qe;
		// back to H: a cell t was last in the band on diagonal min(r-1, 2t + w) (or never: any value will do)
		const int rl = r - 1;
		int tA = 2 * (F.st + lane) + w; tA = tA < rl ? tA : rl; tA = tA > r0 - 1 ? tA : r0 - 1;
		F.HA -= tA * qe; F.HB -= rl * qe;
		F.last_sc = ((r - w) >> 1) - F.st + 63;          // of diagonal r-1 (st0 - st + 63)
	}
	{
		const int rl = stop ? r : r - 1;                 // the last diagonal computed
		F.accB = (rl & 7) == 7 ? 0u : F.accB << (4 * (7 - (rl & 7)));   // shifted -> placed form
	}
	return stop;
}

// lanes lo .. hi of the wave, 0 <= lo <= hi <= 63 (one s_bfm_b64 unless the range is the whole wave)
__device__ __forceinline__ unsigned long long lane_span(int lo, int hi)
{
	unsigned long long m;
	const int n = hi - lo + 1;
	asm("s_bfm_b64 %0, %1, %2" : "=s"(m) : "s"(n), "s"(lo));
	return n >= 64 ? ~0ull : m;
}

struct TailCtl {
	int r, st0, en0;                                     // the coming diagonal and its true band
	int lim;                                             // end of the run (the next move, or the last diagonal + 1)
	int stopped;                                         // 1: z-drop on diagonal r; 2: the band left the matrix before r
};

// One diagonal behind the steady ones, once the band is narrower than 48 cells (see narrow_tail_loop): the band is cut
// by the end of the query and / or the target (st0 = max(r-qlen+1, (r-w+1)>>1), en0 = min(tlen-1, (r+w)>>1)), shrinks,
// never reaches block 4 (hiT <= 62) and refreshes no score past lane 62, so slot B only waits for the next move; the
// computed lanes (blocks 0 .. (en0|15)-st) run under the execution mask.  Plain H here: a one-cell band reads the H a
// cell kept when it left the band (:318).  The end-of-sequence results (:351-357) are taken on every diagonal.
template <bool RIGHT, bool EDGE>
__device__ __forceinline__ void narrow_tail_step(NarrowState &F, const NarrowEnv &E, TailCtl &C, const int zd, const int ex = 0, const int ev = 0)
{
	const int INTMIN = -0x7fffffff - 1;
	const int st = F.st, r = C.r, st0 = C.st0, en0 = C.en0;
	const int loA = st0 - st, hiT = en0 - st;
	const int sc = loA + (((en0 - st0) >> 4) + 1) * 16 - 1;      // last refreshed score lane (:215): <= 62
	const int nTop = (en0 | 15) - st;                    // last computed lane: 15, 31, 47 or 63
	int xpA = dppz_shr1(F.XA), vpA = dppz_shr1(F.VA), HpA = dppz_shr1(F.HA);   // neighbours of r-1
	if (EDGE) { xpA = set_lane0(ex, 0, xpA); vpA = set_lane0(ev, 0, vpA); }
	if (hiT == 0) HpA = set_lane0(F.edge_h, 0, HpA);     // en0 on lane 0: H[en0-1] is the block edge
	const unsigned long long inTM = lane_span(loA, hiT);
	F.qptr -= 1;
	{
		const int znew = narrow_z(F.T0A, F.T1A, F.qptr[1]);          // qs[qlen-1-r+st+lane]
		F.ZA = lane_in(lane_span(loA, sc)) ? znew : F.ZA;               // :214-228 (the 16-byte stores run past en)
	}
	if (lane_in(lane_span(0, nTop))) {
		int xn, vn, un, yn;
		narrow_cell<RIGHT>(F.ZA, xpA, vpA, F.UA, F.YA, E.M24, E.q24, xn, vn, un, yn, F.accA);   // :283
		F.XA = xn; F.VA = vn; F.UA = un; F.YA = yn;
		const bool sp = lane_in(1ull << hiT);
		const int h = (sp ? HpA : F.HA) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329
		F.HA = lane_in(inTM) ? h : F.HA;
	}
	if ((r & 7) == 7) narrow_flush(F, E, r, st);
	// ---- ez updates (:351-357) -----------------------------------------------------------
	{
		const int Hen0 = __builtin_amdgcn_readlane(F.HA, hiT), Hst0 = __builtin_amdgcn_readlane(F.HA, loA);
		const bool c1 = en0 == E.tlen - 1, c2 = r - st0 == E.qlen - 1;
		if (c1 && Hen0 > F.mte) { F.mte = Hen0; F.mte_q = r - (en0 | 15); }               // rounded en (:352)
		if (c2 && Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st0; }
		if (c1 && r == E.qlen + E.tlen - 2) F.score = Hen0;                                 // :356-357
	}
	// ---- exact max (:312-349) and ksw_apply_zdrop (:88-104) ----------------------------
	const unsigned long long mI = ballot(F.HA > F.ez_max) & inTM;
	if (mI) {
		int max_H, max_t;
		if (!(mI & (mI - 1))) {
			const int i = ctz64(mI);
			max_H = __builtin_amdgcn_readlane(F.HA, i); max_t = st + i;
		} else {
			const int hAm = lane_in(inTM) ? F.HA : INTMIN;
			max_H = wave_max_i32_keep(hAm);
			max_t = narrow_max_t(hAm, INTMIN, max_H, false, 0ull, loA, st, st0, en0);
		}
		F.ez_max = max_H; F.ez_max_t = max_t; F.ez_max_q = r - max_t;
	} else if (!(ballot(F.HA >= F.ez_max - zd) & inTM)) {
		// ez.max - max_H > zdrop: the full test of :98-101
		const int hAm = lane_in(inTM) ? F.HA : INTMIN;
		const int max_H = wave_max_i32_keep(hAm);
		const int t = narrow_max_t(hAm, INTMIN, max_H, false, 0ull, loA, st, st0, en0), dq = r - t;
		if (t >= F.ez_max_t && dq >= F.ez_max_q) {
			const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
			const int l = tl > ql ? tl - ql : ql - tl;
			if (F.ez_max - max_H > zd + l * E.e) { C.stopped = 1; C.lim = INTMIN; return; }
		}
	}
	// ---- the band of r + 1 (:196-205) ----
	{
		const int a = r + 2 - E.qlen, b = (r + 2 - E.w) >> 1, c = (r + 1 + E.w) >> 1;
		C.st0 = a > b ? a : b; C.en0 = c < E.tlen - 1 ? c : E.tlen - 1; C.r = r + 1;
		if (C.st0 > C.en0 && r + 1 < E.qlen + E.tlen - 1) { C.stopped = 2; C.lim = INTMIN; }   // :200-203
	}
}

__device__ __forceinline__ unsigned long long bit_clear(unsigned long long m, int i) { asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i)); return m; }
__device__ __forceinline__ unsigned long long bit_set(unsigned long long m, int i) { asm("s_bitset1_b64 %0, %1" : "+s"(m) : "s"(i)); return m; }

// A run of narrow_tail_step diagonals C.r .. lim-1 in the usual shape of the tail: the band is cut by the end of the
// query only (st0 = r - qlen + 1 grows on every diagonal, en0 = (r+w)>>1 < tlen-1 on every other one), and within the
// run the band origin, the number of computed blocks and the number of refreshed 16-byte score groups do not change.
// Then every lane set moves by a bit or two per diagonal, H[st0] is the only end-of-sequence result (:353-354), and
// nothing else has to be worked out.  The caller has done the diagonal of a move (lane-0 edge) with narrow_tail_step.
template <bool RIGHT>
__device__ __forceinline__ void narrow_tail_qrun(NarrowState &F, const NarrowEnv &E, TailCtl &C, const int zd)
{
	const int INTMIN = -0x7fffffff - 1;
	const int st = F.st;
	int r = C.r, loA = C.st0 - st, hiT = C.en0 - st;
	int sc = loA + (((hiT - loA) >> 4) + 1) * 16 - 1;    // last refreshed score lane (:215): <= 62
	const int nTop = (C.en0 | 15) - st;                  // last computed lane
	unsigned long long inTM = lane_span(loA, hiT), refM = lane_span(loA, sc), spM = 1ull << hiT;
	const unsigned long long actM = ~0ull >> (63 - nTop);
	int lim = C.lim, stopped = 0;
	do {
		const int xpA = dppz_shr1(F.XA), vpA = dppz_shr1(F.VA), HpA = dppz_shr1(F.HA);   // neighbours of r-1 (no move: edge 0, :210)
		F.qptr -= 1;
		{
			const int znew = narrow_z(F.T0A, F.T1A, F.qptr[1]);      // qs[qlen-1-r+st+lane]
			F.ZA = lane_in(refM) ? znew : F.ZA;                         // :214-228
		}
		if (lane_in(actM)) {
			int xn, vn, un, yn;
			narrow_cell<RIGHT>(F.ZA, xpA, vpA, F.UA, F.YA, E.M24, E.q24, xn, vn, un, yn, F.accA);   // :283
			F.XA = xn; F.VA = vn; F.UA = un; F.YA = yn;
			const bool sp = lane_in(spM);
			const int h = (sp ? HpA : F.HA) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329
			F.HA = lane_in(inTM) ? h : F.HA;
		}
		if ((r & 7) == 7) narrow_flush(F, E, r, st);
		{                                                            // :353-354 (r - st0 == qlen - 1 on every diagonal here)
			const int Hst0 = __builtin_amdgcn_readlane(F.HA, loA);
			if (Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st + loA; }
		}
		// ---- exact max (:312-349) and ksw_apply_zdrop (:88-104) ----------------------------
		const unsigned long long mI = ballot(F.HA > F.ez_max) & inTM;
		if (mI) {
			int max_H, max_t;
			if (!(mI & (mI - 1))) {
				const int i = ctz64(mI);
				max_H = __builtin_amdgcn_readlane(F.HA, i); max_t = st + i;
			} else {
				const int hAm = lane_in(inTM) ? F.HA : INTMIN;
				max_H = wave_max_i32_keep(hAm);
				max_t = narrow_max_t(hAm, INTMIN, max_H, false, 0ull, loA, st, st + loA, st + hiT);
			}
			F.ez_max = max_H; F.ez_max_t = max_t; F.ez_max_q = r - max_t;
		} else if (!(ballot(F.HA >= F.ez_max - zd) & inTM)) {
			// ez.max - max_H > zdrop: the full test of :98-101
			const int hAm = lane_in(inTM) ? F.HA : INTMIN;
			const int max_H = wave_max_i32_keep(hAm);
			const int t = narrow_max_t(hAm, INTMIN, max_H, false, 0ull, loA, st, st + loA, st + hiT), dq = r - t;
			if (t >= F.ez_max_t && dq >= F.ez_max_q) {
				const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (F.ez_max - max_H > zd + l * E.e) { stopped = 1; lim = INTMIN; }
			}
		}
		// ---- the band of r + 1: st0 + 1, en0 + 1 from an odd r + w ----
		const int p = (r + E.w) & 1;
		inTM = bit_clear(inTM, loA); refM = bit_clear(refM, loA);
		loA += 1; sc += 1; refM = bit_set(refM, sc);
		hiT += p; spM <<= p; inTM |= spM;
		r += 1;
	} while (r < lim);
	C.r = stopped ? r - 1 : r; C.st0 = st + loA; C.en0 = st + hiT;
	if (stopped) { C.stopped = 1; C.lim = INTMIN; }
}

// The diagonals behind the steady ones, r .. total-1.  While the band is still 48 or more cells wide (a handful of
// diagonals) the general narrow_diag<ND_ANY> does them; from then on narrow_tail_step, run by run between two moves of
// the band origin.  Returns true when the sweep stops early (z-drop, or the band leaves the matrix: F.band_exit);
// r is then the diagonal it stopped on.
template <bool RIGHT>
__device__ __forceinline__ bool narrow_tail_loop(NarrowState &F, const NarrowEnv &E, int &r, const int total)
{
	const int lane = lane_id();
	for (; r < total; ++r) {
		int st0, en0, nst, en;
		// narrow_tail_step needs a band that is narrower than 48 cells AND stays so: a band of w <= 47, or one cut by the end of
		// the query (st0 = r - qlen + 1 moves every diagonal from now on) or of the target (en0 = tlen - 1 stays).  An uncut band
		// of w = 48 is 48 and 47 cells wide in turn: its wider diagonals refresh scores in slot B.
		if (ksw_band(r, E.qlen, E.tlen, E.w, st0, en0, nst, en) && en0 - st0 < 48 && r > E.w + 32 && nst > 0 &&
		    (E.w <= 47 || r - E.qlen + 1 >= ((r - E.w + 1) >> 1) || ((r + E.w) >> 1) >= E.tlen - 1)) break;
		if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) return true;
	}
	if (r >= total) return false;
	TailCtl C;
	C.r = r; C.stopped = 0;
	{
		const int a = r + 1 - E.qlen, b = (r + 1 - E.w) >> 1, c = (r + E.w) >> 1;
		C.st0 = a > b ? a : b; C.en0 = c < E.tlen - 1 ? c : E.tlen - 1;
	}
	const int zd = E.zdrop < 0 ? 0x3fffffff : E.zdrop;
	while (C.r < total && !C.stopped) {
		if (C.st0 > C.en0) { C.stopped = 2; break; }     // :200-203 (behind a narrow_tail_qrun)
		const bool moved = (C.st0 & ~15) != F.st;
		int ex = 0, ev = 0;
		if (moved) {
			// the band origin moved one block right: close the traceback slot, rotate the registers 16 lanes, re-seed slot B
			if (C.r & 7) narrow_flush(F, E, C.r - 1, F.st);
			ex = __builtin_amdgcn_readlane(F.XA, 15); ev = __builtin_amdgcn_readlane(F.VA, 15);
			F.edge_h = __builtin_amdgcn_readlane(F.HA, 15);
			const int zB = F.rlB < 0 ? E.ZW24 : narrow_z(F.T0B, F.T1B, E.qs[E.qlen - 1 - F.rlB + F.st + 64 + (lane & 15)]);
			F.XA = (int)rot16((unsigned)F.XA, (unsigned)F.XB, lane); F.VA = (int)rot16((unsigned)F.VA, (unsigned)F.VB, lane);
			F.UA = (int)rot16((unsigned)F.UA, (unsigned)F.UB, lane); F.YA = (int)rot16((unsigned)F.YA, (unsigned)F.YB, lane);
			F.ZA = (int)rot16((unsigned)F.ZA, (unsigned)zB, lane);
			F.HA = (int)rot16((unsigned)F.HA, (unsigned)F.HB, lane);
			F.st = C.st0 & ~15;
			F.qptr += 16;
			F.XB = F.VB = F.UB = F.YB = 0; F.HB = KSW_NEG_INF;
			F.rlB = -1;
			const uint2 ta = E.tbl[E.tg[F.st + lane]], tb = E.tbl[E.tg[F.st + 64 + (lane & 15)]];
			F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
		}
		// the next move: st0 = max(r-qlen+1, (r-w+1)>>1) reaches st + 16
		const int m1 = F.st + 15 + E.qlen, m2 = 2 * (F.st + 16) + E.w - 1;
		int r_end = m1 < m2 ? m1 : m2;
		C.lim = r_end < total ? r_end : total;
		if (moved) narrow_tail_step<RIGHT, true>(F, E, C, zd, ex, ev);
		while (C.r < C.lim) {
			// the usual shape (cut by the end of the query only): a run up to the next diagonal on which the band origin,
			// the computed blocks (en0 crosses a multiple of 16), the refreshed score groups ((en0-st0)>>4 drops), the cut
			// (en0 reaches tlen-1) or the band itself (st0 > en0) changes
			const int rr = C.r, wd = C.en0 - C.st0, p = (rr + E.w) & 1;
			if (wd < 0) { C.stopped = 2; C.lim = -0x7fffffff - 1; break; }   // :200-203 (behind a narrow_tail_qrun)
			const bool qcut = C.st0 == rr + 1 - E.qlen && C.st0 >= (rr + 1 - E.w) >> 1 && ((rr + E.w) >> 1) < E.tlen - 1 && wd >= 0;
			if (qcut) {
				const int r_k = rr + 2 * ((wd & 15) + 1) - 1 + p;            // width falls to 16k - 1
				const int r_e = rr + 2 * (16 - (C.en0 & 15)) - p;             // en0 reaches the next multiple of 16
				const int r_t = 2 * (E.tlen - 1) - E.w;                      // (r+w)>>1 reaches tlen-1
				int e = r_k < r_e ? r_k : r_e; e = e < r_t ? e : r_t; e = e < C.lim ? e : C.lim;
				if (e > rr) {
					const int keep = C.lim;
					C.lim = e;
					narrow_tail_qrun<RIGHT>(F, E, C, zd);
					if (!C.stopped) C.lim = keep;
					continue;
				}
			}
			narrow_tail_step<RIGHT, false>(F, E, C, zd);
		}
	}
	r = C.r;
	if (C.stopped == 2) F.band_exit = 1;
	return C.stopped != 0;
}

// Returns false when the job is not for this sweep (a code outside the 5-letter alphabet; nothing useful in
// `out`): the caller runs ksw_wave_fast() instead.  Precondition: ksw_narrow_ok(P).
template <bool RIGHT>
__device__ inline bool ksw_wave_narrow(const uint8_t *query, int qlen, const uint8_t *target, int tlen,
                                       const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap,
                                       KswOut &out, long long *pacc = nullptr)
{
	const long long tc0 = pacc ? (long long)clock64() : 0;
	const int lane = lane_id();
	const int w = P.w;
	const int q = P.q, e = P.e, qe = q + e, flag = P.flag;
	out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;   // :81-86
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = 0;
	if (qlen <= 0 || tlen <= 0) return true;             // :147
	if (-P.min_sc > 2 * (q + e)) return true;            // :171
	const int TP = (tlen + 15) / 16 * 16 + 96, QR = (qlen + 15) / 16 * 16 + 96;
	uint2 *tbl = (uint2 *)lds;                           // 5 entries, 64 bytes reserved
	uint8_t *tg = lds + 64;                              // target codes, zero padded (sf of :175,:188)
	unsigned *qs = (unsigned *)(tg + TP) + 16;           // selector words of the reversed query, padded with code 0 on
	                                                     // both sides (:187): every index qlen-1-r+t a lane can form is in [-16, QR)
	const unsigned ZW = (unsigned)(2 * qe) & 0xff, ZM = (unsigned)(2 * qe + P.sc_mch) & 0xff, ZX = (unsigned)(2 * qe + P.sc_mis) & 0xff;
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
	const long long tc1 = pacc ? (long long)clock64() : 0;

	NarrowState F;
	F.XA = F.VA = F.UA = F.YA = 0; F.ZA = (int)(ZW << 24);
	F.XB = F.VB = F.UB = F.YB = 0;
	{
		const uint2 ta = tbl[tg[lane]], tb = tbl[tg[64 + (lane & 15)]];
		F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
	}
	F.rlB = -1; F.HA = F.HB = KSW_NEG_INF; F.st = 0; F.qptr = qs + (qlen - 1 + lane); F.qoffB = 64 + (lane & 15) - lane;
	F.edge_h = KSW_NEG_INF; F.last_sc = -1; F.band_exit = 0; F.accA = F.accB = 0;
	F.ez_max = 0; F.ez_max_t = F.ez_max_q = -1; F.mqe = F.mte = F.score = KSW_NEG_INF; F.mqe_t = F.mte_q = -1;
	NarrowEnv E;
	E.tg = tg; E.qs = qs; E.tbl = tbl; E.p = (unsigned *)p; E.qlen = qlen; E.tlen = tlen; E.w = w; E.qe = qe; E.e = e;
	E.pacc = pacc;
	E.zdrop = P.zdrop; E.ZW24 = (int)(ZW << 24); E.M24 = ZM << 24; E.q24 = (int)(((unsigned)q & 0xff) << 24);
	const int total = qlen + tlen - 1;
	// steady diagonals: st0 = (r-w+1)>>1 > r-qlen+1, en0 = (r+w)>>1 < tlen-1, en < r
	// the first w+31 diagonals stay clear of the sequence ends (w = 0: diagonal 1 has no cell at all, :200-203 -- the general
	// narrow_diag sees that)
	const bool roomy = w >= 1 && qlen >= w + 32 && tlen >= w + 32;
	int r_hi = 2 * tlen - 3 - w < 2 * qlen - w - 3 ? 2 * tlen - 3 - w : 2 * qlen - w - 3;
	r_hi = r_hi + 1 < total ? r_hi + 1 : total;
	bool stop = narrow_diag<RIGHT, ND_FIRST>(F, E, 0);
	int r = 1;
	bool tracked = true;                                 // F.rlB is up to date
	if (!stop && roomy && w >= 49 && r < r_hi) {
		// the steady loop takes the early diagonals too
	} else if (!stop && roomy) {
		do { if (narrow_diag<RIGHT, ND_EARLY>(F, E, r)) { stop = true; break; } } while (++r < w + 31);
		tracked = false;
	} else if (!stop) {
		for (; r < total && r < w + 32; ++r) if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) { stop = true; break; }
	}
	if (w >= 49 && !stop && r < r_hi) {                  // w >= 49: a steady band spans blocks 0..3
		stop = narrow_steady_loop<RIGHT>(F, E, r, r_hi);
		tracked = false;
	}
	if (!stop) {
		if (!tracked) F.rlB = lane <= F.last_sc - 64 ? r - 1 : -1;   // what the growing diagonals did not track (see narrow_diag)
		const long long tt0 = pacc ? (long long)clock64() : 0;
		stop = narrow_tail_loop<RIGHT>(F, E, r, total);
		if (pacc && lane == 0) pacc[5] += (long long)clock64() - tt0;
	}
	{
		const int r_last = stop ? (F.band_exit ? r - 1 : r) : total - 1;     // the last diagonal whose cells were computed
		if (r_last >= 0 && (r_last & 7) != 7) narrow_flush(F, E, r_last, F.st);
	}
	WSYNC();
	out.max = F.ez_max; out.zdropped = stop ? 1 : 0; out.max_q = F.ez_max_q; out.max_t = F.ez_max_t;   // every early exit is a z-drop (:98-101, :200-203)
	out.mqe = F.mqe; out.mqe_t = F.mqe_t; out.mte = F.mte; out.mte_q = F.mte_q; out.score = stop ? KSW_NEG_INF : F.score;   // (the reference tests the z-drop before it takes the score of the last diagonal, :355-357: a sweep that stopped has none)
	const long long tc2 = pacc ? (long long)clock64() : 0;
	if (pacc && lane == 0) { pacc[0] += tc1 - tc0; pacc[1] += tc2 - tc1; pacc[3] += 1; }
	ksw_backtrack_wave<1>(p, 0, qlen, tlen, w, flag, stop ? 1 : 0, F.ez_max_t, F.ez_max_q, cig_tmp, cig_cap, out);
	if (pacc && lane == 0) pacc[2] += (long long)clock64() - tc2;
	return true;
}

}  // namespace ihp
