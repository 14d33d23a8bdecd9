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
	asm("v_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(s));
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

// One cell (:116-137 + :262-310); every value is (int8 << 24).  z > 0, so :271 is the identity.
template <bool RIGHT>
__device__ __forceinline__ void narrow_cell(int z, int xp, int vp, int u, int y, unsigned M, int q,
                                            int &xn, int &vn, int &un, int &yn, unsigned &d)
{
	const int a = xp + vp, b = y + u;
	if (!RIGHT) d = a > z ? 1u : 0u;                                    // :265
	else        d = z > a ? 0u : 1u;                                    // :291
	unsigned zz = (unsigned)z > (unsigned)a ? (unsigned)z : (unsigned)a;    // :272 _mm_max_epu8
	if (!RIGHT) { if (b > (int)zz) d = 2u; }                            // :273-274
	else        { if (!((int)zz > b)) d = 2u; }                         // :299-300
	zz = zz > (unsigned)b ? zz : (unsigned)b;                           // :131
	zz = zz < M ? zz : M;                                               // :132
	un = (int)zz - vp; vn = (int)zz - u;                                // :133-134
	const int zq = (int)zz - q;
	const int a2 = a - zq, b2 = b - zq;
	if (!RIGHT) {                                                       // :277-282
		xn = a2 > 0 ? a2 : 0; yn = b2 > 0 ? b2 : 0;
		d |= (a2 > 0 ? 0x08u : 0u) | (b2 > 0 ? 0x10u : 0u);
	} else {                                                            // :303-308
		xn = a2 < 0 ? 0 : a2; yn = b2 < 0 ? 0 : b2;
		d |= (a2 < 0 ? 0u : 0x08u) | (b2 < 0 ? 0u : 0x10u);
	}
}

struct NarrowState {
	// per lane: slot A (t = st + lane), slot B (t = st + 64 + lane, lanes 0..15)
	int XA, VA, UA, YA, ZA, HA;
	unsigned T1A, T0A;
	int XB, VB, UB, YB, HB, rlB;
	const unsigned *qptr;                                // LDS: this lane's selector word for the coming diagonal
	int qoffB;                                           // 64 + (lane & 15) - lane: from qptr to the lane's slot-B word
	unsigned T1B, T0B;
	// wave-uniform
	int st, edge_h, last_sc;
	int ez_max, ez_max_t, ez_max_q, mqe, mqe_t, mte, mte_q, score;
};

struct NarrowEnv {
	const uint8_t *tg;                                   // LDS: target codes
	const unsigned *qs;                                  // LDS: selector words of the reversed query
	const uint2 *tbl;                                    // LDS: z table per target code
	uint8_t *p;
	int qlen, tlen, w, ncol, qe, e, zdrop;
	int q24, ZW24;                                       // q << 24, z of a never-refreshed cell << 24
	unsigned M24;                                        // max_sc << 24
};

__device__ __forceinline__ int narrow_z(unsigned T0, unsigned T1, unsigned sel) { return (int)__builtin_amdgcn_perm(T0, T1, sel); }

// One anti-diagonal; same contract as fast_diag() in ksw_fast.h (returns true when the sweep must stop; every such
// exit is a z-drop for the caller, :98-101 and :200-203).  KIND selects what the diagonal can be:
//   ND_ANY    any r >= 1;
//   ND_FIRST  r == 0 (the same code with r folded);
//   ND_EARLY  1 <= r <= w+30 of a job with qlen, tlen >= w+32: the band starts in block 0 (st == 0, no move), is
//             not yet cut by the sequence ends, and grows: st0 = max(0,(r-w+1)>>1), en0 = min(r,(r+w)>>1);
//   ND_STEADY w in [49, 62] and the band limited by w on both sides with en < r (the caller supplies [st0, en0] and
//             advances it): en0-st0 is in [48, 62], so the refreshed scores end exactly 63 cells after st0, blocks
//             0..3 are always computed and en0 is never on lane 0.
// A block edge x[st-1], v[st-1] is taken (:207-208) exactly on the diagonal where st moves: then st-1 = old st+15
// lies in [last_st, last_en]; without a move st-1 < last_st and the edge is 0 (:210).
// Slot B's score bytes are not tracked per diagonal while the refreshed range only grows (st0, en0 and en0-st0
// non-decreasing, i.e. ND_EARLY and ND_STEADY): then a B lane is "refreshed on the previous diagonal" up to
// last_sc-64 and "never" above, which is all its lazily formed value needs.
enum { ND_ANY = 0, ND_STEADY = 1, ND_EARLY = 2, ND_FIRST = 3 };

template <bool RIGHT, int KIND>
__device__ __forceinline__ bool narrow_diag(NarrowState &F, const NarrowEnv &E, const int r, int st0 = 0, int en0 = 0)
{
	constexpr bool STEADY = KIND == ND_STEADY, EARLY = KIND == ND_EARLY, GROWING = STEADY || EARLY;
	constexpr bool ENDS = KIND == ND_ANY || KIND == ND_FIRST;          // the band may touch the sequence ends
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	int nst, en;
	if (STEADY) { nst = st0 & ~15; en = en0 | 15; }
	else if (EARLY) {
		st0 = (r - E.w + 1) >> 1; st0 = st0 > 0 ? st0 : 0;
		en0 = (r + E.w) >> 1; en0 = en0 < r ? en0 : r;
		nst = 0; en = en0 | 15;
	} else if (!ksw_band(r, E.qlen, E.tlen, E.w, st0, en0, nst, en)) return true;   // :200-203
	int ex = 0, ev = 0;
	if (!EARLY && nst != F.st) {
		// the band origin moved one block right: rotate the registers 16 lanes, re-seed slot B
		ex = __builtin_amdgcn_readlane(F.XA, 15);
		ev = __builtin_amdgcn_readlane(F.VA, 15);
		F.edge_h = __builtin_amdgcn_readlane(F.HA, 15);
		int zB;
		if (GROWING) {
			const int zf = narrow_z(F.T0B, F.T1B, F.qptr[F.qoffB + 1]);    // qs[qlen-r+st+64+lane]: scores of diagonal r-1
			zB = lane_in(STEADY ? 0x7fffull : lane_range(0, F.last_sc - 64)) ? zf : E.ZW24;   // steady: st0 was 16k+15, so last_sc = 78
		} else zB = F.rlB < 0 ? E.ZW24 : narrow_z(F.T0B, F.T1B, E.qs[E.qlen - 1 - F.rlB + F.st + 64 + (lane & 15)]);
		F.XA = (int)rot16((unsigned)F.XA, (unsigned)F.XB, lane); F.VA = (int)rot16((unsigned)F.VA, (unsigned)F.VB, lane);
		F.UA = (int)rot16((unsigned)F.UA, (unsigned)F.UB, lane); F.YA = (int)rot16((unsigned)F.YA, (unsigned)F.YB, lane);
		F.ZA = (int)rot16((unsigned)F.ZA, (unsigned)zB, lane);
		F.HA = (int)rot16((unsigned)F.HA, (unsigned)F.HB, lane);
		F.st = nst;
		F.qptr += 16;
		F.XB = F.VB = F.UB = F.YB = 0; F.HB = KSW_NEG_INF;
		if (!GROWING) F.rlB = -1;
		const uint2 ta = E.tbl[E.tg[nst + lane]], tb = E.tbl[E.tg[nst + 64 + (lane & 15)]];
		F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
	} else if (EARLY || (!STEADY && F.st == 0)) { ev = r ? E.q24 : 0; }     // :211
	// neighbours of r-1 (taken before anything is overwritten); lane 0 gets the block edge x1, v1 (:207-211)
	const int xpA = EARLY ? dppz_shr1(F.XA) : set_lane0(ex, 0, dppz_shr1(F.XA)), vpA = set_lane0(ev, 0, dppz_shr1(F.VA));
	int HpA = dppz_shr1(F.HA);
	if (ENDS) HpA = set_lane0(F.edge_h, 0, HpA);        // otherwise en0 is never on lane 0
	const int st = EARLY ? 0 : F.st;
	const int loA = st0 - st;                            // first true-band lane (<= 15)
	const int hiT = en0 - st;                            // last true-band lane (may be >= 64: slot B)
	const int nTop = en - st;                            // last computed lane: 15, 31, 47, 63 or 79
	const int sc = STEADY ? loA + 63 : st0 + ((en0 - st0) / 16 + 1) * 16 - 1 - st;   // last refreshed score lane (:215), > loA
	const int hasB = (nTop >> 6) & 1;                    // block 4 is computed (an integer: a bool carried across the
	                                                     // DPP asm below is materialised per lane and tested again)
	uint8_t *pr = E.p + (size_t)r * E.ncol;
	const bool has_spec = !ENDS || (r > 0 && en0 > 0);   // H[en0] comes from H[en0-1] + u (:318)
	// lane predicates of slot A: compares against the scalar limits (three VALU compares; as scalar shift/and
	// chains they would cost twice as many instructions on the busier scalar unit)
	const bool geLo = lane >= loA;
	const bool refA = STEADY ? geLo : (geLo && lane <= sc);             // refreshed score lanes (:214-228); sc > loA
	const bool inTA = geLo && lane <= hiT;                              // inside the true band (hiT >= loA)
	const bool spA = (ENDS ? has_spec : true) && lane == hiT;           // H[en0] comes from H[en0-1] + u (:318); in slot B if hiT >= 64
	int hB = INTMIN, hA;
	unsigned long long mInB = 0;
	// ---- slot B (block 4) ------------------------------------------------------------
	if (!GROWING) F.rlB = lane_in(sc >= 64 ? ~0ull >> (127 - sc) : 0ull) ? r : F.rlB;  // :214-228 runs past en; value formed on use
	if (hasB) {                                                        // nTop == 79: the whole block
		const int exB = __builtin_amdgcn_readlane(F.XA, 63), evB = __builtin_amdgcn_readlane(F.VA, 63);
		const int HeB = __builtin_amdgcn_readlane(F.HA, 63);
		int xpB = dppz_shr1(F.XB), vpB = dppz_shr1(F.VB), HpB = dppz_shr1(F.HB);
		xpB = set_lane0(exB, 0, xpB); vpB = set_lane0(evB, 0, vpB);
		HpB = set_lane0(HeB, 0, HpB);
		int zB;
		if (GROWING) {
			const int zf = narrow_z(F.T0B, F.T1B, F.qptr[F.qoffB]);        // qs[qlen-1-r+st+64+lane]
			zB = lane_in(lane_range(0, sc - 64)) ? zf : E.ZW24;
		} else zB = F.rlB < 0 ? E.ZW24 : narrow_z(F.T0B, F.T1B, E.qs[E.qlen - 1 - F.rlB + st + 64 + (lane & 15)]);
		mInB = lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15);           // loA <= 15, so block 4 is never below the band
		if (lane < 16) {
			int ut = F.UB, yt = F.YB;
			if (!STEADY && st + 64 + lane == r) { yt = 0; ut = r ? E.q24 : 0; }   // :212
			int xn, vn, un, yn; unsigned d;
			narrow_cell<RIGHT>(zB, xpB, vpB, ut, yt, E.M24, E.q24, xn, vn, un, yn, d);
			F.XB = xn; F.VB = vn; F.UB = un; F.YB = yn;
			pr[64 + lane] = (uint8_t)d;                                // :283
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
		if (!STEADY && r <= en && r - st < 64) {                       // :212 (only while the band still touches t == r)
			const bool tr = lane_in(1ull << (r - st));
			F.YA = tr ? 0 : F.YA; F.UA = tr ? (r ? E.q24 : 0) : F.UA;
		}
		int xn, vn, un, yn; unsigned d;
		narrow_cell<RIGHT>(F.ZA, xpA, vpA, F.UA, F.YA, E.M24, E.q24, xn, vn, un, yn, d);
		int h;
		const bool sp = spA;
		if (KIND != ND_FIRST) h = (sp ? HpA : F.HA) + (int)((unsigned)(sp ? un : vn) >> 24) - E.qe;   // :318, :323-329
		else h = (int)((unsigned)vn >> 24) - E.qe - E.qe;              // :349
		if (STEADY) {                                                  // a steady band always covers blocks 0..3
			F.XA = xn; F.VA = vn; F.UA = un; F.YA = yn;
			pr[lane] = (uint8_t)d;
		} else {
			const bool act = lane_in(~0ull >> (63 - (nTop < 63 ? nTop : 63)));
			F.XA = act ? xn : F.XA; F.VA = act ? vn : F.VA; F.UA = act ? un : F.UA; F.YA = act ? yn : F.YA;
			if (act) pr[lane] = (uint8_t)d;
		}
		hA = inTA ? h : INTMIN;
		F.HA = inTA ? h : F.HA;
	}
	if (!STEADY) F.last_sc = sc;
	// ---- exact max (:320-348) ----------------------------------------------------------
	int max_H = wave_max_i32_keep(hA);
	if (hasB) { const int mb = wave_max_i32_keep(hB); max_H = mb > max_H ? mb : max_H; }
	// ---- ez updates (:351-357) -----------------------------------------------------------
	if (ENDS) {
		int Hen0 = 0;
		if (en0 == E.tlen - 1 || r - st0 == E.qlen - 1) {
			Hen0 = hiT < 64 ? __builtin_amdgcn_readlane(hA, hiT & 63) : __builtin_amdgcn_readlane(hB, (hiT - 64) & 63);
			const int Hst0 = __builtin_amdgcn_readlane(hA, loA);
			if (en0 == E.tlen - 1 && Hen0 > F.mte) { F.mte = Hen0; F.mte_q = r - en; }        // rounded en (:352)
			if (r - st0 == E.qlen - 1 && Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st0; }
		}
		if (r == E.qlen + E.tlen - 2 && en0 == E.tlen - 1) F.score = Hen0;                  // :356-357
	}
	// ksw_apply_zdrop (:88-104) only looks at max_t when the maximum improves or has fallen more than zdrop below
	// the best one (the test of :98 cannot hold otherwise): everything about max_t, ties included, is skipped on
	// the other diagonals.
	const bool improves = max_H > F.ez_max;
	if (!improves && (E.zdrop < 0 || F.ez_max - max_H <= E.zdrop)) return false;
	int max_t;
	{
		const unsigned long long mA = ballot(hA == max_H);                 // lanes outside the band hold INT_MIN
		const unsigned long long mB = hasB ? ballot(hB == max_H) & mInB : 0ull;
		if (popc64(mA) + popc64(mB) == 1) {
			max_t = mA ? st + ctz64(mA) : st + 64 + ctz64(mB);
		} else {
			// ties: en0 first, then stride classes of the vector part, then the scalar tail
			const unsigned long long m = loA ? ((mA >> loA) | (mB << (64 - loA))) : mA;   // bit i <-> t = st0 + i
			const int ie = en0 - st0, nv = (en0 - st0) / 4 * 4;
			if ((m >> ie) & 1) max_t = en0;
			else {
				const unsigned long long mv = nv ? (m & ((1ull << nv) - 1)) : 0ull;
				max_t = en0;
				if (mv) {
					for (int j = 0; j < 4; ++j) {
						const unsigned long long cm = mv & (0x1111111111111111ull << j);
						if (cm) { max_t = st0 + ctz64(cm); break; }
					}
				} else {
					const unsigned long long mt = m & ~mv;
					if (mt) max_t = st0 + ctz64(mt);
				}
			}
		}
	}
	const int t = max_t, dq = r - max_t;
	if (improves) { F.ez_max = max_H; F.ez_max_t = t; F.ez_max_q = dq; return false; }
	if (t < F.ez_max_t || dq < F.ez_max_q) return false;
	const int tl = t - F.ez_max_t, ql = dq - F.ez_max_q;
	const int l = tl > ql ? tl - ql : ql - tl;
	return F.ez_max - max_H > E.zdrop + l * E.e;
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
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
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
	F.edge_h = KSW_NEG_INF; F.last_sc = -1;
	F.ez_max = 0; F.ez_max_t = F.ez_max_q = -1; F.mqe = F.mte = F.score = KSW_NEG_INF; F.mqe_t = F.mte_q = -1;
	NarrowEnv E;
	E.tg = tg; E.qs = qs; E.tbl = tbl; E.p = p; E.qlen = qlen; E.tlen = tlen; E.w = w; E.ncol = ncol; E.qe = qe; E.e = e;
	E.zdrop = P.zdrop; E.ZW24 = (int)(ZW << 24); E.M24 = ZM << 24; E.q24 = (int)(((unsigned)q & 0xff) << 24);
	const int total = qlen + tlen - 1;
	// steady diagonals: st0 = (r-w+1)>>1 > r-qlen+1, en0 = (r+w)>>1 < tlen-1, en < r
	const bool roomy = qlen >= w + 32 && tlen >= w + 32;   // the first w+31 diagonals stay clear of the sequence ends
	int r_hi = 2 * tlen - 3 - w < 2 * qlen - w - 3 ? 2 * tlen - 3 - w : 2 * qlen - w - 3;
	r_hi = r_hi + 1 < total ? r_hi + 1 : total;
	bool stop = narrow_diag<RIGHT, ND_FIRST>(F, E, 0);
	int r = 1;
	bool tracked = true;                                 // F.rlB is up to date
	if (!stop && roomy) {
		do { if (narrow_diag<RIGHT, ND_EARLY>(F, E, r)) { stop = true; break; } } while (++r < w + 31);
		tracked = false;
	} else if (!stop) {
		for (; r < total && r < w + 32; ++r) if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) { stop = true; break; }
	}
	if (w >= 49 && !stop && r < r_hi) {                  // w >= 49: a steady band spans blocks 0..3
		int st0 = (r - w + 1) >> 1, en0 = (r + w) >> 1;
		do {
			if (narrow_diag<RIGHT, ND_STEADY>(F, E, r, st0, en0)) { stop = true; break; }
			const int up = (r + w) & 1;                  // (r+w)>>1 grows on the step from an odd r+w, (r-w+1)>>1 otherwise
			en0 += up; st0 += 1 - up;
		} while (++r < r_hi);
		F.last_sc = ((r - w) >> 1) - F.st + 63;          // of diagonal r-1 (st0 - st + 63)
		tracked = false;
	}
	if (!stop) {
		if (!tracked) F.rlB = lane <= F.last_sc - 64 ? r - 1 : -1;   // what the growing diagonals did not track (see narrow_diag)
		for (; r < total; ++r) if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) { stop = true; break; }
	}
	WSYNC();
	out.max = F.ez_max; out.zdropped = stop ? 1 : 0; out.max_q = F.ez_max_q; out.max_t = F.ez_max_t;   // every early exit is a z-drop (:98-101, :200-203)
	out.mqe = F.mqe; out.mqe_t = F.mqe_t; out.mte = F.mte; out.mte_q = F.mte_q; out.score = F.score;
	const long long tc2 = pacc ? (long long)clock64() : 0;
	if (pacc && lane == 0) { pacc[0] += tc1 - tc0; pacc[1] += tc2 - tc1; pacc[3] += 1; }
	ksw_backtrack_wave(p, ncol, qlen, tlen, w, flag, stop ? 1 : 0, F.ez_max_t, F.ez_max_q, cig_tmp, cig_cap, out);
	if (pacc && lane == 0) pacc[2] += (long long)clock64() - tc2;
	return true;
}

// ---- the same sweep in pieces, for ksw_pair.h (two alignments side by side); ksw_wave_narrow above stays in one piece: split
// like this it needed 12 more VGPRs at 8 waves per SIMD and spilled them ----
// narrow_prepare: result reset, early outs, LDS tables, state.  Returns 0 to go on, 1 when `out` is already final
// (:147, :171), 2 when the job is not for this sweep (a code outside the 5-letter alphabet).
__device__ __forceinline__ int narrow_prepare(const uint8_t *query, int qlen, const uint8_t *target, int tlen, const KswParams &P, uint8_t *lds,
                                     uint8_t *p, KswOut &out, NarrowState &F, NarrowEnv &E)
{
	const int lane = lane_id();
	const int w = P.w;
	const int q = P.q, e = P.e, qe = q + e;
	out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;   // :81-86
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = 0;
	if (qlen <= 0 || tlen <= 0) return 1;                // :147
	if (-P.min_sc > 2 * (q + e)) return 1;               // :171
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
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
	if (ballot(bad)) return 2;
	WSYNC();
	F.XA = F.VA = F.UA = F.YA = 0; F.ZA = (int)(ZW << 24);
	F.XB = F.VB = F.UB = F.YB = 0;
	{
		const uint2 ta = tbl[tg[lane]], tb = tbl[tg[64 + (lane & 15)]];
		F.T1A = ta.x; F.T0A = ta.y; F.T1B = tb.x; F.T0B = tb.y;
	}
	F.rlB = -1; F.HA = F.HB = KSW_NEG_INF; F.st = 0; F.qptr = qs + (qlen - 1 + lane); F.qoffB = 64 + (lane & 15) - lane;
	F.edge_h = KSW_NEG_INF; F.last_sc = -1;
	F.ez_max = 0; F.ez_max_t = F.ez_max_q = -1; F.mqe = F.mte = F.score = KSW_NEG_INF; F.mqe_t = F.mte_q = -1;
	E.tg = tg; E.qs = qs; E.tbl = tbl; E.p = p; E.qlen = qlen; E.tlen = tlen; E.w = w; E.ncol = ncol; E.qe = qe; E.e = e;
	E.zdrop = P.zdrop; E.ZW24 = (int)(ZW << 24); E.M24 = ZM << 24; E.q24 = (int)(((unsigned)q & 0xff) << 24);
	return 0;
}

// Where the steady diagonals of a job end: st0 = (r-w+1)>>1 > r-qlen+1, en0 = (r+w)>>1 < tlen-1, en < r
__device__ __forceinline__ int narrow_r_hi(int qlen, int tlen, int w)
{
	int r_hi = 2 * tlen - 3 - w < 2 * qlen - w - 3 ? 2 * tlen - 3 - w : 2 * qlen - w - 3;
	const int total = qlen + tlen - 1;
	return r_hi + 1 < total ? r_hi + 1 : total;
}
__device__ __forceinline__ bool narrow_roomy(int qlen, int tlen, int w) { return qlen >= w + 32 && tlen >= w + 32; }   // the first w+31 diagonals stay clear of the sequence ends

// diagonals 0 .. w+30 (w+31 when not roomy); r is the next diagonal on return
template <bool RIGHT>
__device__ __forceinline__ void narrow_head(NarrowState &F, const NarrowEnv &E, int &r, bool &stop, bool &tracked)
{
	const int w = E.w, total = E.qlen + E.tlen - 1;
	stop = narrow_diag<RIGHT, ND_FIRST>(F, E, 0);
	r = 1;
	tracked = true;                                      // F.rlB is up to date
	if (!stop && narrow_roomy(E.qlen, E.tlen, w)) {
		do { if (narrow_diag<RIGHT, ND_EARLY>(F, E, r)) { stop = true; break; } } while (++r < w + 31);
		tracked = false;
	} else if (!stop) {
		for (; r < total && r < w + 32; ++r) if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) { stop = true; break; }
	}
}

// the steady diagonals from r on, then whatever is left
template <bool RIGHT>
__device__ __forceinline__ void narrow_rest(NarrowState &F, const NarrowEnv &E, int &r, bool &stop, bool &tracked)
{
	const int lane = lane_id();
	const int w = E.w, total = E.qlen + E.tlen - 1, r_hi = narrow_r_hi(E.qlen, E.tlen, w);
	if (w >= 49 && !stop && r < r_hi) {                  // w >= 49: a steady band spans blocks 0..3
		int st0 = (r - w + 1) >> 1, en0 = (r + w) >> 1;
		do {
			if (narrow_diag<RIGHT, ND_STEADY>(F, E, r, st0, en0)) { stop = true; break; }
			const int up = (r + w) & 1;                  // (r+w)>>1 grows on the step from an odd r+w, (r-w+1)>>1 otherwise
			en0 += up; st0 += 1 - up;
		} while (++r < r_hi);
		F.last_sc = ((r - w) >> 1) - F.st + 63;          // of diagonal r-1 (st0 - st + 63)
		tracked = false;
	}
	if (!stop) {
		if (!tracked) F.rlB = lane <= F.last_sc - 64 ? r - 1 : -1;   // what the growing diagonals did not track (see narrow_diag)
		for (; r < total; ++r) if (narrow_diag<RIGHT, ND_ANY>(F, E, r)) { stop = true; break; }
	}
}

__device__ __forceinline__ void narrow_finish(const NarrowState &F, const NarrowEnv &E, bool stop, int flag, uint8_t *p, uint32_t *cig_tmp, int cig_cap, KswOut &out)
{
	out.max = F.ez_max; out.zdropped = stop ? 1 : 0; out.max_q = F.ez_max_q; out.max_t = F.ez_max_t;   // every early exit is a z-drop (:98-101, :200-203)
	out.mqe = F.mqe; out.mqe_t = F.mqe_t; out.mte = F.mte; out.mte_q = F.mte_q; out.score = F.score;
	ksw_backtrack_wave(p, E.ncol, E.qlen, E.tlen, E.w, flag, stop ? 1 : 0, F.ez_max_t, F.ez_max_q, cig_tmp, cig_cap, out);
}

}  // namespace ihp
