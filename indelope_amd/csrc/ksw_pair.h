// ksw_pair.h -- two alignments in one wavefront for the steady part of the production sweep (ksw_narrow.h).
//
// k_ksw is instruction-issue bound (~53 VALU + ~55 SALU per anti-diagonal and alignment, at ~2 of the ~3 wave
// instructions a CU issues per cycle) and a lane's int8 cell occupies one byte of a 32-bit register.  On a STEADY
// diagonal -- the band limited by w on both sides, clear of the sequence ends: most of a contig-vs-window alignment --
// the band geometry depends on the diagonal number and w only (st0 = (r-w+1)>>1, en0 = (r+w)>>1), not on the sequence
// lengths: two alignments at the same diagonal share every scalar (band origin, block rotation, lane predicates, loop
// control) and differ in data only.  So the cells of a PAIR of alignments are carried in the two 16-bit halves of a
// register, value << 8 in each half (alignment 1 low, alignment 2 high -- the high half is the single sweep's own
// value << 24 format): v_pk_add_u16 / v_pk_sub_u16 wrap like _mm_add_epi8 / _mm_sub_epi8, v_pk_max_u16 / v_pk_min_u16
// are _mm_max_epu8 / _mm_min_epu8, a saturating v_pk_sub_i16 gives the sign of the int8 compares (_mm_cmpgt_epi8),
// and one v_cndmask with the shared predicate selects for both.  The 32-bit H values, the exact maximum and the
// traceback bytes stay per alignment.  Heads (diagonals 0 .. w+30) and tails run one alignment at a time with the
// single sweep; the state converts with one v_perm per register.  Left-aligned gaps only (flag without KSW_EZ_RIGHT).
// Reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388; results are bit-identical to ksw_wave_narrow<false>.
#pragma once
#include "ksw_narrow.h"

namespace ihp {

#define PK2(NAME, OP) __device__ __forceinline__ unsigned NAME(unsigned a, unsigned b) { unsigned r; asm(OP " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PK2(pk_add, "v_pk_add_u16")
PK2(pk_sub, "v_pk_sub_u16")
PK2(pk_max_u, "v_pk_max_u16")
PK2(pk_min_u, "v_pk_min_u16")
PK2(pk_max_i, "v_pk_max_i16")
#undef PK2
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b) { unsigned r; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_lshr15(unsigned a, unsigned c15) { unsigned r; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(c15), "v"(a)); return r; }
__device__ __forceinline__ unsigned pk_ashr15(unsigned a, unsigned c15) { unsigned r; asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(r) : "v"(c15), "v"(a)); return r; }

// (a1 << 24, a2 << 24) -> a1 << 8 | a2 << 24 and back
__device__ __forceinline__ unsigned pair_pack(int a1, int a2) { return ((unsigned)a1 >> 16) | (unsigned)a2; }
__device__ __forceinline__ int pair_lo(unsigned v) { return (int)((v & 0xff00u) << 16); }
__device__ __forceinline__ int pair_hi(unsigned v) { return (int)(v & 0xff000000u); }

struct PairConst { unsigned M, q, ZW, c15, c2, c8, c16; };     // both halves: max_sc << 8, q << 8, z of a never-refreshed cell << 8; 15, 2, 8, 16

// One cell of both alignments (:116-137 + :262-284, left-aligned); every value is (int8 << 8) per half; z > 0.
__device__ __forceinline__ void pair_cell(unsigned z, unsigned xp, unsigned vp, unsigned u, unsigned y, const PairConst &C,
                                          unsigned &xn, unsigned &vn, unsigned &un, unsigned &yn, unsigned &d)
{
	const unsigned a = pk_add(xp, vp), b = pk_add(y, u);
	d = pk_lshr15(pk_sub_sat(z, a), C.c15);                             // :265  a > z
	unsigned zz = pk_max_u(z, a);                                       // :272  _mm_max_epu8
	const unsigned m2 = pk_ashr15(pk_sub_sat(zz, b), C.c15);            // :273  b > z (signed)
	d = (m2 & C.c2) | (~m2 & d);                                        // :274
	zz = pk_max_u(zz, b);                                               // :131
	zz = pk_min_u(zz, C.M);                                             // :132
	un = pk_sub(zz, vp); vn = pk_sub(zz, u);                            // :133-134
	const unsigned zq = pk_sub(zz, C.q);
	const unsigned a2 = pk_sub(a, zq), b2 = pk_sub(b, zq);
	xn = pk_max_i(a2, 0u); yn = pk_max_i(b2, 0u);                       // :277-280
	d |= pk_min_u(xn, C.c8) | pk_min_u(yn, C.c16);                      // :281-282: a2 > 0 <=> xn != 0 (then xn >= 0x100)
}

// max over the 64 lanes of two values at once; the two chains fill each other's DPP wait states
__device__ __forceinline__ void wave_max2_i32(int v1, int v2, int &m1, int &m2)
{
	int t1, t2;
	asm("s_nop 4\n\t"
	    "v_max_i32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 0\n\t"
	    "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 0\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 0\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
	    "s_nop 0\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
	    "s_nop 0\n\t"
	    "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	    "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
	    "s_nop 1"
	    : "=&v"(t1), "=&v"(t2) : "v"(v1), "v"(v2));
	m1 = __builtin_amdgcn_readlane(t1, 63); m2 = __builtin_amdgcn_readlane(t2, 63);
}

struct PairState {
	// both alignments, value << 8 per half: slot A (t = st + lane), slot B (t = st + 64 + lane, lanes 0..15)
	unsigned XA, VA, UA, YA, ZA, XB, VB, UB, YB;
	// per alignment
	int HA[2], HB[2];
	unsigned T1A[2], T0A[2], T1B[2], T0B[2];
	const unsigned *qptr[2];
	int qoffB;
	int st, edge_h[2];
	int ez_max[2], ez_max_t[2], ez_max_q[2];
};

// ksw_apply_zdrop and the bookkeeping of the running maximum (:312-357, :88-104) for one alignment on a steady
// diagonal -- the tail of narrow_diag<.., ND_STEADY>.  Returns true when the sweep of this alignment must stop.
__device__ __forceinline__ bool pair_ez(int hA, int hB, int hasB, unsigned long long mInB, int max_H, int st, int st0, int en0, int loA, int r,
                                        int zdrop, int e, int &ez_max, int &ez_max_t, int &ez_max_q)
{
	const bool improves = max_H > ez_max;
	if (!improves && (zdrop < 0 || ez_max - max_H <= zdrop)) return false;
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
	if (improves) { ez_max = max_H; ez_max_t = t; ez_max_q = dq; return false; }
	if (t < ez_max_t || dq < ez_max_q) return false;
	const int tl = t - ez_max_t, ql = dq - ez_max_q;
	const int l = tl > ql ? tl - ql : ql - tl;
	return ez_max - max_H > zdrop + l * e;
}

// One steady anti-diagonal of both alignments (narrow_diag<false, ND_STEADY> twice, sharing everything wave-uniform).
// Returns bit k set when alignment k must stop.
__device__ __forceinline__ int pair_diag(PairState &S, const NarrowEnv &E1, const NarrowEnv &E2, const PairConst &C, const int r, const int st0, const int en0)
{
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	const int nst = st0 & ~15, en = en0 | 15;
	int ex = 0, ev = 0;
	if (nst != S.st) {
		// the band origin moved one block right: rotate the registers 16 lanes, re-seed slot B
		ex = __builtin_amdgcn_readlane((int)S.XA, 15);
		ev = __builtin_amdgcn_readlane((int)S.VA, 15);
		S.edge_h[0] = __builtin_amdgcn_readlane(S.HA[0], 15); S.edge_h[1] = __builtin_amdgcn_readlane(S.HA[1], 15);
		const unsigned zf = ((unsigned)narrow_z(S.T0B[0], S.T1B[0], S.qptr[0][S.qoffB + 1] >> 16)) | (unsigned)narrow_z(S.T0B[1], S.T1B[1], S.qptr[1][S.qoffB + 1]);
		const unsigned zB = lane_in(0x7fffull) ? zf : C.ZW;             // steady: st0 was 16k+15, so last_sc = 78
		S.XA = rot16(S.XA, S.XB, lane); S.VA = rot16(S.VA, S.VB, lane);
		S.UA = rot16(S.UA, S.UB, lane); S.YA = rot16(S.YA, S.YB, lane);
		S.ZA = rot16(S.ZA, zB, lane);
		S.HA[0] = (int)rot16((unsigned)S.HA[0], (unsigned)S.HB[0], lane); S.HA[1] = (int)rot16((unsigned)S.HA[1], (unsigned)S.HB[1], lane);
		S.st = nst;
		S.qptr[0] += 16; S.qptr[1] += 16;
		S.XB = S.VB = S.UB = S.YB = 0; S.HB[0] = S.HB[1] = KSW_NEG_INF;
		{
			const uint2 ta = E1.tbl[E1.tg[nst + lane]], tb = E1.tbl[E1.tg[nst + 64 + (lane & 15)]];
			S.T1A[0] = ta.x; S.T0A[0] = ta.y; S.T1B[0] = tb.x; S.T0B[0] = tb.y;
			const uint2 tc = E2.tbl[E2.tg[nst + lane]], td = E2.tbl[E2.tg[nst + 64 + (lane & 15)]];
			S.T1A[1] = tc.x; S.T0A[1] = tc.y; S.T1B[1] = td.x; S.T0B[1] = td.y;
		}
	}
	// neighbours of r-1 (taken before anything is overwritten); lane 0 gets the block edge x1, v1 (:207-211)
	const unsigned xpA = (unsigned)set_lane0(ex, 0, dppz_shr1((int)S.XA)), vpA = (unsigned)set_lane0(ev, 0, dppz_shr1((int)S.VA));
	const int HpA0 = dppz_shr1(S.HA[0]), HpA1 = dppz_shr1(S.HA[1]);     // en0 is never on lane 0
	const int st = S.st;
	const int loA = st0 - st, hiT = en0 - st, nTop = en - st;
	const int sc = loA + 63;
	const int hasB = (nTop >> 6) & 1;
	uint8_t *pr1 = E1.p + (size_t)r * E1.ncol, *pr2 = E2.p + (size_t)r * E2.ncol;
	const bool geLo = lane >= loA;
	const bool inTA = geLo && lane <= hiT;
	const bool spA = lane == hiT;                                       // H[en0] comes from H[en0-1] + u (:318); in slot B if hiT >= 64
	int hB0 = INTMIN, hB1 = INTMIN;
	unsigned long long mInB = 0;
	// ---- slot B (block 4) ------------------------------------------------------------
	if (hasB) {
		const int exB = __builtin_amdgcn_readlane((int)S.XA, 63), evB = __builtin_amdgcn_readlane((int)S.VA, 63);
		const int He0 = __builtin_amdgcn_readlane(S.HA[0], 63), He1 = __builtin_amdgcn_readlane(S.HA[1], 63);
		const unsigned xpB = (unsigned)set_lane0(exB, 0, dppz_shr1((int)S.XB)), vpB = (unsigned)set_lane0(evB, 0, dppz_shr1((int)S.VB));
		const int HpB0 = set_lane0(He0, 0, dppz_shr1(S.HB[0])), HpB1 = set_lane0(He1, 0, dppz_shr1(S.HB[1]));
		const unsigned zf = ((unsigned)narrow_z(S.T0B[0], S.T1B[0], S.qptr[0][S.qoffB] >> 16)) | (unsigned)narrow_z(S.T0B[1], S.T1B[1], S.qptr[1][S.qoffB]);
		const unsigned zB = lane_in(lane_range(0, sc - 64)) ? zf : C.ZW;
		mInB = lane_range(0, hiT - 64 < 15 ? hiT - 64 : 15);               // loA <= 15, so block 4 is never below the band
		if (lane < 16) {
			unsigned xn, vn, un, yn, d;
			pair_cell(zB, xpB, vpB, S.UB, S.YB, C, xn, vn, un, yn, d);
			S.XB = xn; S.VB = vn; S.UB = un; S.YB = yn;
			pr1[64 + lane] = (uint8_t)d; pr2[64 + lane] = (uint8_t)(d >> 16);   // :283
			const bool sp = 64 + lane == hiT;
			const unsigned uv = sp ? un : vn;
			const int h0 = (sp ? HpB0 : S.HB[0]) + (int)((uv >> 8) & 0xffu) - E1.qe;   // :318, :323-329
			const int h1 = (sp ? HpB1 : S.HB[1]) + (int)(uv >> 24) - E1.qe;
			const bool inT = lane_in(mInB);
			hB0 = inT ? h0 : INTMIN; S.HB[0] = inT ? h0 : S.HB[0];
			hB1 = inT ? h1 : INTMIN; S.HB[1] = inT ? h1 : S.HB[1];
		}
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	int hA0, hA1;
	{
		const unsigned znew = ((unsigned)narrow_z(S.T0A[0], S.T1A[0], *S.qptr[0] >> 16)) | (unsigned)narrow_z(S.T0A[1], S.T1A[1], *S.qptr[1]);
		S.qptr[0] -= 1; S.qptr[1] -= 1;
		S.ZA = geLo ? znew : S.ZA;                                          // :214-228
		unsigned xn, vn, un, yn, d;
		pair_cell(S.ZA, xpA, vpA, S.UA, S.YA, C, xn, vn, un, yn, d);
		const unsigned uv = spA ? un : vn;
		const int h0 = (spA ? HpA0 : S.HA[0]) + (int)((uv >> 8) & 0xffu) - E1.qe;       // :318, :323-329
		const int h1 = (spA ? HpA1 : S.HA[1]) + (int)(uv >> 24) - E1.qe;
		S.XA = xn; S.VA = vn; S.UA = un; S.YA = yn;                         // a steady band always covers blocks 0..3
		pr1[lane] = (uint8_t)d; pr2[lane] = (uint8_t)(d >> 16);
		hA0 = inTA ? h0 : INTMIN; S.HA[0] = inTA ? h0 : S.HA[0];
		hA1 = inTA ? h1 : INTMIN; S.HA[1] = inTA ? h1 : S.HA[1];
	}
	// ---- exact max (:320-348), one reduction for both --------------------------------
	int max0, max1;
	wave_max2_i32(hA0, hA1, max0, max1);
	if (hasB) {
		int b0, b1;
		wave_max2_i32(hB0, hB1, b0, b1);
		max0 = b0 > max0 ? b0 : max0; max1 = b1 > max1 ? b1 : max1;
	}
	int stop = 0;
	if (pair_ez(hA0, hB0, hasB, mInB, max0, st, st0, en0, loA, r, E1.zdrop, E1.e, S.ez_max[0], S.ez_max_t[0], S.ez_max_q[0])) stop |= 1;
	if (pair_ez(hA1, hB1, hasB, mInB, max1, st, st0, en0, loA, r, E2.zdrop, E2.e, S.ez_max[1], S.ez_max_t[1], S.ez_max_q[1])) stop |= 2;
	return stop;
}

// The steady diagonals [r, r_end) of two alignments that both stand at diagonal r in steady shape (heads done, not
// stopped).  On return r is the next diagonal of both; bit k of the result says alignment k has to stop (it z-dropped
// on diagonal r - 1).
__device__ inline int pair_steady(NarrowState &F1, NarrowState &F2, const NarrowEnv &E1, const NarrowEnv &E2, int &r, int r_end)
{
	const int w = E1.w;
	PairState S;
	PairConst C;
	C.M = (E1.M24 >> 16) | E1.M24; C.q = ((unsigned)E1.q24 >> 16) | (unsigned)E1.q24; C.ZW = ((unsigned)E1.ZW24 >> 16) | (unsigned)E1.ZW24;
	C.c15 = 0x000f000fu; C.c2 = 0x00020002u; C.c8 = 0x00080008u; C.c16 = 0x00100010u;
	S.XA = pair_pack(F1.XA, F2.XA); S.VA = pair_pack(F1.VA, F2.VA); S.UA = pair_pack(F1.UA, F2.UA); S.YA = pair_pack(F1.YA, F2.YA);
	S.ZA = pair_pack(F1.ZA, F2.ZA);
	S.XB = pair_pack(F1.XB, F2.XB); S.VB = pair_pack(F1.VB, F2.VB); S.UB = pair_pack(F1.UB, F2.UB); S.YB = pair_pack(F1.YB, F2.YB);
	S.HA[0] = F1.HA; S.HA[1] = F2.HA; S.HB[0] = F1.HB; S.HB[1] = F2.HB;
	S.T1A[0] = F1.T1A; S.T0A[0] = F1.T0A; S.T1B[0] = F1.T1B; S.T0B[0] = F1.T0B;
	S.T1A[1] = F2.T1A; S.T0A[1] = F2.T0A; S.T1B[1] = F2.T1B; S.T0B[1] = F2.T0B;
	S.qptr[0] = F1.qptr; S.qptr[1] = F2.qptr; S.qoffB = F1.qoffB;
	S.st = F1.st; S.edge_h[0] = F1.edge_h; S.edge_h[1] = F2.edge_h;
	S.ez_max[0] = F1.ez_max; S.ez_max_t[0] = F1.ez_max_t; S.ez_max_q[0] = F1.ez_max_q;
	S.ez_max[1] = F2.ez_max; S.ez_max_t[1] = F2.ez_max_t; S.ez_max_q[1] = F2.ez_max_q;
	int stop = 0;
	int st0 = (r - w + 1) >> 1, en0 = (r + w) >> 1;
	while (r < r_end) {
		stop = pair_diag(S, E1, E2, C, r, st0, en0);
		const int up = (r + w) & 1;
		en0 += up; st0 += 1 - up;
		++r;
		if (stop) break;
	}
	F1.XA = pair_lo(S.XA); F2.XA = pair_hi(S.XA); F1.VA = pair_lo(S.VA); F2.VA = pair_hi(S.VA);
	F1.UA = pair_lo(S.UA); F2.UA = pair_hi(S.UA); F1.YA = pair_lo(S.YA); F2.YA = pair_hi(S.YA);
	F1.ZA = pair_lo(S.ZA); F2.ZA = pair_hi(S.ZA);
	F1.XB = pair_lo(S.XB); F2.XB = pair_hi(S.XB); F1.VB = pair_lo(S.VB); F2.VB = pair_hi(S.VB);
	F1.UB = pair_lo(S.UB); F2.UB = pair_hi(S.UB); F1.YB = pair_lo(S.YB); F2.YB = pair_hi(S.YB);
	F1.HA = S.HA[0]; F2.HA = S.HA[1]; F1.HB = S.HB[0]; F2.HB = S.HB[1];
	F1.T1A = S.T1A[0]; F1.T0A = S.T0A[0]; F1.T1B = S.T1B[0]; F1.T0B = S.T0B[0];
	F2.T1A = S.T1A[1]; F2.T0A = S.T0A[1]; F2.T1B = S.T1B[1]; F2.T0B = S.T0B[1];
	F1.qptr = S.qptr[0]; F2.qptr = S.qptr[1];
	F1.st = F2.st = S.st; F1.edge_h = S.edge_h[0]; F2.edge_h = S.edge_h[1];
	F1.ez_max = S.ez_max[0]; F1.ez_max_t = S.ez_max_t[0]; F1.ez_max_q = S.ez_max_q[0];
	F2.ez_max = S.ez_max[1]; F2.ez_max_t = S.ez_max_t[1]; F2.ez_max_q = S.ez_max_q[1];
	F1.last_sc = F2.last_sc = ((r - w) >> 1) - S.st + 63;      // of diagonal r-1, as at the end of the single steady loop
	return stop;
}

// Two jobs with the same parameters, left-aligned gaps.  Returns false when a job is not for the narrow sweep at all (a
// code outside the alphabet): the caller then runs both one at a time.  Otherwise both results are final.
__device__ inline bool ksw_wave_narrow_pair(const uint8_t *q1, int qlen1, const uint8_t *t1, int tlen1, const uint8_t *q2, int qlen2,
                                            const uint8_t *t2, int tlen2, const KswParams P, uint8_t *lds1, uint8_t *lds2, uint8_t *p1, uint8_t *p2,
                                            uint32_t *ct1, uint32_t *ct2, int cig_cap, KswOut &out1, KswOut &out2)
{
	NarrowState F1, F2;
	NarrowEnv E1, E2;
	const int pr1 = narrow_prepare(q1, qlen1, t1, tlen1, P, lds1, p1, out1, F1, E1);
	const int pr2 = narrow_prepare(q2, qlen2, t2, tlen2, P, lds2, p2, out2, F2, E2);
	if (pr1 == 2 || pr2 == 2) return false;
	int r1 = 0, r2 = 0; bool stop1 = false, stop2 = false, tr1 = true, tr2 = true;
	if (!pr1) narrow_head<false>(F1, E1, r1, stop1, tr1);
	if (!pr2) narrow_head<false>(F2, E2, r2, stop2, tr2);
	const int w = P.w;
	if (!pr1 && !pr2 && !stop1 && !stop2 && w >= 49 && narrow_roomy(qlen1, tlen1, w) && narrow_roomy(qlen2, tlen2, w)) {
		const int h1 = narrow_r_hi(qlen1, tlen1, w), h2 = narrow_r_hi(qlen2, tlen2, w);
		const int r_end = h1 < h2 ? h1 : h2;
		if (r1 == r2 && r1 < r_end) {                             // both stand at diagonal w + 31
			int r = r1;
			const int st = pair_steady(F1, F2, E1, E2, r, r_end);
			// an alignment that stopped did so ON diagonal r - 1 (the single loops leave r there); the other goes on at r
			stop1 = (st & 1) != 0; stop2 = (st & 2) != 0;
			r1 = stop1 ? r - 1 : r; r2 = stop2 ? r - 1 : r;
			tr1 = tr2 = false;
		}
	}
	if (!pr1) { narrow_rest<false>(F1, E1, r1, stop1, tr1); }
	if (!pr2) { narrow_rest<false>(F2, E2, r2, stop2, tr2); }
	WSYNC();
	if (!pr1) narrow_finish(F1, E1, stop1, P.flag, p1, ct1, cig_cap, out1);
	if (!pr2) narrow_finish(F2, E2, stop2, P.flag, p2, ct2, cig_cap, out2);
	return true;
}

}  // namespace ihp
