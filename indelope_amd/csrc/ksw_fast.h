// ksw_fast.h -- register-resident ksw2 anti-diagonal sweep for narrow bands (0 <= w <= 62),
// the production case (indelope.nim:221: bw=50).  Same contract and bit-exact results as
// ksw_wave() in ksw_dev.h (reference: src/ksw2/csrc/ksw2_extz2_sse.c:113-388).
//
// The reference's computed band on a diagonal is at most five 16-byte SSE blocks
// [st, en] = [st0/16*16, (en0+16)/16*16-1].  Lanes are band-relative:
//     slot A: lane l  <->  t = st + l          (blocks 0..3, all 64 lanes)
//     slot B: lane l  <->  t = st + 64 + l     (block 4, l < 16)
// u, v, x, y, s, H and the target base of each cell live in VGPRs; the t-1 neighbour
// comes from a DPP wave_shr:1 (the SSE code's _mm_slli_si128 carry, :119-125) with the
// block edge x1/v1 (:207-211) injected as the DPP `old` value.  When st advances by 16
// (every ~32 diagonals) the registers are rotated 16 lanes with two ds_bpermute each and
// slot B is re-seeded from the zeroed state of :173.  The per-diagonal exact max (:312-349)
// is a DPP max + ballot; ties (rare) are resolved with the reference's stride-4 order on
// the scalar unit.  Only the reversed query and the target bytes are kept in LDS.
#pragma once
#include "ksw_dev.h"

namespace ihp {

__host__ __device__ __forceinline__ size_t ksw_fast_lds_bytes(int qlen, int tlen)
{
	return (size_t)((tlen + 15) / 16) * 16 + 96 + 16 + (size_t)((qlen + 15) / 16) * 16 + 96;
}

__device__ __forceinline__ bool ksw_fast_ok(int w) { return w >= 0 && w <= 62; }

__device__ __forceinline__ unsigned dpp_shr1(unsigned v, unsigned edge)
{   // lane l gets v[l-1]; lane 0 gets `edge`
	return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned rot16(unsigned a, unsigned b, int lane)
{   // new A lane l = old A lane l+16 (l < 48), old B lane l-48 (l >= 48)
	const unsigned ra = (unsigned)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, (int)a);
	const unsigned rb = (unsigned)__builtin_amdgcn_ds_bpermute(((lane - 48) & 63) << 2, (int)b);
	return lane < 48 ? ra : rb;
}

struct FastConst { unsigned qe2, sc_mch, sc_mis, m1, max_sc8, q8; };

__device__ __forceinline__ unsigned score_byte(unsigned sfb, unsigned qb, const FastConst &C)
{   // :219-226
	const unsigned sv = sfb == qb ? C.sc_mch : C.sc_mis;      // written as selects: no exec-mask branches
	const bool wild = (sfb == C.m1) | (qb == C.m1);
	return wild ? 0u : sv;
}

// One cell of the recurrence (:116-137 + :262-310).  In: sv, xt1, vt1, ut, yt.  Out: new x,v,u,y and d.
template <bool RIGHT>
__device__ __forceinline__ void ksw_cell(unsigned sv, unsigned xt1, unsigned vt1, unsigned ut, unsigned yt,
                                         const FastConst &C, unsigned &xn, unsigned &vn, unsigned &un, unsigned &yn, unsigned &d)
{
	unsigned z = (sv + C.qe2) & 0xff;
	unsigned a = (xt1 + vt1) & 0xff, b = (yt + ut) & 0xff;
	if (!RIGHT) d = sext8(a) > sext8(z) ? 1 : 0;                          // :265
	else          d = sext8(z) > sext8(a) ? 0 : 1;                          // :291
	z = sext8(z) > 0 ? z : 0;                                               // :271 (SSE2 path)
	z = z > a ? z : a;                                                      // :272 unsigned max
	if (!RIGHT) { if (sext8(b) > sext8(z)) d = 2; }                       // :273-274
	else          { if (!(sext8(z) > sext8(b))) d = 2; }                    // :299-300
	z = z > b ? z : b;                                                      // :131
	z = z < C.max_sc8 ? z : C.max_sc8;                                      // :132
	un = (z - vt1) & 0xff; vn = (z - ut) & 0xff;                            // :133-134
	z = (z - C.q8) & 0xff;
	a = (a - z) & 0xff; b = (b - z) & 0xff;
	if (!RIGHT) {
		const bool ta = sext8(a) > 0, tb = sext8(b) > 0;                    // :277-282
		xn = ta ? a : 0; yn = tb ? b : 0;
		d |= (ta ? 0x08u : 0u) | (tb ? 0x10u : 0u);
	} else {
		const bool ta = 0 > sext8(a), tb = 0 > sext8(b);                    // :303-308
		xn = ta ? 0 : a; yn = tb ? 0 : b;
		d |= (ta ? 0u : 0x08u) | (tb ? 0u : 0x10u);
	}
}

// Everything a diagonal reads and writes; lives in registers (the struct is scalar-replaced after inlining).
struct FastState {
	unsigned xA, vA, uA, yA, sA, sfA;                    // slot A cells
	unsigned xB, vB, uB, yB, sfB;                        // slot B cells
	int rlB;                                             // slot B: diagonal of the last score refresh (lazy s[])
	int HA, HB;
	int st;                                              // current computed-band origin (multiple of 16)
	unsigned edge_x, edge_v;                             // x[st-1], v[st-1] when valid (:207-210)
	int edge_h;                                          // H[st-1]: frozen once t = st-1 left the band
	int last_st, last_en;
	int ez_max, ez_max_t, ez_max_q, mqe, mqe_t, mte, mte_q, score, zdropped;
};

struct FastEnv {
	const uint8_t *tg, *qr;
	uint8_t *p;
	int qlen, tlen, w, ncol, qe, e, zdrop;
	bool with_cigar;
};

// One anti-diagonal.  STEADY = the band is limited by w on both sides, ends before the last target base, does
// not reach the last query base and lies above t == r (true for w+32 <= r < min(2 tlen, 2 qlen) - w - 2): the
// start/end special cases of the reference (:212, :349, :351-357) cannot apply and are compiled out.
// Returns true when the sweep must stop (band exit :200-203 or z-drop :98-101).
template <bool RIGHT, bool STEADY>
__device__ __forceinline__ bool fast_diag(FastState &F, const FastEnv &E, const FastConst &C, int r)
{
	const int lane = lane_id();
	const int INTMIN = -0x7fffffff - 1;
	int st0, en0, nst, en;
	if (STEADY) {
		st0 = (r - E.w + 1) >> 1; en0 = (r + E.w) >> 1;
		nst = st0 & ~15; en = ((en0 + 16) & ~15) - 1;
	} else if (!ksw_band(r, E.qlen, E.tlen, E.w, st0, en0, nst, en)) { F.zdropped = 1; return true; }   // :200-203
	unsigned ex = 0, ev = 0;
	if (nst != F.st) {
		// the band origin moved one block right: rotate the registers 16 lanes, re-seed slot B
		const bool valid = STEADY || (F.st + 15 >= F.last_st && F.st + 15 <= F.last_en);
		ex = valid ? (unsigned)__builtin_amdgcn_readlane((int)F.xA, 15) : 0u;
		ev = valid ? (unsigned)__builtin_amdgcn_readlane((int)F.vA, 15) : 0u;
		F.edge_h = __builtin_amdgcn_readlane(F.HA, 15);
		F.xA = rot16(F.xA, F.xB, lane); F.vA = rot16(F.vA, F.vB, lane);
		F.uA = rot16(F.uA, F.uB, lane); F.yA = rot16(F.yA, F.yB, lane);
		const unsigned sBm = F.rlB < 0 ? 0u : score_byte(F.sfB, E.qr[E.qlen - 1 - F.rlB + F.st + 64 + (lane & 15)], C);
		F.sA = rot16(F.sA, sBm, lane); F.sfA = rot16(F.sfA, F.sfB, lane);
		F.HA = (int)rot16((unsigned)F.HA, (unsigned)F.HB, lane);
		F.st = nst;
		F.xB = F.vB = F.uB = F.yB = 0; F.rlB = -1; F.HB = KSW_NEG_INF;
		F.sfB = E.tg[F.st + 64 + (lane & 15)];
	} else if (!STEADY && F.st == 0) { ev = r ? C.q8 : 0; }       // :211; for st > 0 without a move x1 = v1 = 0 (:210)
	const int st = F.st;
	const int loA = st0 - st;                            // first true-band lane
	const int hiT = en0 - st;                            // last true-band lane (may be >= 64: slot B)
	const int nTop = en - st;                            // last computed lane (<= 79)
	const int sc = st0 + ((en0 - st0) / 16 + 1) * 16 - 1 - st;   // last refreshed score lane (:215)
	const int qbase = E.qlen - 1 - r + st;               // qrr[t] = qr[qbase + lane]
	uint8_t *pr = E.p + (size_t)r * E.ncol;
	// neighbours of r-1 (taken before anything is overwritten)
	const unsigned xpA = dpp_shr1(F.xA, ex), vpA = dpp_shr1(F.vA, ev);
	const int HpA = (int)dpp_shr1((unsigned)F.HA, (unsigned)F.edge_h);
	int hB = INTMIN, hA = INTMIN;
	const int spec = (STEADY || (r > 0 && en0 > 0)) ? hiT : -1000;      // lane of the H[en0] special case (:318)
	// ---- slot B (block 4) ------------------------------------------------------------
	F.rlB = lane <= sc - 64 ? r : F.rlB;                                // :214-228 runs past en; value formed on use
	if (nTop >= 64) {
		const unsigned exB = (unsigned)__builtin_amdgcn_readlane((int)F.xA, 63), evB = (unsigned)__builtin_amdgcn_readlane((int)F.vA, 63);
		const int HeB = __builtin_amdgcn_readlane(F.HA, 63);
		const unsigned xpB = dpp_shr1(F.xB, exB), vpB = dpp_shr1(F.vB, evB);
		const int HpB = (int)dpp_shr1((unsigned)F.HB, (unsigned)HeB);
		const unsigned sB = F.rlB < 0 ? 0u : score_byte(F.sfB, E.qr[E.qlen - 1 - F.rlB + st + 64 + (lane & 15)], C);
		if (lane <= nTop - 64) {
			unsigned ut = F.uB, yt = F.yB;
			if (!STEADY && st + 64 + lane == r) { yt = 0; ut = r ? C.q8 : 0; }   // :212
			unsigned xn, vn, un, yn, d;
			ksw_cell<RIGHT>(sB, xpB, vpB, ut, yt, C, xn, vn, un, yn, d);
			F.xB = xn; F.vB = vn; F.uB = un; F.yB = yn;
			if (E.with_cigar) pr[64 + lane] = (uint8_t)d;        // :283
			const int l = 64 + lane;
			if (l >= loA && l <= hiT) {
				hB = (l == spec ? HpB + (int)un : F.HB + (int)vn) - E.qe;   // :318, :323-329
				F.HB = hB;
			}
		}
	}
	// ---- slot A (blocks 0..3) --------------------------------------------------------
	{
		const unsigned qbA = E.qr[qbase + lane];
		const unsigned snew = score_byte(F.sfA, qbA, C);
		F.sA = ((lane >= loA) & (lane <= sc)) ? snew : F.sA;       // :214-228
		if (!STEADY && r <= en && r - st < 64) {             // :212 (only while the band still touches t == r)
			const bool tr = st + lane == r;
			F.yA = tr ? 0u : F.yA; F.uA = tr ? (r ? C.q8 : 0u) : F.uA;
		}
		unsigned xn, vn, un, yn, d;
		ksw_cell<RIGHT>(F.sA, xpA, vpA, F.uA, F.yA, C, xn, vn, un, yn, d);
		const bool act = STEADY || lane <= nTop;             // a steady band always covers blocks 0..3
		const bool inT = act & (lane >= loA) & (lane <= hiT);
		int h;
		if (STEADY || r > 0) h = (lane == spec ? HpA + (int)un : F.HA + (int)vn) - E.qe;   // :318, :323-329
		else h = (int)vn - E.qe - E.qe;                      // :349
		if (STEADY) {
			F.xA = xn; F.vA = vn; F.uA = un; F.yA = yn;
			if (E.with_cigar) pr[lane] = (uint8_t)d;
		} else {
			F.xA = act ? xn : F.xA; F.vA = act ? vn : F.vA; F.uA = act ? un : F.uA; F.yA = act ? yn : F.yA;
			if (E.with_cigar && act) pr[lane] = (uint8_t)d;
		}
		hA = inT ? h : INTMIN;
		F.HA = inT ? h : F.HA;
	}
	// ---- exact max (:320-348) ----------------------------------------------------------
	int max_H = wave_max_i32(hA), max_t;
	if (nTop >= 64) { const int mb = wave_max_i32(hB); max_H = mb > max_H ? mb : max_H; }
	{
		const unsigned long long mA = ballot(hA == max_H && lane >= loA && lane <= hiT);
		const unsigned long long mB = nTop >= 64 ? ballot(hB == max_H && lane < 16 && 64 + lane >= loA && 64 + lane <= hiT) : 0ull;
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
	// ---- ez updates (:351-357) -----------------------------------------------------------
	int Hen0 = 0;
	if (!STEADY && (en0 == E.tlen - 1 || r - st0 == E.qlen - 1)) {
		Hen0 = hiT < 64 ? __builtin_amdgcn_readlane(hA, hiT & 63) : __builtin_amdgcn_readlane(hB, (hiT - 64) & 63);
		const int Hst0 = __builtin_amdgcn_readlane(hA, loA);
		if (en0 == E.tlen - 1 && Hen0 > F.mte) { F.mte = Hen0; F.mte_q = r - en; }        // rounded en (:352)
		if (r - st0 == E.qlen - 1 && Hst0 > F.mqe) { F.mqe = Hst0; F.mqe_t = st0; }
	}
	{                                                                             // ksw_apply_zdrop :88-104
		const int t = max_t;
		if (max_H > F.ez_max) { F.ez_max = max_H; F.ez_max_t = t; F.ez_max_q = r - t; }
		else if (t >= F.ez_max_t && r - t >= F.ez_max_q) {
			const int tl = t - F.ez_max_t, ql = (r - t) - F.ez_max_q;
			const int l = tl > ql ? tl - ql : ql - tl;
			if (E.zdrop >= 0 && F.ez_max - max_H > E.zdrop + l * E.e) { F.zdropped = 1; return true; }
		}
	}
	if (!STEADY && r == E.qlen + E.tlen - 2 && en0 == E.tlen - 1) F.score = Hen0;           // :356-357
	F.last_st = st; F.last_en = en;
	return false;
}

template <bool RIGHT>
__device__ inline void ksw_wave_fast(const uint8_t *query, int qlen, const uint8_t *target, int tlen,
                                     const KswParams P, uint8_t *lds, uint8_t *p, uint32_t *cig_tmp, int cig_cap,
                                     KswOut &out, long long *pacc = nullptr)
{
	const long long tc0 = pacc ? (long long)clock64() : 0;
	const int lane = lane_id();
	const int w = P.w;
	const int q = P.q, e = P.e, qe = q + e, flag = P.flag;
	out.max = 0; out.zdropped = 0; out.max_q = out.max_t = out.mqe_t = out.mte_q = -1;   // :81-86
	out.mqe = out.mte = out.score = KSW_NEG_INF; out.n_cigar = 0;
	if (P.m <= 0 || qlen <= 0 || tlen <= 0) return;      // :147
	if (-P.min_sc > 2 * (q + e)) return;                 // :171
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	const int TP = (tlen + 15) / 16 * 16 + 96, QR = (qlen + 15) / 16 * 16 + 96;
	uint8_t *tg = lds;                                   // target codes, zero padded (sf of :175,:188)
	uint8_t *qr = lds + TP + 16;                         // reversed query, zero padded on both sides (:187):
	                                                     // every index qlen-1-r+t a lane can form lies in [-16, QR)
	FastConst C;
	C.qe2 = (unsigned)(qe * 2) & 0xff; C.sc_mch = (unsigned)P.sc_mch & 0xff; C.sc_mis = (unsigned)P.sc_mis & 0xff;
	C.m1 = (unsigned)(P.m - 1) & 0xff; C.max_sc8 = (unsigned)(P.sc_mch + qe * 2) & 0xff; C.q8 = (unsigned)q & 0xff;
	const bool with_cigar = !(flag & KSW_EZ_SCORE_ONLY);
	for (int i = lane; i < TP; i += 64) tg[i] = i < tlen ? (P.encode_ascii ? enc_base(target[i]) : target[i]) : 0;
	if (lane < 16) qr[lane - 16] = 0;
	for (int i = lane; i < QR; i += 64) {
		uint8_t b = 0;
		if (i < qlen) { b = query[qlen - 1 - i]; if (P.encode_ascii) b = enc_base(b); }
		qr[i] = b;
	}
	WSYNC();
	const long long tc1 = pacc ? (long long)clock64() : 0;

	FastState F;
	F.xA = F.vA = F.uA = F.yA = F.sA = 0; F.sfA = tg[lane];
	F.xB = F.vB = F.uB = F.yB = 0; F.sfB = tg[64 + (lane & 15)];
	F.rlB = -1; F.HA = F.HB = KSW_NEG_INF; F.st = 0;
	F.edge_x = F.edge_v = 0; F.edge_h = KSW_NEG_INF; F.last_st = F.last_en = -1;
	F.ez_max = 0; F.ez_max_t = F.ez_max_q = -1; F.mqe = F.mte = F.score = KSW_NEG_INF; F.mqe_t = F.mte_q = -1; F.zdropped = 0;
	FastEnv E;
	E.tg = tg; E.qr = qr; E.p = p; E.qlen = qlen; E.tlen = tlen; E.w = w; E.ncol = ncol; E.qe = qe; E.e = e;
	E.zdrop = P.zdrop; E.with_cigar = with_cigar;
	const int total = qlen + tlen - 1;
	// steady diagonals: st0 = (r-w+1)>>1 > r-qlen+1, en0 = (r+w)>>1 < tlen-1, en < r
	const int r_lo = w + 32;
	int r_hi = 2 * tlen - 3 - w < 2 * qlen - w - 3 ? 2 * tlen - 3 - w : 2 * qlen - w - 3;
	r_hi = r_hi + 1 < total ? r_hi + 1 : total;
	int r = 0;
	bool stop = false;
	for (; r < total && r < r_lo && !stop; ++r) stop = fast_diag<RIGHT, false>(F, E, C, r);
	if (w >= 49) for (; r < r_hi && !stop; ++r) stop = fast_diag<RIGHT, true>(F, E, C, r);   // w >= 49: a steady band spans blocks 0..3
	for (; r < total && !stop; ++r) stop = fast_diag<RIGHT, false>(F, E, C, r);
	out.max = F.ez_max; out.zdropped = F.zdropped; out.max_q = F.ez_max_q; out.max_t = F.ez_max_t;
	out.mqe = F.mqe; out.mqe_t = F.mqe_t; out.mte = F.mte; out.mte_q = F.mte_q; out.score = F.zdropped ? KSW_NEG_INF : F.score;   // (the reference tests the z-drop before it takes the score of the last diagonal, :355-357: a sweep that stopped has none)
	WSYNC();
	const long long tc2 = pacc ? (long long)clock64() : 0;
	if (pacc && lane == 0) { pacc[0] += tc1 - tc0; pacc[1] += tc2 - tc1; pacc[3] += 1; }
	if (!with_cigar) return;
	ksw_backtrack_wave(p, ncol, qlen, tlen, w, flag, F.zdropped, F.ez_max_t, F.ez_max_q, cig_tmp, cig_cap, out);
	if (pacc && lane == 0) pacc[2] += (long long)clock64() - tc2;
}

}  // namespace ihp
