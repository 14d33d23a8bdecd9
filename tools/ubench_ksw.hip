// ubench_ksw.hip -- what a ksw2 anti-diagonal costs on gfx950 under different cell encodings (diagnostics, not product).
//
// Every variant runs the dependency structure of the steady diagonal of ksw_narrow.h (neighbour shift, score lookup,
// cell, four traceback compares, H update, the two threshold compares) `iters` times per wave at 8 waves per SIMD and
// reports wall-clock nanoseconds per (alignment, diagonal).  The values are synthetic; only the instruction stream matters.
//   V0  one alignment per wave, int8 in the top byte, compare -> v_addc traceback (the round-3 kernel)
//   V1  same, the four compare masks leave by two s_store_dwordx4 per diagonal
//   V2  two alignments per wave in 16-bit halves (v_pk_*), SDWA compares -> v_addc, 32-bit H per alignment
//   V3  as V2, masks leave by four s_store_dwordx4
//   V4  as V2 with H kept as a packed 16-bit offset
//   V5  as V3 with H kept as a packed 16-bit offset
//   V6  packed, traceback bits gathered from saturating differences (no compares), packed H
// Also the issue rate of the few ops the earlier table lacks.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_ksw tools/ubench_ksw.hip && gpurun_out/ubench_ksw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ int dppz_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ bool lane_in(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ unsigned shl1_in(unsigned acc, unsigned long long m)
{
	asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(acc) : "s"(m) : "vcc");
	return acc;
}
#define PK2(NAME, OP) __device__ __forceinline__ unsigned NAME(unsigned a, unsigned b) { unsigned r; asm(OP " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PK2(pk_add, "v_pk_add_u16")
PK2(pk_sub, "v_pk_sub_u16")
PK2(pk_max_u, "v_pk_max_u16")
PK2(pk_min_u, "v_pk_min_u16")
PK2(pk_max_i, "v_pk_max_i16")
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b) { unsigned r; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned pk_lshr(unsigned a, unsigned n) { unsigned r; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(r) : "v"(n), "v"(a)); return r; }
__device__ __forceinline__ unsigned pk_min_us(unsigned a, unsigned s) { unsigned r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ unsigned pk_sub_s(unsigned a, unsigned s) { unsigned r; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
__device__ __forceinline__ unsigned bfi(unsigned m, unsigned a, unsigned b) { unsigned r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(a), "v"(b)); return r; }
// signed 16-bit compares of one half of two packed registers -> lane mask
__device__ __forceinline__ unsigned long long cmp_gt_lo(unsigned a, unsigned b) { unsigned long long m; asm("v_cmp_gt_i16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:WORD_0" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cmp_gt_hi(unsigned a, unsigned b) { unsigned long long m; asm("v_cmp_gt_i16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:WORD_1" : "=s"(m) : "v"(a), "v"(b)); return m; }
__device__ __forceinline__ unsigned long long cmp_ne0_lo(unsigned a) { unsigned long long m; asm("v_cmp_ne_u16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(m) : "v"(a), "v"(0)); return m; }
__device__ __forceinline__ unsigned long long cmp_ne0_hi(unsigned a) { unsigned long long m; asm("v_cmp_ne_u16_sdwa %0, %1, %2 src0_sel:WORD_1 src1_sel:DWORD" : "=s"(m) : "v"(a), "v"(0)); return m; }

__device__ __forceinline__ void sstore4(unsigned long long m0, unsigned long long m1, unsigned long long base, unsigned off)
{
	u4 d = {(unsigned)m0, (unsigned)(m0 >> 32), (unsigned)m1, (unsigned)(m1 >> 32)};
	asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(d), "s"(base), "s"(off) : "memory");
}

struct Res { long long cycles; };

template <int V>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8))) void k_step(Res *out, int iters, unsigned *gbuf, size_t wave_bytes, int *sink)
{
	extern __shared__ unsigned lds[];
	const int lane = lane_id();
	const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
	for (int i = threadIdx.x; i < 2048 + 64; i += blockDim.x) lds[i] = (unsigned)(((i * 2654435761u) >> 13) & 3) + 1 << 24 | 0x000c0c0cu | (((i * 40503u) >> 7) & 3) + 1 << 8;
	__syncthreads();
	unsigned *row = gbuf + (size_t)wave * (wave_bytes / 4);
	const unsigned long long sbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)row)) |
	                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)row >> 32)) << 32);
	unsigned soff = 0;
	const unsigned *qptr = lds + 2048 + lane;
	int XA = 0, VA = lane << 20, UA = 0, YA = 0, ZA = 10 << 24, HA = -lane, HA1 = -lane * 2;
	unsigned T0 = 0x0b080808u + lane, T1 = 0x080b0808u;
	unsigned accA = 0, accB = 0;
	unsigned long long geLoM = ~0ull << 3, spM = 1ull << 54;
	int thrI = 1 << 20, thrI1 = 1 << 20, zd = 400;
	const int q24 = 4 << 24; const unsigned M24 = 11u << 24;
	const unsigned qP = 0x04000400u, MP = 0x0b000b00u;
	const long long t0 = clock64();
	for (int r = 0; r < iters; ++r) {
		const unsigned sel = *qptr; qptr -= 1;
		if (V <= 1) {
			const int xp = dppz_shr1(XA), vp = dppz_shr1(VA), Hp = dppz_shr1(HA);
			const int zn = (int)__builtin_amdgcn_perm(T0, T1, sel);
			ZA = lane_in(geLoM) ? zn : ZA;
			const int a = xp + vp, b = YA + UA;
			const unsigned long long c1 = ballot(a > ZA);
			unsigned zz = (unsigned)ZA > (unsigned)a ? (unsigned)ZA : (unsigned)a;
			const unsigned long long c2 = ballot(b > (int)zz);
			zz = zz > (unsigned)b ? zz : (unsigned)b;
			zz = zz < M24 ? zz : M24;
			const int un = (int)zz - vp, vn = (int)zz - UA;
			const int zq = (int)zz - q24;
			const int a2 = a - zq, b2 = b - zq;
			const unsigned long long c3 = ballot(a2 > 0), c4 = ballot(b2 > 0);
			XA = a2 > 0 ? a2 : 0; YA = b2 > 0 ? b2 : 0; UA = un; VA = vn;
			if (V == 0) {
				accA = shl1_in(accA, c1); accA = shl1_in(accA, c2); accA = shl1_in(accA, c3); accA = shl1_in(accA, c4);
				if ((r & 7) == 7) row[(r >> 3) * 64 + lane] = accA;
			} else {
				sstore4(c1, c2, sbase, soff); sstore4(c3, c4, sbase, soff + 16); soff += 32;
			}
			const bool sp = lane_in(spM);
			const int h = (sp ? Hp : HA) + (int)((unsigned)(sp ? un : vn) >> 24);
			HA = lane_in(geLoM) ? h : HA;
			if (ballot(HA > thrI)) { thrI = __builtin_amdgcn_readlane(HA, 5); }
			else if (!(ballot(HA >= thrI - zd) & geLoM)) { thrI -= 1; }
			thrI += 5;
		} else {
			// packed: XA VA UA YA ZA hold (v0 << 8 | v1 << 24)
			const unsigned xp = (unsigned)dppz_shr1(XA), vp = (unsigned)dppz_shr1(VA);
			const unsigned zn = __builtin_amdgcn_perm(T0, T1, sel);
			ZA = lane_in(geLoM) ? (int)zn : ZA;
			const unsigned z = (unsigned)ZA;
			const unsigned a = pk_add(xp, vp), b = pk_add((unsigned)YA, (unsigned)UA);
			const unsigned zz1 = pk_max_u(z, a);
			unsigned zz = pk_max_u(zz1, b);
			zz = pk_min_us(zz, MP);
			const unsigned un = pk_sub(zz, vp), vn = pk_sub(zz, (unsigned)UA);
			const unsigned zq = pk_sub_s(zz, qP);
			const unsigned a2 = pk_sub(a, zq), b2 = pk_sub(b, zq);
			const unsigned xn = pk_max_i(a2, 0u), yn = pk_max_i(b2, 0u);
			if (V == 6) {
				// sign bits of saturating differences, gathered into a nibble per half at bits 15..12, shifted into the accumulator
				const unsigned s1 = pk_sub_sat(z, a), s2 = pk_sub_sat(zz1, b);
				const unsigned s3 = pk_add(xn, 0x7fff7fffu), s4 = pk_add(yn, 0x7fff7fffu);
				unsigned n = bfi(0x80008000u, s1, pk_lshr(s2, 0x00010001u));
				unsigned m = bfi(0x80008000u, s3, pk_lshr(s4, 0x00010001u));
				n = bfi(0xc000c000u, n, pk_lshr(m, 0x00020002u));
				accA = bfi(0xf000f000u, n, pk_lshr(accA, 0x00040004u));
				if ((r & 3) == 3) row[(r >> 2) * 64 + lane] = accA;
			} else {
				const unsigned long long c1l = cmp_gt_lo(a, z), c1h = cmp_gt_hi(a, z);
				const unsigned long long c2l = cmp_gt_lo(b, zz1), c2h = cmp_gt_hi(b, zz1);
				const unsigned long long c3l = cmp_ne0_lo(xn), c3h = cmp_ne0_hi(xn);
				const unsigned long long c4l = cmp_ne0_lo(yn), c4h = cmp_ne0_hi(yn);
				if (V == 2 || V == 4) {
					accA = shl1_in(accA, c1l); accA = shl1_in(accA, c2l); accA = shl1_in(accA, c3l); accA = shl1_in(accA, c4l);
					accB = shl1_in(accB, c1h); accB = shl1_in(accB, c2h); accB = shl1_in(accB, c3h); accB = shl1_in(accB, c4h);
					if ((r & 7) == 7) { row[(r >> 3) * 128 + lane] = accA; row[(r >> 3) * 128 + 64 + lane] = accB; }
				} else {
					sstore4(c1l, c2l, sbase, soff); sstore4(c3l, c4l, sbase, soff + 16);
					sstore4(c1h, c2h, sbase, soff + 32); sstore4(c3h, c4h, sbase, soff + 48); soff += 64;
				}
			}
			XA = (int)xn; YA = (int)yn; UA = (int)un; VA = (int)vn;
			const bool sp = lane_in(spM);
			const unsigned uv = sp ? un : vn;
			if (V == 2 || V == 3) {
				const int Hp0 = dppz_shr1(HA), Hp1 = dppz_shr1(HA1);
				const int h0 = (sp ? Hp0 : HA) + (int)((uv >> 8) & 0xffu);
				const int h1 = (sp ? Hp1 : HA1) + (int)(uv >> 24);
				HA = lane_in(geLoM) ? h0 : HA; HA1 = lane_in(geLoM) ? h1 : HA1;
				if (ballot(HA > thrI)) { thrI = __builtin_amdgcn_readlane(HA, 5); }
				else if (!(ballot(HA >= thrI - zd) & geLoM)) { thrI -= 1; }
				if (ballot(HA1 > thrI1)) { thrI1 = __builtin_amdgcn_readlane(HA1, 5); }
				else if (!(ballot(HA1 >= thrI1 - zd) & geLoM)) { thrI1 -= 1; }
				thrI += 5; thrI1 += 5;
			} else {
				const unsigned Gp = (unsigned)dppz_shr1(HA);
				const unsigned g = pk_add(sp ? Gp : (unsigned)HA, pk_lshr(uv, 0x00080008u));
				HA = lane_in(geLoM) ? (int)g : HA;
				// thresholds as packed halves in SGPRs: lo compare by SDWA, hi compare as a 32-bit compare against thr << 16 | 0xffff
				unsigned long long il, ih;
				asm("v_cmp_gt_i16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(il) : "v"(HA), "s"(thrI));
				asm("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(ih) : "v"(HA), "s"(thrI1));
				if (il) { thrI = __builtin_amdgcn_readlane(HA, 5) & 0x7fff; }
				else {
					unsigned long long zl;
					asm("v_cmp_ge_i16_sdwa %0, %1, %2 src0_sel:WORD_0 src1_sel:DWORD" : "=s"(zl) : "v"(HA), "s"(thrI - zd));
					if (!(zl & geLoM)) thrI -= 1;
				}
				if (ih) { thrI1 = __builtin_amdgcn_readlane(HA, 7) | 0xffff; }
				else {
					unsigned long long zh;
					asm("v_cmp_ge_i32_e64 %0, %1, %2" : "=s"(zh) : "v"(HA), "s"(thrI1 - (zd << 16)));
					if (!(zh & geLoM)) thrI1 -= 1;
				}
				thrI += 5; thrI1 += 5 << 16;
			}
		}
		if (r & 1) geLoM = (geLoM << 1) | (geLoM >> 63); else spM = (spM << 1) | (spM >> 63);
		if ((r & 31) == 31) { qptr += 32; soff = 0; }
		if ((r & 1023) == 1023) qptr = lds + 2048 + lane;
	}
	if (V == 1 || V == 3 || V == 5) asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
	const long long t1 = clock64();
	if (lane == 0) out[wave].cycles = t1 - t0;
	if (XA + VA + UA + YA + ZA + HA + HA1 + (int)accA + (int)accB + thrI + thrI1 == 0x7fffffff) *sink = 1;
}

// ---- single-op issue rates (8 chains, as tools/ubench_issue.hip) ----
#define REP8(X) X X X X X X X X
template <int KIND>
__global__ void k_op(Res *out, int iters, int *sink)
{
	const int lane = threadIdx.x & 63;
	int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
	int s0 = 1;
	unsigned long long m = 0x00ff00ff00ff00ffull;
	const long long t0 = clock64();
	for (int it = 0; it < iters; ++it) {
#define OP8(T) asm volatile(REP8(T(%0,%1) T(%1,%2) T(%2,%3) T(%3,%4) T(%4,%5) T(%5,%6) T(%6,%7) T(%7,%0)) \
	: "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0), "s"(m) : "vcc")
		if (KIND == 0) {
#define T0_(a, b) "v_cndmask_b32_e64 " #a ", " #a ", " #b ", %9\n"
			OP8(T0_);
		} else if (KIND == 1) {
#define T1_(a, b) "v_cmp_gt_i16_sdwa vcc, " #a ", " #b " src0_sel:WORD_1 src1_sel:WORD_1\n"
			OP8(T1_);
		} else if (KIND == 2) {
#define T2_(a, b) "v_add_u32_sdwa " #a ", " #a ", " #b " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
			OP8(T2_);
		} else if (KIND == 3) {
#define T3_(a, b) "v_lshrrev_b32 " #a ", 3, " #b "\n"
			OP8(T3_);
		} else if (KIND == 4) {
#define T4_(a, b) "v_or_b32 " #a ", " #a ", " #b "\n"
			OP8(T4_);
		} else if (KIND == 5) {
#define T5_(a, b) "v_bfi_b32 " #a ", %8, " #a ", " #b "\n"
			OP8(T5_);
		} else if (KIND == 6) {
#define T6_(a, b) "v_pk_sub_i16 " #a ", " #a ", " #b " clamp\n"
			OP8(T6_);
		} else if (KIND == 7) {
#define T7_(a, b) "v_pk_lshrrev_b16 " #a ", 1, " #b "\n"
			OP8(T7_);
		} else if (KIND == 8) {
#define T8_(a, b) "v_max_i32 " #a ", 0, " #b "\n"
			OP8(T8_);
		} else if (KIND == 9) {
#define T9_(a, b) "v_sub_co_u32 " #a ", vcc, " #a ", " #b "\n"
			OP8(T9_);
		} else if (KIND == 10) {
#define T10_(a, b) "v_mov_b32_dpp " #a ", " #b " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
			OP8(T10_);
		} else if (KIND == 11) {
#define T11_(a, b) "v_and_or_b32 " #a ", " #a ", " #b ", " #b "\n"
			OP8(T11_);
		} else if (KIND == 12) {
#define T12_(a, b) "v_cmp_gt_i32_e64 s[20:21], " #a ", " #b "\n"
			asm volatile(REP8(T12_(%0,%1) T12_(%1,%2) T12_(%2,%3) T12_(%3,%4) T12_(%4,%5) T12_(%5,%6) T12_(%6,%7) T12_(%7,%0))
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) :: "s20", "s21");
		} else if (KIND == 13) {
#define T13_(a, b) "v_sub_u32 " #a ", " #a ", " #b "\n v_max_u32 " #b ", " #b ", " #a "\n"
			asm volatile(REP8(T13_(%0,%1) T13_(%2,%3) T13_(%4,%5) T13_(%6,%7))
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 14) {
#define T14_(a, b) "v_pk_max_u16 " #a ", " #a ", " #b "\n"
			OP8(T14_);
		} else if (KIND == 15) {
#define T15_(a, b) "v_lshlrev_b32 " #a ", 1, " #b "\n"
			OP8(T15_);
		} else if (KIND == 16) {
#define T16_(a, b) "v_ashrrev_i32 " #a ", 31, " #b "\n"
			OP8(T16_);
		} else if (KIND == 17) {
#define T17_(a, b) "v_and_b32 " #a ", " #a ", " #b "\n"
			OP8(T17_);
		}
	}
	const long long t1 = clock64();
	if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)].cycles = t1 - t0;
	if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 == 0x7fffffff) *sink = 1;
}


// ---- are scalar stores safe when the data SGPRs are overwritten right behind them?  every iteration writes two fresh compare
// masks into the SAME four SGPRs and stores them with no wait; the host recomputes every mask ----
__global__ __launch_bounds__(512) void k_sstore_check(unsigned *gbuf, size_t wave_bytes, int iters)
{
	const int lane = lane_id();
	const int wave = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
	unsigned *row = gbuf + (size_t)wave * (wave_bytes / 4);
	const unsigned long long sbase = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)row)) |
	                                 ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)row >> 32)) << 32);
	unsigned soff = 0;
	for (int r = 0; r < iters; ++r) {
		const unsigned x = ((unsigned)lane * 2654435761u + (unsigned)r * 40503u + (unsigned)wave * 977u) >> 7;
		const unsigned y = ((unsigned)lane * 40503u + (unsigned)r * 2654435761u) >> 7;
		asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %0, %2\n s_nop 1\n s_store_dwordx4 s[20:23], %3, %4"
		             :: "v"(x), "v"(y), "v"(y ^ 0x5555u), "s"(sbase), "s"(soff) : "s20", "s21", "s22", "s23", "memory");
		soff += 16;
	}
	asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
}

static void run_check(int cus)
{
	const int iters = 2048, W = 8;
	const size_t wave_bytes = (size_t)iters * 16;
	const size_t nw = (size_t)cus * 4 * W;
	unsigned *gbuf;
	CHECK(hipMalloc(&gbuf, nw * wave_bytes));
	CHECK(hipMemset(gbuf, 0xee, nw * wave_bytes));
	hipLaunchKernelGGL(k_sstore_check, dim3(cus * 4), dim3(64 * W), 0, 0, gbuf, wave_bytes, iters);
	CHECK(hipDeviceSynchronize());
	std::vector<unsigned> h(nw * wave_bytes / 4);
	CHECK(hipMemcpy(h.data(), gbuf, nw * wave_bytes, hipMemcpyDeviceToHost));
	size_t bad = 0, first = (size_t)-1;
	for (size_t w = 0; w < nw; ++w)
		for (int r = 0; r < iters; ++r) {
			unsigned long long m0 = 0, m1 = 0;
			for (int lane = 0; lane < 64; ++lane) {
				const unsigned x = ((unsigned)lane * 2654435761u + (unsigned)r * 40503u + (unsigned)w * 977u) >> 7;
				const unsigned y = ((unsigned)lane * 40503u + (unsigned)r * 2654435761u) >> 7;
				if (x > y) m0 |= 1ull << lane;
				if (x < (y ^ 0x5555u)) m1 |= 1ull << lane;
			}
			const unsigned *p = &h[(w * iters + r) * 4];
			if (p[0] != (unsigned)m0 || p[1] != (unsigned)(m0 >> 32) || p[2] != (unsigned)m1 || p[3] != (unsigned)(m1 >> 32)) { if (!bad) first = w * iters + r; ++bad; }
		}
	printf("scalar-store check: %zu waves x %d stores of 16 bytes, %zu wrong (first at %zd)\n", nw, iters, bad, (ssize_t)first);
	CHECK(hipFree(gbuf));
}

static int g_khz = 2400000;

template <int V>
static void run_step(const char *name, int cus, int aln_per_wave)
{
	const int iters = 2048, W = 8;                         // W waves per workgroup, 4 workgroups per CU = 8 waves per SIMD
	const size_t wave_bytes = 128 * 1024;
	const size_t nw = (size_t)cus * 4 * W;
	Res *d; int *sink; unsigned *gbuf;
	CHECK(hipMalloc(&d, sizeof(Res) * nw));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipMalloc(&gbuf, nw * wave_bytes));
	hipLaunchKernelGGL(k_step<V>, dim3(cus * 4), dim3(64 * W), 16 * 1024, 0, d, 64, gbuf, wave_bytes, sink);
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL(k_step<V>, dim3(cus * 4), dim3(64 * W), 16 * 1024, 0, d, iters, gbuf, wave_bytes, sink);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<Res> h(nw);
	CHECK(hipMemcpy(h.data(), d, sizeof(Res) * nw, hipMemcpyDeviceToHost));
	double avg = 0; for (auto &r : h) avg += (double)r.cycles; avg /= (double)nw;
	// chip-wide cost of one (alignment, diagonal): wall time / (waves * alignments per wave * diagonals)
	const double ns = (double)ms * 1e6 / ((double)nw * aln_per_wave * iters);
	printf("%-46s %8.3f ms  %7.1f wave-cycles/diag  %8.4f ns per (alignment, diagonal) chip-wide  [x 573 diag x 200k aln = %6.2f ms]\n",
	       name, ms, avg / iters, ns, ns * 573 * 200000 * 1e-6);
	CHECK(hipFree(d)); CHECK(hipFree(sink)); CHECK(hipFree(gbuf));
}

template <int KIND>
static void run_op(const char *name, int cus, int per_iter = 64)
{
	const int iters = 2000;
	Res *d; int *sink;
	CHECK(hipMalloc(&d, sizeof(Res) * cus * 32));
	CHECK(hipMalloc(&sink, 4));
	printf("%-34s", name);
	const int ws[] = {8, 32};
	for (int w : ws) {
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		const int nb = w > 16 ? 2 : 1;
		const int tpb = 64 * (w > 16 ? 16 : w);
		hipLaunchKernelGGL(k_op<KIND>, dim3(cus * nb), dim3(tpb), 0, 0, d, 10, sink);
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_op<KIND>, dim3(cus * nb), dim3(tpb), 0, 0, d, iters, sink);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		const double wall = (double)w * iters * per_iter / ((double)ms * 1e-3 * (double)g_khz * 1e3);
		printf(" | W=%2d %.2f ms, wall %.2f instr/cyc/CU", w, ms, wall);
	}
	printf("\n");
	CHECK(hipFree(d)); CHECK(hipFree(sink));
}

int main()
{
	hipDeviceProp_t pr;
	CHECK(hipGetDeviceProperties(&pr, 0));
	const int cus = pr.multiProcessorCount;
	g_khz = pr.clockRate;
	printf("%s, %d CUs, clock %d kHz\n", pr.gcnArchName, cus, pr.clockRate);
	run_check(cus);
	run_step<0>("V0 single, top byte, addc traceback", cus, 1);
	run_step<1>("V1 single, top byte, s_store traceback", cus, 1);
	run_step<2>("V2 pair pk16, sdwa cmp + addc, H 32-bit", cus, 2);
	run_step<3>("V3 pair pk16, sdwa cmp + s_store, H 32-bit", cus, 2);
	run_step<4>("V4 pair pk16, sdwa cmp + addc, H packed", cus, 2);
	run_step<5>("V5 pair pk16, sdwa cmp + s_store, H packed", cus, 2);
	run_step<6>("V6 pair pk16, sign-gather traceback, H packed", cus, 2);
	run_op<0>("v_cndmask_b32 sgpr mask", cus);
	run_op<1>("v_cmp_gt_i16_sdwa", cus);
	run_op<2>("v_add_u32_sdwa byte3", cus);
	run_op<3>("v_lshrrev_b32", cus);
	run_op<15>("v_lshlrev_b32", cus);
	run_op<16>("v_ashrrev_i32", cus);
	run_op<4>("v_or_b32 v,v", cus);
	run_op<17>("v_and_b32 v,v", cus);
	run_op<5>("v_bfi_b32 s,v,v", cus);
	run_op<11>("v_and_or_b32", cus);
	run_op<6>("v_pk_sub_i16 clamp", cus);
	run_op<7>("v_pk_lshrrev_b16", cus);
	run_op<14>("v_pk_max_u16", cus);
	run_op<8>("v_max_i32 0,v", cus);
	run_op<9>("v_sub_co_u32", cus);
	run_op<10>("v_mov_b32_dpp wave_shr:1", cus);
	run_op<12>("v_cmp_gt_i32 -> sgpr pair", cus);
	run_op<13>("v_sub_u32 / v_max_u32 alternating", cus);
	return 0;
}
