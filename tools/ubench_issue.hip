// ubench_issue.hip -- instruction-issue microbenchmarks for gfx950 (diagnostics, not part of the product).
//
// The assembly and ksw2 kernels are instruction-issue / latency bound, so the numbers that price a design are:
// how many VALU / SALU / mixed wave-instructions a CU issues per cycle at 1..8 waves per SIMD, what a dependent
// VALU -> SGPR -> VALU hop costs (v_readlane, ballot), and the LDS round trip under load.
// One workgroup of 64*W threads per CU (96 KB of dynamic LDS forces one workgroup per CU), every wave runs the
// same straight-line block `iters` times and stamps s_memtime around it.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/ubench_issue tools/ubench_issue.hip && gpurun_out/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Res { long long cycles; };

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ void k_bench(Res *out, int iters, int *sink)
{
	extern __shared__ int lds[];
	const int lane = threadIdx.x & 63;
	int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
	int s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
	double d0 = lane, d1 = lane + 1, d2 = lane + 2, d3 = lane + 3;          // 64-bit register pairs for v_pk_fma_f32
	for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 4 + 64) & 16383;   // pointer chase table (byte offsets)
	__syncthreads();
	const long long t0 = clock64();
	for (int it = 0; it < iters; ++it) {
		if (KIND == 0) {          // 64 independent-ish VALU (8 chains)
			asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
			                  "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0));
		} else if (KIND == 1) {   // 64 SALU (8 chains)
			asm volatile(REP8("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
			                  "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n")
			             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) :: "scc");
		} else if (KIND == 2) {   // 32 VALU + 32 SALU interleaved
			asm volatile(REP8("v_add_u32 %0, %0, 1\n s_add_u32 %4, %4, 1\n v_add_u32 %1, %1, 1\n s_add_u32 %5, %5, 1\n"
			                  "v_add_u32 %2, %2, 1\n s_add_u32 %6, %6, 1\n v_add_u32 %3, %3, 1\n s_add_u32 %7, %7, 1\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
		} else if (KIND == 3) {   // 64 dependent VALU (one chain)
			asm volatile(REP64("v_add_u32 %0, %0, 1\n") : "+v"(v0));
		} else if (KIND == 4) {   // 64 VALU with DPP (8 chains)
			asm volatile(REP8("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                  "v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                  "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
			                  "v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 5) {   // 32 x (v_readlane -> v_add with the SGPR): dependent VALU -> SGPR -> VALU hops (64 instr)
			asm volatile(REP8("v_readlane_b32 %1, %0, 3\n v_add_u32 %0, %0, %1\n v_readlane_b32 %1, %0, 5\n v_add_u32 %0, %0, %1\n"
			                  "v_readlane_b32 %1, %0, 7\n v_add_u32 %0, %0, %1\n v_readlane_b32 %1, %0, 9\n v_add_u32 %0, %0, %1\n")
			             : "+v"(v0), "+s"(s0));
		} else if (KIND == 6) {   // 16 x (v_cmp -> s_and -> v_cndmask -> v_add): ballot-style hop (64 instr)
			asm volatile(REP8("v_cmp_lt_u32 vcc, %1, %0\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %0, 1\n"
			                  "v_cmp_lt_u32 vcc, %1, %0\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %0, %0, 1\n")
			             : "+v"(v0) : "v"(v1) : "vcc");
		} else if (KIND == 7) {   // 16 dependent ds_read_b32 (pointer chase): LDS latency
			asm volatile(REP8("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v0) :: "memory");
		} else if (KIND == 8) {   // 64 v_pk_add_u16 (8 chains)
			asm volatile(REP8("v_pk_add_u16 %0, %0, %0\n v_pk_add_u16 %1, %1, %1\n v_pk_add_u16 %2, %2, %2\n v_pk_add_u16 %3, %3, %3\n"
			                  "v_pk_add_u16 %4, %4, %4\n v_pk_add_u16 %5, %5, %5\n v_pk_add_u16 %6, %6, %6\n v_pk_add_u16 %7, %7, %7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 9) {   // 64 v_pk_max_i16 / v_pk_min_u16 mix (8 chains)
			asm volatile(REP8("v_pk_max_i16 %0, %0, %1\n v_pk_min_u16 %1, %1, %2\n v_pk_max_i16 %2, %2, %3\n v_pk_min_u16 %3, %3, %4\n"
			                  "v_pk_max_i16 %4, %4, %5\n v_pk_min_u16 %5, %5, %6\n v_pk_max_i16 %6, %6, %7\n v_pk_min_u16 %7, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 10) {  // 16 independent ds_read_b128 then one wait: LDS throughput (16 instr)
			int4 a;
			asm volatile(REP8("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n") "s_waitcnt lgkmcnt(0)\n" : "=&v"(a) : "v"((lane * 16) & 16383) : "memory");
			v0 += a.x;
		} else if (KIND == 11) {  // 16 dependent ds_bpermute: crossbar latency
			asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(v0) : "v"(v1 * 4) : "memory");
		} else if (KIND == 12) {  // v_perm_b32 x 64 (8 chains)
			asm volatile(REP8("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n"
			                  "v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(0x03020100 + s0));
		} else if (KIND == 14) {  // control: 64 v_fma_f32 (8 chains)
			asm volatile(REP8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
			                  "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 15) {  // control: 64 v_pk_fma_f32 on register pairs (4 chains of 64-bit operands)
			asm volatile(REP8("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
			                  "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
			             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
		} else if (KIND == 16) {  // the ops of this library's inner loops: v_alignbit_b32, v_xor_b32, v_min3_u32, v_cndmask (8 chains)
			asm volatile(REP8("v_alignbit_b32 %0, %1, %0, 7\n v_xor_b32 %1, %1, %2\n v_min3_u32 %2, %2, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
			                  "v_alignbit_b32 %4, %5, %4, 9\n v_xor_b32 %5, %5, %6\n v_min3_u32 %6, %6, %7, %0\n v_cndmask_b32 %7, %7, %0, vcc\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) :: "vcc");
		} else if (KIND == 17) {  // 64 v_mov_b32 (8 chains)
			asm volatile(REP8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
			                  "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 18) {  // v_xor_b32 with two DIFFERENT vector sources (8 chains)
			asm volatile(REP8("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %4\n"
			                  "v_xor_b32 %4, %4, %5\n v_xor_b32 %5, %5, %6\n v_xor_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 19) {  // control: v_fma_f32 with three DIFFERENT vector sources (8 chains)
			asm volatile(REP8("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %1, %2, %3, %4\n v_fma_f32 %2, %3, %4, %5\n v_fma_f32 %3, %4, %5, %6\n"
			                  "v_fma_f32 %4, %5, %6, %7\n v_fma_f32 %5, %6, %7, %0\n v_fma_f32 %6, %7, %0, %1\n v_fma_f32 %7, %0, %1, %2\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 20) {  // v_xor_b32 with ONE vector source and an inline constant (8 chains)
			asm volatile(REP8("v_xor_b32 %0, 5, %0\n v_xor_b32 %1, 5, %1\n v_xor_b32 %2, 5, %2\n v_xor_b32 %3, 5, %3\n"
			                  "v_xor_b32 %4, 5, %4\n v_xor_b32 %5, 5, %5\n v_xor_b32 %6, 5, %6\n v_xor_b32 %7, 5, %7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
		} else if (KIND == 21) {
			asm volatile(REP8("v_alignbit_b32 %0, %1, %0, 7\n"
			                  "v_alignbit_b32 %1, %2, %1, 7\n"
			                  "v_alignbit_b32 %2, %3, %2, 7\n"
			                  "v_alignbit_b32 %3, %4, %3, 7\n"
			                  "v_alignbit_b32 %4, %5, %4, 7\n"
			                  "v_alignbit_b32 %5, %6, %5, 7\n"
			                  "v_alignbit_b32 %6, %7, %6, 7\n"
			                  "v_alignbit_b32 %7, %0, %7, 7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 22) {
			asm volatile(REP8("v_min3_u32 %0, %0, %1, %2\n"
			                  "v_min3_u32 %1, %1, %2, %3\n"
			                  "v_min3_u32 %2, %2, %3, %4\n"
			                  "v_min3_u32 %3, %3, %4, %5\n"
			                  "v_min3_u32 %4, %4, %5, %6\n"
			                  "v_min3_u32 %5, %5, %6, %7\n"
			                  "v_min3_u32 %6, %6, %7, %0\n"
			                  "v_min3_u32 %7, %7, %0, %1\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 23) {
			asm volatile(REP8("v_cndmask_b32 %0, %0, %1, vcc\n"
			                  "v_cndmask_b32 %1, %1, %2, vcc\n"
			                  "v_cndmask_b32 %2, %2, %3, vcc\n"
			                  "v_cndmask_b32 %3, %3, %4, vcc\n"
			                  "v_cndmask_b32 %4, %4, %5, vcc\n"
			                  "v_cndmask_b32 %5, %5, %6, vcc\n"
			                  "v_cndmask_b32 %6, %6, %7, vcc\n"
			                  "v_cndmask_b32 %7, %7, %0, vcc\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "vcc");
		} else if (KIND == 24) {
			asm volatile(REP8("v_max_u32 %0, %0, %1\n"
			                  "v_max_u32 %1, %1, %2\n"
			                  "v_max_u32 %2, %2, %3\n"
			                  "v_max_u32 %3, %3, %4\n"
			                  "v_max_u32 %4, %4, %5\n"
			                  "v_max_u32 %5, %5, %6\n"
			                  "v_max_u32 %6, %6, %7\n"
			                  "v_max_u32 %7, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 25) {
			asm volatile(REP8("v_cmp_lt_u32 vcc, %0, %1\n"
			                  "v_cmp_lt_u32 vcc, %1, %2\n"
			                  "v_cmp_lt_u32 vcc, %2, %3\n"
			                  "v_cmp_lt_u32 vcc, %3, %4\n"
			                  "v_cmp_lt_u32 vcc, %4, %5\n"
			                  "v_cmp_lt_u32 vcc, %5, %6\n"
			                  "v_cmp_lt_u32 vcc, %6, %7\n"
			                  "v_cmp_lt_u32 vcc, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "vcc");
		} else if (KIND == 26) {
			asm volatile(REP8("v_sub_u32 %0, %0, %1\n"
			                  "v_sub_u32 %1, %1, %2\n"
			                  "v_sub_u32 %2, %2, %3\n"
			                  "v_sub_u32 %3, %3, %4\n"
			                  "v_sub_u32 %4, %4, %5\n"
			                  "v_sub_u32 %5, %5, %6\n"
			                  "v_sub_u32 %6, %6, %7\n"
			                  "v_sub_u32 %7, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 27) {
			asm volatile(REP8("v_xor_b32 %0, %8, %0\n"
			                  "v_xor_b32 %1, %8, %1\n"
			                  "v_xor_b32 %2, %8, %2\n"
			                  "v_xor_b32 %3, %8, %3\n"
			                  "v_xor_b32 %4, %8, %4\n"
			                  "v_xor_b32 %5, %8, %5\n"
			                  "v_xor_b32 %6, %8, %6\n"
			                  "v_xor_b32 %7, %8, %7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 28) {
			asm volatile(REP8("v_lshl_add_u32 %0, %0, 3, %1\n"
			                  "v_lshl_add_u32 %1, %1, 3, %2\n"
			                  "v_lshl_add_u32 %2, %2, 3, %3\n"
			                  "v_lshl_add_u32 %3, %3, 3, %4\n"
			                  "v_lshl_add_u32 %4, %4, 3, %5\n"
			                  "v_lshl_add_u32 %5, %5, 3, %6\n"
			                  "v_lshl_add_u32 %6, %6, 3, %7\n"
			                  "v_lshl_add_u32 %7, %7, 3, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 29) {
			asm volatile(REP8("v_bfe_u32 %0, %0, 3, 9\n"
			                  "v_bfe_u32 %1, %1, 3, 9\n"
			                  "v_bfe_u32 %2, %2, 3, 9\n"
			                  "v_bfe_u32 %3, %3, 3, 9\n"
			                  "v_bfe_u32 %4, %4, 3, 9\n"
			                  "v_bfe_u32 %5, %5, 3, 9\n"
			                  "v_bfe_u32 %6, %6, 3, 9\n"
			                  "v_bfe_u32 %7, %7, 3, 9\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 30) {
			asm volatile(REP8("v_and_b32 %0, 0x55555555, %0\n"
			                  "v_and_b32 %1, 0x55555555, %1\n"
			                  "v_and_b32 %2, 0x55555555, %2\n"
			                  "v_and_b32 %3, 0x55555555, %3\n"
			                  "v_and_b32 %4, 0x55555555, %4\n"
			                  "v_and_b32 %5, 0x55555555, %5\n"
			                  "v_and_b32 %6, 0x55555555, %6\n"
			                  "v_and_b32 %7, 0x55555555, %7\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 31) {
			asm volatile(REP8("v_add_u32 %0, %0, %1\n"
			                  "v_add_u32 %1, %1, %2\n"
			                  "v_add_u32 %2, %2, %3\n"
			                  "v_add_u32 %3, %3, %4\n"
			                  "v_add_u32 %4, %4, %5\n"
			                  "v_add_u32 %5, %5, %6\n"
			                  "v_add_u32 %6, %6, %7\n"
			                  "v_add_u32 %7, %7, %0\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 32) {
			asm volatile(REP8("v_addc_co_u32 %0, vcc, %0, %0, vcc\n"
			                  "v_addc_co_u32 %1, vcc, %1, %1, vcc\n"
			                  "v_addc_co_u32 %2, vcc, %2, %2, vcc\n"
			                  "v_addc_co_u32 %3, vcc, %3, %3, vcc\n"
			                  "v_addc_co_u32 %4, vcc, %4, %4, vcc\n"
			                  "v_addc_co_u32 %5, vcc, %5, %5, vcc\n"
			                  "v_addc_co_u32 %6, vcc, %6, %6, vcc\n"
			                  "v_addc_co_u32 %7, vcc, %7, %7, vcc\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "vcc");
		} else if (KIND == 33) {
			asm volatile(REP8("v_readlane_b32 %8, %0, 5\n"
			                  "v_readlane_b32 %8, %1, 5\n"
			                  "v_readlane_b32 %8, %2, 5\n"
			                  "v_readlane_b32 %8, %3, 5\n"
			                  "v_readlane_b32 %8, %4, 5\n"
			                  "v_readlane_b32 %8, %5, 5\n"
			                  "v_readlane_b32 %8, %6, 5\n"
			                  "v_readlane_b32 %8, %7, 5\n")
			             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(s0) : "memory");
		} else if (KIND == 13) {  // 16 x (v_cmp_eq -> s_ff1 -> v_readlane(sgpr idx via m0-free form) ...): ballot + ctz + bcast hop
			asm volatile(REP8("v_cmp_ne_u32 vcc, %0, %2\n s_ff1_i32_b64 %1, vcc\n s_and_b32 %1, %1, 63\n v_readlane_b32 %1, %0, %1\n v_add_u32 %0, %0, %1\n"
			                  "v_cmp_ne_u32 vcc, %0, %2\n s_ff1_i32_b64 %1, vcc\n s_and_b32 %1, %1, 63\n")
			             : "+v"(v0), "+s"(s0) : "v"(v1) : "vcc", "scc");
		}
	}
	const long long t1 = clock64();
	if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)].cycles = t1 - t0;
	if (d0 + d1 + d2 + d3 == 12345.0) *sink = 2;
	if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 == 0x7fffffff) *sink = 1;
}

static int g_khz = 2400000;
template <int KIND>
static void run(const char *name, int instr_per_iter, int cus)
{
	const int iters = 2000;
	Res *d; int *sink;
	CHECK(hipMalloc(&d, sizeof(Res) * cus * 32));
	CHECK(hipMalloc(&sink, 4));
	CHECK(hipFuncSetAttribute((const void *)k_bench<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
	printf("%-34s", name);
	const int ws[] = {1, 2, 4, 8, 16, 32};
	for (int w : ws) {
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(k_bench<KIND>, dim3(cus), dim3(64 * (w > 16 ? 16 : w)), 96 * 1024, 0, d, 10, sink);   // warm
		const int nb = w > 16 ? 2 : 1;      // 32 waves per CU = two 16-wave workgroups (40 KB LDS each)
		const size_t ldsb = w > 16 ? 40 * 1024 : 96 * 1024;
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k_bench<KIND>, dim3(cus * nb), dim3(64 * (w > 16 ? 16 : w)), ldsb, 0, d, iters, sink);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<Res> h((size_t)cus * w);
		CHECK(hipMemcpy(h.data(), d, sizeof(Res) * h.size(), hipMemcpyDeviceToHost));
		double avg = 0; for (auto &r : h) avg += (double)r.cycles; avg /= (double)h.size();
		const double per_wave = avg / ((double)iters * instr_per_iter);          // cycles per instruction seen by one wave
		const double per_cu = (double)w / per_wave;                             // instructions per cycle per CU
		// wall clock: what the whole launch retired per cycle and CU (the per-wave clock64 column over-counts above one wave per SIMD)
		const double wall = (double)cus * w * iters * instr_per_iter / ((double)ms * 1e-3 * (double)g_khz * 1e3) / cus;
		printf(" | W=%2d %6.2f cyc/instr/wave %5.2f instr/cyc/CU (%.2f ms, wall %.2f instr/cyc/CU)", w, per_wave, per_cu, ms, wall);
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
	printf("%s, %d CUs, clock %d kHz; W = waves per CU (one workgroup per CU; 32 = two workgroups)\n", pr.gcnArchName, cus, pr.clockRate);
	run<0>("VALU v_add_u32 x64 (8 chains)", 64, cus);
	run<3>("VALU v_add_u32 x64 (1 dep chain)", 64, cus);
	run<1>("SALU s_add_u32 x64 (8 chains)", 64, cus);
	run<2>("VALU+SALU interleaved 32+32", 64, cus);
	run<4>("VALU DPP row_shr:1 x64", 64, cus);
	run<14>("control: v_fma_f32 x64", 64, cus);
	run<15>("control: v_pk_fma_f32 x64", 64, cus);
	run<19>("control: v_fma_f32 3 srcs x64", 64, cus);
	run<17>("v_mov_b32 x64", 64, cus);
	run<20>("v_xor_b32 v, const, v x64", 64, cus);
	run<18>("v_xor_b32 v, v, v' x64", 64, cus);
	run<16>("alignbit/xor/min3/cndmask x64", 64, cus);
	run<21>("v_alignbit_b32", 64, cus);
	run<22>("v_min3_u32", 64, cus);
	run<23>("v_cndmask_b32 (vcc)", 64, cus);
	run<24>("v_max_u32 v,v,v'", 64, cus);
	run<25>("v_cmp_lt_u32 -> vcc", 64, cus);
	run<26>("v_sub_u32 v,v,v'", 64, cus);
	run<27>("v_xor_b32 v, s, v", 64, cus);
	run<28>("v_lshl_add_u32", 64, cus);
	run<29>("v_bfe_u32", 64, cus);
	run<30>("v_and_b32 literal", 64, cus);
	run<31>("v_add_u32 v,v,v'", 64, cus);
	run<32>("v_addc_co_u32 (vcc in/out)", 64, cus);
	run<8>("v_pk_add_u16 x64", 64, cus);
	run<9>("v_pk_max_i16/min_u16 x64", 64, cus);
	run<12>("v_perm_b32 x64", 64, cus);
	run<5>("readlane->v_add dep hop x32 (64 i)", 64, cus);
	run<6>("v_cmp->s_and->cndmask->add x16", 64, cus);
	run<13>("cmp->ff1->readlane->add x8 (64 i)", 64, cus);
	run<7>("ds_read_b32 dependent x16", 16, cus);
	run<11>("ds_bpermute dependent x16", 16, cus);
	run<10>("ds_read_b128 x16 + wait", 16, cus);
	return 0;
}
