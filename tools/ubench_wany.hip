#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ unsigned fsh(unsigned hi, unsigned lo, unsigned sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
__device__ __forceinline__ bool window_any(unsigned w0, unsigned w1, unsigned qh)
{
	unsigned x[16];
	x[0] = w0 ^ qh;
#pragma unroll
	for (int j = 1; j < 16; ++j) x[j] = fsh(w1, w0, 2u * j) ^ qh;
	const unsigned m0 = min(min(x[0], x[1]), min(x[2], x[3])), m1 = min(min(x[4], x[5]), min(x[6], x[7]));
	const unsigned m2 = min(min(x[8], x[9]), min(x[10], x[11])), m3 = min(min(x[12], x[13]), min(x[14], x[15]));
	return min(min(m0, m1), min(m2, m3)) == 0;
}
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmin(unsigned a, unsigned b)
{
	u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
	u16x2 r = __builtin_elementwise_min(x, y);
	return __builtin_bit_cast(unsigned, r);
}
// the 16 windows as four byte-aligned groups: V >> 2r holds windows r, r+4, r+8, r+12 at byte offsets 0..3
__device__ __forceinline__ bool window_any_qsad(unsigned w0, unsigned w1, unsigned qh)
{
	const unsigned long long V = ((unsigned long long)w1 << 32) | w0;
	unsigned long long s0 = __builtin_amdgcn_qsad_pk_u16_u8(V, qh, 0ull);
	unsigned long long s1 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 2, qh, 0ull);
	unsigned long long s2 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 4, qh, 0ull);
	unsigned long long s3 = __builtin_amdgcn_qsad_pk_u16_u8(V >> 6, qh, 0ull);
	unsigned a = pkmin((unsigned)s0, (unsigned)(s0 >> 32)), b = pkmin((unsigned)s1, (unsigned)(s1 >> 32));
	unsigned c = pkmin((unsigned)s2, (unsigned)(s2 >> 32)), d = pkmin((unsigned)s3, (unsigned)(s3 >> 32));
	unsigned m = pkmin(pkmin(a, b), pkmin(c, d));
	return (m & 0xffffu) == 0 || (m >> 16) == 0;
}
template <int V> __global__ void k(const unsigned *in, unsigned *out, int iters)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned w0 = in[3 * t], w1 = in[3 * t + 1], qh = in[3 * t + 2];
	unsigned acc = 0;
	for (int i = 0; i < iters; ++i) {
		const bool r = V ? window_any_qsad(w0, w1, qh) : window_any(w0, w1, qh);
		acc += r;
		w0 = w0 * 1664525u + 1013904223u + acc; w1 ^= w0 >> 3;
		if ((i & 7) == 0) qh = fsh(w1, w0, 2u * ((w0 >> 20) & 15));     // plant a match now and then
	}
	out[t] = acc;
}
int main()
{
	const int N = 256 * 64 * 32, iters = 2000;
	unsigned *h = (unsigned *)malloc(12 * N), *d, *o0, *o1;
	srand(1);
	for (int i = 0; i < 3 * N; ++i) h[i] = (unsigned)rand() * 2654435761u;
	hipMalloc(&d, 12 * N); hipMalloc(&o0, 4 * N); hipMalloc(&o1, 4 * N);
	hipMemcpy(d, h, 12 * N, hipMemcpyHostToDevice);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int v = 0; v < 2; ++v) for (int rep = 0; rep < 2; ++rep) {
		hipEventRecord(e0);
		if (v) hipLaunchKernelGGL(k<1>, dim3(N / 64), dim3(64), 0, 0, d, o1, iters);
		else hipLaunchKernelGGL(k<0>, dim3(N / 64), dim3(64), 0, 0, d, o0, iters);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		printf("variant %d: %.3f ms (%.1f ps per window_any per lane)\n", v, ms, ms * 1e9 / ((double)N * iters));
	}
	unsigned *a = (unsigned *)malloc(4 * N), *b = (unsigned *)malloc(4 * N);
	hipMemcpy(a, o0, 4 * N, hipMemcpyDeviceToHost); hipMemcpy(b, o1, 4 * N, hipMemcpyDeviceToHost);
	long long diff = 0, sum = 0;
	for (int i = 0; i < N; ++i) { diff += a[i] != b[i]; sum += a[i]; }
	printf("differences %lld, matches %lld\n", diff, sum);
	return 0;
}
