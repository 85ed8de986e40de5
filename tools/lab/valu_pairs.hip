// lab: issue cost of PAIRS of vector instructions (A, B alternating, 8 wavefronts per SIMD), gfx950: which kinds overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int OP> __device__ __forceinline__ void emit(float (&v)[16], double (&d)[8], int i, float sel) {
	if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
	if (OP == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(sel));
	if (OP == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 2) & 15]));
	if (OP == 3) asm volatile("v_min_f32_dpp %0, -%1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 2) & 15]));
	if (OP == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
	if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
	if (OP == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
	if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "s"(__builtin_amdgcn_read_exec()));
	if (OP == 8) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(v[(i + 1) & 15]) : "vcc");
	if (OP == 9) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i & 7]) : "v"(v[i]));
	if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
	if (OP == 11) asm volatile("s_nop 0");
	if (OP == 12) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
	if (OP == 13) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
}
static const char* kNames[] = {"min_f32", "med3_f32", "mov_dpp", "min_dpp", "xor", "fma_f32", "add_f64", "cndmask", "cmp", "cvt_f64", "mov", "s_nop", "add_u32", "fma_f64"};
constexpr int NOPS = 14;
template <int A, int B>
__global__ __launch_bounds__(256) void pair_kernel(float* out, int iters)
{
	float v[16]; double d[8];
#pragma unroll
	for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x * 16 + i) * 1.0001f;
#pragma unroll
	for (int i = 0; i < 8; ++i) d[i] = (double)(threadIdx.x + i) * 1.0000001;
	const float sel = (threadIdx.x & 1) ? __builtin_inff() : -__builtin_inff();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int rep = 0; rep < 4; ++rep)
#pragma unroll
			for (int i = 0; i < 16; ++i) { emit<A>(v, d, i, sel); emit<B>(v, d, i, sel); }
	}
	float acc = 0.f;
#pragma unroll
	for (int i = 0; i < 16; ++i) acc += v[i];
#pragma unroll
	for (int i = 0; i < 8; ++i) acc += (float)d[i];
	if (acc == 12345.678f) out[0] = acc;
}
static double g_t[NOPS][NOPS];
template <int A, int B> static void run1(float* out)
{
	const int iters = 1000, blocks = 256 * 8;
	hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	pair_kernel<A, B><<<blocks, 256>>>(out, 10);
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(e0);
	pair_kernel<A, B><<<blocks, 256>>>(out, iters);
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1);
	const double pairs_per_simd = (double)blocks * 4 / (256 * 4) * iters * 64;
	int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	g_t[A][B] = ms * 1e-3 * clk * 1e3 / pairs_per_simd;
}
template <int A, int B> struct Row { static void go(float* out) { run1<A, B>(out); if constexpr (B + 1 < NOPS) Row<A, B + 1>::go(out); } };
template <int A> struct All { static void go(float* out) { Row<A, A>::go(out); if constexpr (A + 1 < NOPS) All<A + 1>::go(out); } };
int main()
{
	float* out; (void)hipMalloc(&out, 4);
	All<0>::go(out);
	printf("cycles per (A, B) pair, per SIMD; diagonal = two of the same\n%-9s", "");
	for (int b = 0; b < NOPS; ++b) printf("%9s", kNames[b]);
	printf("\n");
	for (int a = 0; a < NOPS; ++a) {
		printf("%-9s", kNames[a]);
		for (int b = 0; b < NOPS; ++b) { if (b < a) printf("%9s", ""); else printf("%9.2f", g_t[a][b]); }
		printf("\n");
	}
	return 0;
}
