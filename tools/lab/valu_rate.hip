// lab: issue rate of the vector instructions the B* kernel is made of (cycles per wave64 instruction per SIMD), gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters)
{
	float v[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) v[i] = (float)(threadIdx.x * 16 + i) * 1.0001f;
	double d[8];
#pragma unroll
	for (int i = 0; i < 8; ++i) d[i] = (double)(threadIdx.x + i) * 1.0000001;
	const float sel = (threadIdx.x & 1) ? __builtin_inff() : -__builtin_inff();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				if (OP == 0) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 1) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(sel));
				if (OP == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 6 && i < 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
				if (OP == 7 && i < 8) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
				if (OP == 8 && i < 8) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
				if (OP == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i & 7]) : "v"(d[(i + 1) & 7]));
				if (OP == 12) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 13) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 14) asm volatile("v_min_i32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 15) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(sel));
				if (OP == 16) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
				if (OP == 17) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 18) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 19) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 21) asm volatile("v_min_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 22) asm volatile("v_min_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 23) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "s"(__builtin_amdgcn_read_exec()));
				if (OP == 24) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
				if (OP == 25) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 26) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 27) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(v[(i + 1) & 15]) : "vcc");
				if (OP == 28) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
				if (OP == 29) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
				if (OP == 40 && i == 0) asm volatile(
					"v_xor_b32 %0, %8, %0\nv_xor_b32 %1, %8, %1\nv_xor_b32 %2, %8, %2\nv_xor_b32 %3, %8, %3\nv_xor_b32 %4, %8, %4\nv_xor_b32 %5, %8, %5\nv_xor_b32 %6, %8, %6\nv_xor_b32 %7, %8, %7\n"
					"v_min_f32_dpp %0, -%0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %1, -%1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					"v_min_f32_dpp %2, -%2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %3, -%3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					"v_min_f32_dpp %4, -%4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %5, -%5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					"v_min_f32_dpp %6, -%6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %7, -%7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
					: "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[8]));
				if (OP == 41 && i < 8) { float t; asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(v[i])); asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(t), "v"(sel)); }
				if (OP == 42 && i < 8) { asm volatile("v_xor_b32 %0, %0, %1\nv_min_f32 %0, %0, %2" : "+v"(v[i]) : "v"(v[8]), "v"(v[9])); }
				if (OP == 43 && i == 0) asm volatile(
					"v_xor_b32 %0, %8, %0\nv_xor_b32 %1, %8, %1\nv_xor_b32 %2, %8, %2\nv_xor_b32 %3, %8, %3\nv_xor_b32 %4, %8, %4\nv_xor_b32 %5, %8, %5\nv_xor_b32 %6, %8, %6\nv_xor_b32 %7, %8, %7\n"
					"v_min_f32 %0, %0, %8\nv_min_f32 %1, %1, %8\nv_min_f32 %2, %2, %8\nv_min_f32 %3, %3, %8\nv_min_f32 %4, %4, %8\nv_min_f32 %5, %5, %8\nv_min_f32 %6, %6, %8\nv_min_f32 %7, %7, %8\n"
					: "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(v[8]));
				if (OP == 30) asm volatile("v_min_f32_dpp %0, -%1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
				if (OP == 32) asm volatile("v_min_f32_dpp %0, -%0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
				if (OP == 33) asm volatile("v_min_f32 %0, -%0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
			}
		}
	}
	float acc = 0.f;
#pragma unroll
	for (int i = 0; i < 16; ++i) acc += v[i];
#pragma unroll
	for (int i = 0; i < 8; ++i) acc += (float)d[i];
	if (acc == 12345.678f) out[0] = acc;
}

template <int OP>
static void run(const char* name, int per_iter)
{
	float* out; hipMalloc(&out, 4);
	const int iters = 2000, blocks = 256 * 8;   // 8 workgroups of 4 wavefronts per CU: 8 wavefronts per SIMD
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	rate_kernel<OP><<<blocks, 256>>>(out, 10);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	rate_kernel<OP><<<blocks, 256>>>(out, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double insts_per_simd = (double)blocks * 4 / (256 * 4) * iters * per_iter;   // wave instructions per SIMD
	int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
	printf("%-28s %8.3f ms  %6.2f cycles per wave64 instruction per SIMD (at %d MHz)\n", name, ms, ms * 1e-3 * clk * 1e3 / insts_per_simd, clk / 1000);
	hipFree(out);
}

int main()
{
	run<0>("v_min_f32", 64); run<5>("v_max_f32", 64); run<1>("v_med3_f32", 64); run<2>("v_mov_dpp quad_perm", 64);
	run<3>("v_mov_dpp row_half_mirror", 64); run<4>("v_fma_f32", 64); run<9>("v_mov_b32", 64); run<10>("v_cndmask_b32", 64);
	run<11>("v_pk_fma_f32", 64);
	run<12>("v_min_u32", 64); run<13>("v_max_u32", 64); run<14>("v_min_i32", 64); run<15>("v_med3_i32", 64); run<16>("v_min3_f32", 64);
	run<17>("v_add_f32", 64); run<18>("v_mul_f32", 64); run<19>("v_add_u32", 64); run<20>("v_and_b32", 64);
	run<21>("v_min_u32_dpp quad_perm", 64); run<22>("v_min_f32_dpp quad_perm", 64); run<23>("v_cndmask_b32 (sgpr mask)", 64);
	run<24>("v_max3_u32", 64); run<25>("v_pk_min_u16", 64); run<26>("v_sub_u32", 64); run<27>("v_cmp_lt_u32", 64); run<28>("v_perm_b32", 64); run<29>("v_bfi_b32", 64);
	run<30>("v_min_f32_dpp neg", 64); run<31>("v_xor_b32", 64); run<32>("v_min_f32_dpp neg self half_mirror", 64); run<33>("v_min_f32 neg (VOP3)", 64);
	run<40>("8 xor + 8 min_dpp neg (per 16)", 4 * 16); run<41>("8 x (mov_dpp, med3) (per 16)", 4 * 16); run<42>("8 x (xor, min) alternating (per 16)", 4 * 16); run<43>("8 xor + 8 min (per 16)", 4 * 16);
	run<6>("v_add_f64", 32); run<7>("v_fma_f64", 32); run<8>("v_cvt_f64_f32", 32);
	return 0;
}
