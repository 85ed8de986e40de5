// synth.hip -- device-side synthetic stamp cubes (bench / test utility; not on the reference path).
//
// Same data model as photometry_amd/simulate.py::fill_cubes (after the reference's
// simulation/simulateFITS.py:338-405): stars = mag2flux fluxes spread by a pixel-integrated
// Gaussian PSF with per-cadence jitter, a smooth sinusoidal background, Gaussian noise
// sigma = sqrt(signal + bkg + readnoise^2), a fraction of NaN pixel-cadences.  The noise comes from
// a counter-based hash RNG, so the cubes are a pure function of (seed, target, pixel, cadence).
// Float32 arithmetic is plenty for synthetic inputs.
#include "common.h"
#include <cmath>

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}

__device__ __forceinline__ float u01(uint32_t b) { return ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float gauss_int(float x, float centre, float inv_d) {
	return 0.5f * (erff((x - centre + 0.5f) * inv_d) - erff((x - centre - 0.5f) * inv_d));
}

__global__ __launch_bounds__(256) void tp_synth_kernel(
	int n_cad, int height, int width, int64_t t_pitch, int n_slots,
	const double* __restrict__ star_params, const double* __restrict__ sigma_psf,
	const double* __restrict__ bkg_level, const double* __restrict__ bkg_phase,
	const double* __restrict__ jitter, float readnoise, float nan_fraction, uint64_t seed,
	float* __restrict__ images, float* __restrict__ images_err, float* __restrict__ backgrounds, float* __restrict__ raw)
{
	const int target = blockIdx.x;
	const int P = height * width;
	const int nq = (int)(t_pitch >> 2);
	const int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
	if (idx >= (int64_t)P * nq) return;
	const int p = (int)(idx / nq);
	const int q = (int)(idx - (int64_t)p * nq);
	const int r = p / width, c = p - r * width;
	const float inv_d = 1.0f / (1.41421356f * (float)sigma_psf[target]);
	const float level = (float)bkg_level[target];
	const float phase = (float)bkg_phase[target];
	const double* sp = star_params + (int64_t)target * n_slots * 3;

	float img[4], err[4], bkg[4], rw[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int k = q * 4 + j;
		if (k >= n_cad) { img[j] = 0.f; err[j] = 0.f; bkg[j] = 0.f; rw[j] = 0.f; continue; }
		const float jc = (float)jitter[2 * k], jr = (float)jitter[2 * k + 1];
		// per-cadence multiplicative variability of the main target (same for every pixel)
		const uint64_t hv = splitmix64(seed ^ (0xABCDull << 40) ^ ((uint64_t)target * 1000003ull + (uint64_t)k));
		const float var = 1.0f + 1e-3f * sqrtf(-2.0f * __logf(u01((uint32_t)hv))) * __cosf(6.2831853f * u01((uint32_t)(hv >> 32)));
		float signal = 0.f;
		for (int s = 0; s < n_slots; s++) {
			const float f = (float)sp[3 * s + 2];
			if (f <= 0.f) continue;
			const float g = gauss_int((float)r, (float)sp[3 * s] + jr, inv_d) * gauss_int((float)c, (float)sp[3 * s + 1] + jc, inv_d);
			signal += g * f * (s == 0 ? var : 1.0f);
		}
		const float b = level * (1.0f + 0.05f * __sinf(6.2831853f * (float)k / (float)n_cad * 3.0f + phase));
		const float sigma = sqrtf(signal + b + readnoise * readnoise);
		const uint64_t h = splitmix64(seed ^ (((uint64_t)target * (uint64_t)P + (uint64_t)p) * (uint64_t)n_cad + (uint64_t)k));
		const float gn = sqrtf(-2.0f * __logf(u01((uint32_t)h))) * __cosf(6.2831853f * u01((uint32_t)(h >> 32)));
		const uint64_t h2 = splitmix64(h);
		const bool isnan_px = u01((uint32_t)h2) < nan_fraction;
		const float v = signal + gn * sigma;
		rw[j] = isnan_px ? __builtin_nanf("") : (v + b);
		img[j] = isnan_px ? __builtin_nanf("") : v;
		err[j] = isnan_px ? __builtin_nanf("") : sigma;
		bkg[j] = b;
	}
	const int64_t off = ((int64_t)target * P + p) * t_pitch + (int64_t)q * 4;
	if (images) *reinterpret_cast<float4*>(images + off) = make_float4(img[0], img[1], img[2], img[3]);
	if (images_err) *reinterpret_cast<float4*>(images_err + off) = make_float4(err[0], err[1], err[2], err[3]);
	if (backgrounds) *reinterpret_cast<float4*>(backgrounds + off) = make_float4(bkg[0], bkg[1], bkg[2], bkg[3]);
	if (raw) *reinterpret_cast<float4*>(raw + off) = make_float4(rw[0], rw[1], rw[2], rw[3]);
}

} // namespace

extern "C" int tp_synth_fill(tp_ctx* ctx, const tp_cube_desc* desc, int32_t n_slots,
	const double* d_star_params, const double* d_sigma_psf, const double* d_bkg_level,
	const double* d_bkg_phase, const double* d_jitter, double readnoise, double nan_fraction,
	uint64_t seed, float* d_images, float* d_images_err, float* d_backgrounds, float* d_raw)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_synth_fill: bad cube descriptor");
	TP_REQUIRE(ctx, desc->t_pitch % 4 == 0, "tp_synth_fill: t_pitch must be a multiple of 4");
	TP_REQUIRE(ctx, d_star_params && d_sigma_psf && d_bkg_level && d_bkg_phase && d_jitter && n_slots > 0, "tp_synth_fill: null scene pointer");
	for (const float* p : {d_images, d_images_err, d_backgrounds, d_raw})
		TP_REQUIRE(ctx, p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) == 0, "tp_synth_fill: cubes must be 16-byte aligned");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	const int64_t per_target = (int64_t)desc->height * desc->width * (desc->t_pitch / 4);
	TP_REQUIRE(ctx, (per_target + 255) / 256 <= 65535, "tp_synth_fill: stamp cube too large");
	dim3 grid((unsigned)desc->n_targets, (unsigned)((per_target + 255) / 256)), block(256);
	TP_LAUNCH(ctx, TPK_SYNTH, tp_synth_kernel, grid, block, 0,
		desc->n_cad, desc->height, desc->width, desc->t_pitch, (int)n_slots,
		d_star_params, d_sigma_psf, d_bkg_level, d_bkg_phase, d_jitter, (float)readnoise, (float)nan_fraction, seed,
		d_images, d_images_err, d_backgrounds, d_raw);
	TP_LAUNCH_CHECK(ctx, "tp_synth_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
