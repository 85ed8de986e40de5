// Calibration of rocprofv3's FETCH_SIZE on gfx950 by load width and access shape: every kernel reads a known number of bytes
// exactly once from a 4 GiB buffer (far beyond the 256 MiB Infinity Cache); run under `rocprofv3 --pmc FETCH_SIZE` and divide.
//   hipcc --offload-arch=gfx950 -O3 tools/fetchcal.hip -o tools/fetchcal
// Shapes: W4 / W8 / W16 = fully coalesced 4 / 8 / 16 bytes per lane (sum image: 16, fused kernel: 8 and 16);
// SEG32 = eight consecutive lanes on 32 contiguous bytes of one row, eight rows 5 248 bytes apart per wave instruction (the
// stamp-background kernel: lane = (cadence, pixel)); ROW4 = 4 bytes per lane, one 256-byte piece per wave instruction out of rows
// 5 248 bytes apart, the next instruction in another row (the LinPSF fit's walk).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void coalesced(const T* __restrict__ p, size_t n, float* out)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	float acc = 0.f;
	for (; i + 3 * stride < n; i += 4 * stride) {
		T v[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) v[u] = p[i + u * stride];
#pragma unroll
		for (int u = 0; u < 4; ++u) acc += reinterpret_cast<const float*>(&v[u])[0];
	}
	if (acc == 123.456f) out[0] = acc;
}

// rows of `pitch` floats; a workgroup owns 32 consecutive columns of `rows_per_block` rows: lane = (column 0..7 of the wave's 8, row 0..7)
__global__ __launch_bounds__(256) void seg32(const float* __restrict__ p, int pitch, int n_rows, float* out)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int col = blockIdx.y * 32 + wave * 8 + (lane >> 3), g = lane & 7;
	const float* base = p + (size_t)blockIdx.x * 256 * pitch;      // 256 rows per block
	float acc = 0.f;
	if (col < pitch)
		for (int j = 0; j < 32; ++j) acc += base[(size_t)(j * 8 + g) * pitch + col];
	if (acc == 123.456f) out[0] = acc;
	(void)n_rows;
}

__global__ __launch_bounds__(256) void row4(const float* __restrict__ p, int pitch, int n_rows, float* out)
{
	const int col = blockIdx.y * 256 + threadIdx.x;
	const float* base = p + (size_t)blockIdx.x * 256 * pitch;
	float acc = 0.f;
	if (col < pitch)
		for (int j = 0; j < 256; ++j) acc += base[(size_t)j * pitch + col];
	if (acc == 123.456f) out[0] = acc;
	(void)n_rows;
}

int main()
{
	const size_t bytes = (size_t)4 << 30;
	void* d; float* o;
	if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
	(void)hipMemset(d, 0, bytes);
	const int blocks = 8192;
	// every kernel reads bytes_read bytes: printed so that the counter can be divided by it
	hipLaunchKernelGGL(coalesced<float>, dim3(blocks), dim3(256), 0, 0, (const float*)d, bytes / 4, o);
	hipLaunchKernelGGL(coalesced<float2>, dim3(blocks), dim3(256), 0, 0, (const float2*)d, bytes / 8, o);
	hipLaunchKernelGGL(coalesced<float4>, dim3(blocks), dim3(256), 0, 0, (const float4*)d, bytes / 16, o);
	const int pitch = 1312;
	const int n_rows = (int)(bytes / 4 / pitch) / 256 * 256;
	hipLaunchKernelGGL(seg32, dim3(n_rows / 256, (pitch + 31) / 32), dim3(256), 0, 0, (const float*)d, pitch, n_rows, o);
	hipLaunchKernelGGL(row4, dim3(n_rows / 256, (pitch + 255) / 256), dim3(256), 0, 0, (const float*)d, pitch, n_rows, o);
	if (hipDeviceSynchronize() != hipSuccess) return 2;
	const size_t n4 = bytes / 4, per = (size_t)blocks * 256 * 4;
	printf("bytes_read coalesced<float> %zu coalesced<float2> %zu coalesced<float4> %zu seg32 %zu row4 %zu\n",
		(n4 / per) * per * 4, ((bytes / 8) / per) * per * 8, ((bytes / 16) / per) * per * 16, (size_t)n_rows * pitch * 4, (size_t)n_rows * pitch * 4);
	return 0;
}
