// Diagnostic: ceiling of a pure streaming READ on this GPU (sum of a float buffer with 16-byte loads), to put the A1 phase
// (5.5 TB/s) in perspective.  hipcc --offload-arch=gfx950 -O3 tools/readbw.hip -o /tmp/readbw && /tmp/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int UNROLL>
__global__ __launch_bounds__(256) void readsum(const float4* __restrict__ p, size_t n4, float* out)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	float acc = 0.f;
	for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
		float4 v[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
		for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
	}
	if (acc == 123.456f) out[0] = acc;
}

int main()
{
	const size_t bytes = (size_t)12 << 30;
	float4* d; float* o;
	hipMalloc(&d, bytes); hipMalloc(&o, 4);
	hipMemset(d, 0, bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int blocks : {2048, 8192, 32768, 131072}) {
		for (int rep = 0; rep < 2; ++rep) {
			hipEventRecord(e0);
			hipLaunchKernelGGL(readsum<8>, dim3(blocks), dim3(256), 0, 0, d, bytes / 16, o);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			if (rep) printf("blocks %d unroll 8: %.3f ms  %.2f TB/s\n", blocks, ms, bytes / (ms * 1e-3) / 1e12);
		}
	}
	return 0;
}
