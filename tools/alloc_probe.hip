// tools/alloc_probe.hip -- developer microbenchmark: read bandwidth of 4 GiB buffers as a function of how / in
// which order they were allocated (plain hipMalloc vs hipExtMallocWithFlags(hipDeviceMallocContiguous)).
//   hipcc -O3 --offload-arch=gfx950 tools/alloc_probe.hip -o /tmp/alloc_probe && /tmp/alloc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_plain(const float4 *p, size_t n4, float *out) {
	float acc = 0;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) out[0] = acc;
}

int main() {
	const size_t bytes = 4ull << 30;
	float *out; CK(hipMalloc(&out, 64));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timeit = [&](const char *name, void *g) {
		CK(hipMemset(g, 0x3c, bytes));
		for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, 0, (const float4 *)g, bytes / 16, out);
		CK(hipEventRecord(e0));
		for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_plain, dim3(2048), dim3(256), 0, 0, (const float4 *)g, bytes / 16, out);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
		printf("%-28s %p  %.3f ms  %.0f GB/s\n", name, g, ms, bytes / ms / 1e6);
	};
	void *p[6];
	for (int i = 0; i < 6; i++) {
		if (i % 2 == 0) CK(hipMalloc(&p[i], bytes));
		else if (hipExtMallocWithFlags(&p[i], bytes, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous alloc failed\n"); p[i] = nullptr; continue; }
		timeit(i % 2 == 0 ? "hipMalloc" : "hipExtMalloc contiguous", p[i]);
	}
	return 0;
}
