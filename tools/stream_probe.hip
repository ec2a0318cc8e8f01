// tools/stream_probe.hip -- developer microbenchmark (not part of the product): how fast can a 4 GiB float
// volume be READ on this GPU with (a) a plain grid-stride float4 sweep, (b) the tile walk of k_sweep with
// dword loads and no arithmetic, (c) the same walk with 16-byte loads.  Used to separate "access pattern"
// from "instruction stream" when tuning k_sweep (measurement notes).
//   hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_plain(const float4 *p, size_t n4, float *out) {
	float acc = 0;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) out[0] = acc;
}

// tile walk: block = 4 waves side by side in x (256 samples each), 64 rows per plane, rz+1 planes
template <int VEC, int BATCH>
__global__ __launch_bounds__(256) void k_tile(const float *g, uint32_t n, uint32_t nYT, uint32_t rz, float *out) {
	const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const uint32_t yt = blockIdx.x % nYT, zc = blockIdx.x / nYT;
	const uint32_t y0 = yt * 63, nrows = min(64u, n - y0), z_lo = zc * rz, z_hi = min(z_lo + rz, n - 1);
	float acc = 0;
	for (uint32_t p = z_lo; p <= z_hi; p++) {
		const float *plane = g + ((size_t)p * n + y0) * n + wv * 256;
		for (uint32_t r = 0; r < nrows; r += BATCH) {
			if (VEC == 1) {
				float d[BATCH][4];
#pragma unroll
				for (int rr = 0; rr < BATCH; rr++)
#pragma unroll
					for (int k = 0; k < 4; k++) d[rr][k] = plane[(size_t)min(r + rr, nrows - 1) * n + 64 * k + lane];
#pragma unroll
				for (int rr = 0; rr < BATCH; rr++)
#pragma unroll
					for (int k = 0; k < 4; k++) acc += d[rr][k];
			} else {
				float4 d[BATCH];
#pragma unroll
				for (int rr = 0; rr < BATCH; rr++) d[rr] = *(const float4 *)(plane + (size_t)min(r + rr, nrows - 1) * n + 4 * lane);
#pragma unroll
				for (int rr = 0; rr < BATCH; rr++) acc += d[rr].x + d[rr].y + d[rr].z + d[rr].w;
			}
		}
	}
	if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
	const uint32_t n = 1024;
	const size_t bytes = (size_t)n * n * n * 4;
	float *g, *out;
	CK(hipMalloc(&g, bytes)); CK(hipMalloc(&out, 64));
	CK(hipMemset(g, 0x3c, bytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timeit = [&](const char *name, auto launch, double traffic) {
		for (int i = 0; i < 3; i++) launch();
		CK(hipEventRecord(e0));
		const int it = 20;
		for (int i = 0; i < it; i++) launch();
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
		printf("%-34s %7.3f ms  %7.1f GB/s algorithmic  (%7.1f GB/s incl. halo re-reads)\n", name, ms, bytes / ms / 1e6, traffic / ms / 1e6);
	};
	for (int blocks : {2048, 4096, 8192, 16384})
		timeit(("plain float4, blocks=" + std::to_string(blocks)).c_str(), [&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(256), 0, 0, (const float4 *)g, bytes / 16, out); }, (double)bytes);
	const uint32_t nYT = (n - 1 + 62) / 63;
	for (uint32_t rz : {8u, 16u, 32u, 64u}) {
		const uint32_t nZC = (n - 1 + rz - 1) / rz;
		const double traffic = (double)bytes * (64.0 / 63.0) * ((rz + 1.0) / rz);
		char nm[96];
		snprintf(nm, sizeof nm, "tile dword  batch4  rz=%u", rz);
		timeit(nm, [&] { hipLaunchKernelGGL((k_tile<1, 4>), dim3(nYT * nZC), dim3(256), 0, 0, g, n, nYT, rz, out); }, traffic);
		snprintf(nm, sizeof nm, "tile dword  batch8  rz=%u", rz);
		timeit(nm, [&] { hipLaunchKernelGGL((k_tile<1, 8>), dim3(nYT * nZC), dim3(256), 0, 0, g, n, nYT, rz, out); }, traffic);
		snprintf(nm, sizeof nm, "tile float4 batch4  rz=%u", rz);
		timeit(nm, [&] { hipLaunchKernelGGL((k_tile<4, 4>), dim3(nYT * nZC), dim3(256), 0, 0, g, n, nYT, rz, out); }, traffic);
		snprintf(nm, sizeof nm, "tile float4 batch16 rz=%u", rz);
		timeit(nm, [&] { hipLaunchKernelGGL((k_tile<4, 16>), dim3(nYT * nZC), dim3(256), 0, 0, g, n, nYT, rz, out); }, traffic);
	}
	return 0;
}
