// tools/write_probe.hip -- developer microbenchmark (not part of the product): what do WRITES cost a kernel that
// streams reads?  A grid-stride float4 read of a 4 GiB buffer (the sweep's job, simplified) in which every wave, every
// `period` read steps, stores one piece of `piece` bytes (contiguous, whole wave) at a scattered place of a second
// buffer - the sweep's hand-over of bit planes, simplified.  Varies the piece size at constant bytes written, and the
// bytes written at constant piece size.
//   hipcc -O3 --offload-arch=gfx950 tools/write_probe.hip -o /tmp/write_probe && /tmp/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// waves: 4096 (1024 blocks of 256); every wave reads its share in steps of 64 x 16 B = 1 KiB
__global__ __launch_bounds__(256) void k_rw(const float4 *p, size_t n4, uint4 *w, size_t wslots /* 1 KiB slots */, uint32_t period, uint32_t pieces_kib,
                                            float *out) {
	const uint32_t lane = threadIdx.x & 63u;
	const size_t wave = blockIdx.x * 4ull + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4ull;
	float acc = 0;
	uint32_t step = 0;
	uint64_t rnd = wave * 0x9E3779B97F4A7C15ull + 12345;
	for (size_t i = wave * 64 + lane; i < n4; i += nwaves * 64, step++) {
		const float4 v = p[i];
		acc += v.x + v.y + v.z + v.w;
		if (period && (step % period) == period - 1) {
			rnd = rnd * 6364136223846793005ull + 1442695040888963407ull;
			const size_t slot = (rnd >> 20) % (wslots - pieces_kib);
			for (uint32_t k = 0; k < pieces_kib; k++) w[(slot + k) * 64 + lane] = uint4{(uint32_t)step, lane, (uint32_t)k, (uint32_t)wave};
		}
	}
	if (acc == 123.456f) out[0] = acc;
}

int main() {
	const size_t bytes = 4ull << 30, wbytes = 1ull << 30;
	float4 *g; uint4 *w; float *out;
	CK(hipMalloc(&g, bytes)); CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&out, 64));
	CK(hipMemset(g, 0x3c, bytes)); CK(hipMemset(w, 0, wbytes));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int blocks = 1024;  // 4096 waves: one resident round, like k_sweep
	const size_t steps_per_wave = bytes / 1024 / (blocks * 4);  // 1024
	auto run = [&](uint32_t period, uint32_t pieces_kib) {
		for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_rw, dim3(blocks), dim3(256), 0, 0, g, bytes / 16, w, wbytes / 1024, period, pieces_kib, out);
		CK(hipEventRecord(e0));
		const int it = 10;
		for (int i = 0; i < it; i++) hipLaunchKernelGGL(k_rw, dim3(blocks), dim3(256), 0, 0, g, bytes / 16, w, wbytes / 1024, period, pieces_kib, out);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
		const double written = period ? (double)(steps_per_wave / period) * blocks * 4 * pieces_kib * 1024.0 : 0.0;
		printf("period %5u piece %3u KiB: %7.3f ms  read %6.0f GB/s  written %7.1f MB in %7.0f pieces\n", period, pieces_kib, ms, bytes / ms / 1e6, written / 1e6,
		       period ? (double)(steps_per_wave / period) * blocks * 4 : 0.0);
		return ms;
	};
	run(0, 0);
	printf("-- 64 MB written, piece size varies\n");
	run(64, 1); run(128, 2); run(256, 4); run(512, 8); run(1024, 16);
	printf("-- 2 KiB pieces, amount varies\n");
	run(512, 2); run(128, 2); run(32, 2); run(8, 2);
	printf("-- 1 GB written, piece size varies\n");
	run(4, 1); run(8, 2); run(32, 8); run(128, 32);
	run(0, 0);
	return 0;
}
