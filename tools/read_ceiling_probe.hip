// tools/read_ceiling_probe.hip -- developer microbenchmark (round 4): what does a plain read-only kernel reach on this part,
// by blocks per CU, loads in flight per lane and cache policy?  The ceiling bench.py's roofline.read_ceiling is set beside
// (mc33hip_probe_read: 8 blocks per CU, 4 nontemporal 16-byte loads in flight).
//   hipcc -O3 --offload-arch=gfx950 tools/read_ceiling_probe.hip -o tools/read_ceiling_probe && tools/read_ceiling_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const u32x4 *p, size_t n16, unsigned *sink) {
	const size_t stride = (size_t)gridDim.x * 256;
	size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	u32x4 acc = {0, 0, 0, 0};
	for (; i + (U - 1) * stride < n16; i += U * stride) {
		u32x4 v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(p + i + k * stride) : p[i + k * stride];
#pragma unroll
		for (int k = 0; k < U; k++) acc ^= v[k];
	}
	for (; i < n16; i += stride) acc ^= p[i];
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1u);
}
// every block a contiguous piece (a wave reads 1 KiB rows of it one after the other), U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_blocked(const u32x4 *p, size_t n16, unsigned *sink) {
	const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
	u32x4 acc = {0, 0, 0, 0};
	size_t i = lo + threadIdx.x;
	for (; i + (U - 1) * 256 < hi; i += U * 256) {
		u32x4 v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(p + i + k * 256) : p[i + k * 256];
#pragma unroll
		for (int k = 0; k < U; k++) acc ^= v[k];
	}
	for (; i < hi; i += 256) acc ^= p[i];
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) atomicAdd(sink, 1u);
}

// the same contiguous pieces read with DWORD loads (what k_sweep issues: 256 bytes per wave instruction), U in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_blocked_dword(const unsigned *p, size_t n4, unsigned *sink) {
	const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
	unsigned acc = 0;
	size_t i = lo + threadIdx.x;
	for (; i + (U - 1) * 256 < hi; i += U * 256) {
		unsigned v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(p + i + k * 256) : p[i + k * 256];
#pragma unroll
		for (int k = 0; k < U; k++) acc ^= v[k];
	}
	for (; i < hi; i += 256) acc ^= p[i];
	if (acc == 0x9E3779B9u) atomicAdd(sink, 1u);
}
// ... and with 8-byte loads
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read_blocked_b64(const unsigned long long *p, size_t n8, unsigned *sink) {
	const size_t per = (n8 + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < n8 ? lo + per : n8;
	unsigned long long acc = 0;
	size_t i = lo + threadIdx.x;
	for (; i + (U - 1) * 256 < hi; i += U * 256) {
		unsigned long long v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(p + i + k * 256) : p[i + k * 256];
#pragma unroll
		for (int k = 0; k < U; k++) acc ^= v[k];
	}
	for (; i < hi; i += 256) acc ^= p[i];
	if (acc == 0x9E3779B9ull) atomicAdd(sink, 1u);
}

// The sweep's own walk (k_sweep, float, 1024^3 points): a wave = a tile of 256 samples x 64 rows x DEPTH planes, a block = the 4
// tiles of a 1024-sample row group side by side; per plane 16 batches of 4 rows x 4 dword loads (x = 64 k + lane), the next
// batch asked for before the current one is consumed.  SHAPE 0: as the sweep; 1: a wave takes WHOLE 4 KiB rows (16 loads per row,
// a tile of 1024 samples x 16 rows), the 4 waves of a block are 4 row strips of the same 64 rows; 2: as 0 but the 4 waves of a
// block are 4 consecutive z chunks of one (segment, y tile) column instead of 4 segments side by side.
template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(256) void k_tile_walk(const unsigned *g, unsigned n, unsigned *sink) {
	const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned nY = n / 64, nZ = n / DEPTH;
	unsigned seg, yt, zc;
	if (SHAPE == 2) { const unsigned t = blockIdx.x * 4 + wv; zc = t % nZ; const unsigned c = t / nZ; seg = c % 4; yt = c / 4; }
	else { yt = blockIdx.x % nY; zc = blockIdx.x / nY; seg = wv; }
	if (yt >= nY || zc >= nZ) return;
	unsigned acc = 0;
	unsigned dA[16], dB[16];
	auto issue = [&](unsigned (&d)[16], unsigned p, unsigned b) {
		const unsigned *base = g + ((size_t)p * n + yt * 64) * n;
#pragma unroll
		for (int rr = 0; rr < 4; rr++)
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if (SHAPE == 1) d[rr * 4 + k] = __builtin_nontemporal_load(base + (size_t)(wv * 16 + b) * n + (rr * 4 + k) * 64 + lane);  // batch b = ROW wv*16+b, 16 loads along it
				else d[rr * 4 + k] = __builtin_nontemporal_load(base + (size_t)(b * 4 + rr) * n + seg * 256 + k * 64 + lane);
			}
	};
	const unsigned T = DEPTH * 16;
	issue(dA, zc * DEPTH, 0);
	for (unsigned t = 0; t < T; t += 2) {
		const unsigned t1 = t + 1 < T ? t + 1 : t, t2 = t + 2 < T ? t + 2 : t;
		issue(dB, zc * DEPTH + t1 / 16, t1 % 16);
#pragma unroll
		for (int k = 0; k < 16; k++) acc ^= dA[k];
		issue(dA, zc * DEPTH + t2 / 16, t2 % 16);
#pragma unroll
		for (int k = 0; k < 16; k++) acc ^= dB[k];
	}
	if (acc == 0x9E3779B9u) atomicAdd(sink, 1u);
}

int main() {
	const size_t bytes = (size_t)4 << 30;
	u32x4 *g; unsigned *sink;
	CK(hipMalloc(&g, bytes)); CK(hipMalloc(&sink, 64));
	CK(hipMemset(g, 0x3c, bytes)); CK(hipMemset(sink, 0, 64));
	int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	auto timeit = [&](const char *name, auto launch) {
		for (int i = 0; i < 3; i++) launch();
		float best = 1e9f;
		for (int rep = 0; rep < 10; rep++) {
			CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
		}
		printf("%-52s %7.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
	};
	char nm[128];
#define RUN(K, U, NT) for (int bpc : {1, 2, 4, 8, 16}) { snprintf(nm, sizeof nm, #K " U=%d %s blocks/CU=%d", U, NT ? "nt" : "  ", bpc); \
		timeit(nm, [&] { hipLaunchKernelGGL((K<U, NT>), dim3(cus * bpc), dim3(256), 0, 0, g, bytes / 16, sink); }); }
	RUN(k_read, 4, true) RUN(k_read, 8, true) RUN(k_read, 16, true) RUN(k_read, 8, false)
	RUN(k_read_blocked, 8, true) RUN(k_read_blocked, 16, true)
#define RUN2(K, T, DIV, U, NT) for (int bpc : {1, 2, 4, 8}) { snprintf(nm, sizeof nm, #K " U=%d %s blocks/CU=%d", U, NT ? "nt" : "  ", bpc); \
		timeit(nm, [&] { hipLaunchKernelGGL((K<U, NT>), dim3(cus * bpc), dim3(256), 0, 0, (const T *)g, bytes / DIV, sink); }); }
	{
		const unsigned n = 1024;
		timeit("k_tile_walk as the sweep, 16 planes deep (4096 waves)", [&] { hipLaunchKernelGGL((k_tile_walk<0, 16>), dim3((n / 64) * (n / 16)), dim3(256), 0, 0, (const unsigned *)g, n, sink); });
		timeit("k_tile_walk as the sweep, 32 planes deep (2048 waves)", [&] { hipLaunchKernelGGL((k_tile_walk<0, 32>), dim3((n / 64) * (n / 32)), dim3(256), 0, 0, (const unsigned *)g, n, sink); });
		timeit("k_tile_walk as the sweep, 8 planes deep (8192 waves)", [&] { hipLaunchKernelGGL((k_tile_walk<0, 8>), dim3((n / 64) * (n / 8)), dim3(256), 0, 0, (const unsigned *)g, n, sink); });
		timeit("k_tile_walk whole rows per wave, 16 planes deep", [&] { hipLaunchKernelGGL((k_tile_walk<1, 16>), dim3((n / 64) * (n / 16)), dim3(256), 0, 0, (const unsigned *)g, n, sink); });
		timeit("k_tile_walk block = 4 z chunks of a column, 16 deep", [&] { hipLaunchKernelGGL((k_tile_walk<2, 16>), dim3((n / 64) * (n / 16)), dim3(256), 0, 0, (const unsigned *)g, n, sink); });
	}
	RUN2(k_read_blocked_dword, unsigned, 4, 16, true) RUN2(k_read_blocked_dword, unsigned, 4, 32, true) RUN2(k_read_blocked_dword, unsigned, 4, 64, true)
	RUN2(k_read_blocked_b64, unsigned long long, 8, 16, true) RUN2(k_read_blocked_b64, unsigned long long, 8, 32, true)
	return 0;
}
