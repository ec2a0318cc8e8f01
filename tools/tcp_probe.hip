// tools/tcp_probe.hip -- developer microbenchmark (not part of the product): what a wave pays in the vector L1 (TCP)
// for one load instruction, by how its 64 lanes spread over cache lines.  The vertex emit pass of round 2 asked for
// 12 short pieces per record with every lane in a different sample row (64 lines per instruction); this probe prices the
// alternatives of a cooperative staging load - one row per lane, per pair of lanes, per quad, fully coalesced - for data
// that sits in the L1 (same addresses again and again) and in the L2 (a footprint walked round and round).
//   hipcc -O3 --offload-arch=gfx950 tools/tcp_probe.hip -o /tmp/tcp_probe && /tmp/tcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// PAT: 0 = lane-per-row, dword      1 = lane-per-row, dwordx2     2 = lane-per-row, dwordx4
//      3 = pair-per-row, dwordx4 (32 B of a row per pair)         4 = quad-per-row, dwordx4 (64 B of a row per quad)
//      5 = coalesced dwordx4 (1 KiB per instruction)              6 = coalesced dword (256 B per instruction)
//      7 = 8 lanes per row, dwordx4 (128 B = one line per 8 lanes)
// KOFF: byte distance between the 8 loads a lane keeps in flight (512: eight different lines of its row; 16: one line - with step 0 the
// wave then touches 64 lines, 8 KiB, and all waves start at row 0: L1 hits)
// rows are `pitch` bytes apart; a wave owns 64 consecutive rows of the footprint; step: rows the wave moves on per
// iteration (0: the same lines every time = L1 hits after the first)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int PAT, int KOFF>
__global__ __launch_bounds__(256) void k_probe(const char *buf, uint32_t nrows, uint32_t pitch, uint32_t iters, uint32_t step, float *out) {
	const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
	uint32_t row0 = step ? (wave * 64u) % nrows : 0u;
	uint32_t rsel, off;
	if (PAT <= 2) { rsel = lane; off = 0; }
	else if (PAT == 3) { rsel = lane >> 1; off = (lane & 1u) * 16u; }
	else if (PAT == 4) { rsel = lane >> 2; off = (lane & 3u) * 16u; }
	else if (PAT == 7) { rsel = lane >> 3; off = (lane & 7u) * 16u; }
	else if (PAT == 5) { rsel = 0; off = lane * 16u; }
	else { rsel = 0; off = lane * 4u; }
	float acc = 0;
	for (uint32_t it = 0; it < iters; it++) {
		const char *p = buf + (size_t)((row0 + rsel) % nrows) * pitch + off;
		// 8 loads in flight, at 8 different column offsets of the rows (512 B apart: different lines)
		if (PAT == 0 || PAT == 6) {
			float v[8];
#pragma unroll
			for (int k = 0; k < 8; k++) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(v[k]) : "v"(p), "n"(k * KOFF));
			asm volatile("s_waitcnt vmcnt(0)" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");  // (the loaded registers are inputs of the wait: they stay allocated while the loads are in flight - without that the compiler reuses them, and the data that arrives late lands in a pointer)
#pragma unroll
			for (int k = 0; k < 8; k++) acc += v[k];
		} else if (PAT == 1) {
			v2f v[8];
#pragma unroll
			for (int k = 0; k < 8; k++) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(v[k]) : "v"(p), "n"(k * KOFF));
			asm volatile("s_waitcnt vmcnt(0)" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");  // (the loaded registers are inputs of the wait: they stay allocated while the loads are in flight - without that the compiler reuses them, and the data that arrives late lands in a pointer)
#pragma unroll
			for (int k = 0; k < 8; k++) acc += v[k].x + v[k].y;
		} else {
			v4f v[8];
#pragma unroll
			for (int k = 0; k < 8; k++) {
				if (PAT == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(p + (size_t)k * pitch));
				else asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v[k]) : "v"(p), "n"(k * KOFF));
			}
			asm volatile("s_waitcnt vmcnt(0)" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]) : "memory");  // (the loaded registers are inputs of the wait: they stay allocated while the loads are in flight - without that the compiler reuses them, and the data that arrives late lands in a pointer)
#pragma unroll
			for (int k = 0; k < 8; k++) acc += v[k].x + v[k].y + v[k].z + v[k].w;
		}
		row0 = (row0 + step) % nrows;
	}
	if (acc == 123.456f) out[0] = acc;
}

int main() {
	const uint32_t pitch = 4096;
	const size_t max_bytes = 512ull << 20;
	char *buf; float *out;
	CK(hipMalloc(&buf, max_bytes + 65536)); CK(hipMalloc(&out, 64));
	CK(hipMemset(buf, 0x3c, max_bytes + 65536));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	const char *names[8] = {"row/lane dword", "row/lane dwordx2", "row/lane dwordx4", "row/pair dwordx4 (32 B)", "row/quad dwordx4 (64 B)", "coalesced dwordx4 (1 KiB)", "coalesced dword (256 B)", "row/8 lanes dwordx4 (128 B)"};
	auto run = [&](int pat, uint32_t blocks, uint32_t nrows, uint32_t iters, uint32_t step) {
		auto launch = [&] {
			switch (pat) {
			case 0: if (step) hipLaunchKernelGGL((k_probe<0, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<0, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 1: if (step) hipLaunchKernelGGL((k_probe<1, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<1, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 2: if (step) hipLaunchKernelGGL((k_probe<2, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<2, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 3: if (step) hipLaunchKernelGGL((k_probe<3, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<3, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 4: if (step) hipLaunchKernelGGL((k_probe<4, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<4, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 5: if (step) hipLaunchKernelGGL((k_probe<5, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<5, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			case 6: if (step) hipLaunchKernelGGL((k_probe<6, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<6, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			default: if (step) hipLaunchKernelGGL((k_probe<7, 512>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); else hipLaunchKernelGGL((k_probe<7, 16>), dim3(blocks), dim3(256), 0, 0, buf, nrows, pitch, iters, step, out); break;
			}
		};
		launch(); launch();
		CK(hipEventRecord(e0));
		const int reps = 5;
		for (int i = 0; i < reps; i++) launch();
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
		const double wave_instr = (double)blocks * 4 * iters * 8;                 // load instructions issued, chip-wide
		const double ns_per_instr_cu = ms * 1e6 / (wave_instr / cus);             // per CU: time between load instructions
		return ns_per_instr_cu;
	};
	for (int waves_per_cu : {4, 8, 16}) {
		const uint32_t blocks = (uint32_t)cus * (uint32_t)waves_per_cu / 4;
		printf("---- %d waves per CU (%u blocks) ---- ns per load instruction and CU (cycles at 2.4 GHz)\n", waves_per_cu, blocks);
		printf("%-30s %18s %18s %18s\n", "pattern", "L1-hot (step 0)", "L2 (32 MiB walk)", "HBM (512 MiB walk)");
		for (int pat = 0; pat < 8; pat++) {
			const double a = run(pat, blocks, 8192, 400, 0);                      // same 64 rows per wave: L1
			const double b = run(pat, blocks, 8192, 400, 64 * 17);                // 8192 rows x 4 KiB = 32 MiB: every XCD's L2 sees all of it... 4 MiB L2: mostly Infinity Cache
			const double c = run(pat, blocks, (uint32_t)(max_bytes / pitch), 200, 64 * 1031);
			printf("%-30s %8.2f (%6.1f) %8.2f (%6.1f) %8.2f (%6.1f)\n", names[pat], a, a * 2.4, b, b * 2.4, c, c * 2.4);
		}
	}
	// footprint that fits one XCD's L2: 512 rows x 4 KiB = 2 MiB
	printf("---- 16 waves per CU, 2 MiB footprint (L2 hits) ----\n");
	for (int pat = 0; pat < 8; pat++) {
		const double b = run(pat, (uint32_t)cus * 4, 512, 400, 64 * 3);
		printf("%-30s %8.2f (%6.1f)\n", names[pat], b, b * 2.4);
	}
	return 0;
}
