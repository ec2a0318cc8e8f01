// Developer probe (GPU box): what a device-to-host copy of one surface array costs by kind of destination memory -
// fresh malloc pages, warm (already touched) malloc pages, the same block after hipHostRegister, hipHostMalloc - and what
// registering / unregistering costs.  hipcc --offload-arch=gfx950 -O2 tools/d2h_probe.hip -o tools/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
	const size_t mb = argc > 1 ? atoi(argv[1]) : 94;
	const size_t n = mb << 20;
	void *d = nullptr;
	CK(hipMalloc(&d, n));
	CK(hipMemset(d, 1, n));
	hipStream_t st;
	CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	for (int thp = 0; thp < 2; thp++) {
		for (int rep = 0; rep < 3; rep++) {
			void *h = nullptr;
			if (posix_memalign(&h, 2u << 20, n)) return 1;
			if (thp) madvise(h, n, MADV_HUGEPAGE);
			double t0 = now();
			CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			double t1 = now();
			CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			double t2 = now();
			CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			double t3 = now();
			hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
			double t4 = now();
			if (e != hipSuccess) { printf("hipHostRegister: %s\n", hipGetErrorString(e)); return 1; }
			CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			double t5 = now();
			CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
			double t6 = now();
			CK(hipHostUnregister(h));
			double t7 = now();
			free(h);
			double t8 = now();
			printf("thp %d  %zu MB: fresh %.2f ms (%.1f GB/s)  warm %.2f  warm %.2f (%.1f GB/s)  register %.2f ms  pinned %.2f  pinned %.2f (%.1f GB/s)  unregister %.2f  free %.2f\n",
			       thp, mb, t1 - t0, n / 1e6 / (t1 - t0), t2 - t1, t3 - t2, n / 1e6 / (t3 - t2), t4 - t3, t5 - t4, t6 - t5, n / 1e6 / (t6 - t5), t7 - t6, t8 - t7);
		}
	}
	void *hp = nullptr;
	double t0 = now();
	CK(hipHostMalloc(&hp, n, hipHostMallocDefault));
	double t1 = now();
	CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
	double t2 = now();
	CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
	double t3 = now();
	// pinned staging + memcpy into a warm malloc block (the extra copy)
	void *h = malloc(n); memset(h, 0, n);
	double t4 = now();
	memcpy(h, hp, n);
	double t5 = now();
	printf("hipHostMalloc %.2f ms; copy %.2f, %.2f ms (%.1f GB/s); memcpy pinned -> warm malloc %.2f ms (%.1f GB/s)\n", t1 - t0, t2 - t1, t3 - t2, n / 1e6 / (t3 - t2), t5 - t4, n / 1e6 / (t5 - t4));
	// colour fill of 15.6 MB fresh vs warm
	{
		const size_t cn = 3903888;
		int *c = (int *)malloc(cn * 4);
		double a0 = now();
		for (size_t k = 0; k < cn; k++) c[k] = 0x5c5c5c;
		double a1 = now();
		for (size_t k = 0; k < cn; k++) c[k] = 0x5c5c5d;
		double a2 = now();
		printf("colour fill of %zu ints: fresh %.2f ms, warm %.2f ms (%d)\n", cn, a1 - a0, a2 - a1, c[cn / 2]);
		free(c);
	}
	return 0;
}
