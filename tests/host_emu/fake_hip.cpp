// tests/host_emu/fake_hip.cpp -- TEST INFRASTRUCTURE ONLY.  Never linked into a product library.
//
// The part of the device-level C ABI (include/mc33_hip.h) that the host layer csrc/mc33_capi.c calls, served by the host
// emulator (emu.cpp: the device's per-cell logic run serially on the CPU) on "device memory" that is plain malloc memory.
// Linked with the REAL mc33_capi.c into tests/host_emu/libMC33_hostlogic_<type>.so, it lets the CPU suite run the host logic of
// the reference's API exactly as shipped - the z-slab cut behind MC33_HIP_DEVICES, the thread per device, prefix sums and
// offsets into the caller's arrays, the host block cache, calculate_isosurfaces, MC33_grid_changed, the error paths - in a
// container that has no GPU (tests/test_capi_host_logic.py).  What it cannot check is the kernels: the GPU suite does that.
//
// "Devices": MC33_FAKE_DEVICES of them (default 4), all the same CPU; every entry point takes one lock (the emulator keeps its
// plane window in globals), so the threads mc33_capi.c starts per device really run, one at a time.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/mc33_hip.h"

#if defined(FAKE_U16)
typedef uint16_t sample_t;
#define EMU_SLAB emu_slab_u16
#else
typedef float sample_t;
#define EMU_SLAB emu_slab_f32
#endif

struct emu_surface {
	uint32_t nV, nT;
	float *V, *N;
	uint32_t *T;
};
struct emu_slab {
	uint32_t z_begin, z_end, ghost, id_base;
	uint32_t plane_lo, plane_hi;
};
extern "C" int EMU_SLAB(const sample_t *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0, const double *d, float iso,
                        const emu_slab *slab, emu_surface *out, unsigned long long *violations);
extern "C" void emu_free(emu_surface *s);

static std::mutex g_lock;
static unsigned long long g_out_of_window = 0;  // reads of planes a context does not hold, over all calls (must stay 0)
static unsigned long long g_calls[8];           // create, upload, count, emit, emit_download, extract, sweep_many, download_many

struct mc33hip_ctx {
	mc33hip_grid_desc desc;
	std::vector<sample_t> grid;  // the WHOLE grid's extent, only the resident planes ever written (the others stay 0: a read of them is counted)
	bool uploaded = false, counted = false;
	mc33hip_range range{};
	double iso = 0;
	mc33hip_counts counts{};
};

static int run(mc33hip_ctx *c, unsigned id_base, emu_surface *s) {
	const mc33hip_grid_desc &d = c->desc;
	emu_slab sl{c->range.z_begin, c->range.z_end, c->range.ghost_below ? 1u : 0u, id_base, d.plane0, d.plane0 + d.npz_resident - 1u};
	unsigned long long viol = 0;
	const int rc = EMU_SLAB(c->grid.data(), d.npx, d.npy, d.nz_total + 1u, d.r0, d.d, (float)c->iso, &sl, s, &viol);
	g_out_of_window += viol;
	return rc == 0 ? MC33HIP_OK : MC33HIP_ERUNTIME;
}

extern "C" {

unsigned long long fake_out_of_window_reads(void) { return g_out_of_window; }
unsigned long long fake_calls(int k) { return k >= 0 && k < 8 ? g_calls[k] : 0; }

const char *mc33hip_last_error(void) { return ""; }
int mc33hip_device_count(void) {
	const char *e = getenv("MC33_FAKE_DEVICES");
	return e && *e ? atoi(e) : 4;
}

int mc33hip_create(mc33hip_ctx **out, const mc33hip_grid_desc *d) {
	if (!out || !d) return MC33HIP_EINVAL;
	*out = nullptr;
	if (d->sample_bytes != (int)sizeof(sample_t) || d->npx < 2 || d->npy < 2 || d->npz_resident < 2 || d->nz_total < 1) return MC33HIP_EINVAL;
	if ((uint64_t)d->plane0 + d->npz_resident > (uint64_t)d->nz_total + 1) return MC33HIP_EINVAL;
	if (d->device >= mc33hip_device_count()) return MC33HIP_ENOGPU;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[0]++;
	mc33hip_ctx *c = new mc33hip_ctx;
	c->desc = *d;
	c->grid.assign((size_t)d->npx * d->npy * (d->nz_total + 1u), (sample_t)0);
	*out = c;
	return MC33HIP_OK;
}
void mc33hip_destroy(mc33hip_ctx *c) { delete c; }
int mc33hip_set_normal_neg(mc33hip_ctx *c, int on) { return c && !on ? MC33HIP_OK : MC33HIP_EINVAL; }  // (the emulator has the default orientation only)
int mc33hip_own_stream(mc33hip_ctx *c) { return c ? MC33HIP_OK : MC33HIP_EINVAL; }
int mc33hip_set_inclined(mc33hip_ctx *c, const double *A, const double *Ai, int) { return c && !A && !Ai ? MC33HIP_OK : MC33HIP_EINVAL; }
int mc33hip_synchronize(mc33hip_ctx *c) { return c ? MC33HIP_OK : MC33HIP_EINVAL; }
int mc33hip_download_wait(mc33hip_ctx *c) { return c ? MC33HIP_OK : MC33HIP_EINVAL; }

int mc33hip_upload_rows(mc33hip_ctx *c, const void *const *const *F) {
	if (!c || !F) return MC33HIP_EINVAL;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[1]++;
	const mc33hip_grid_desc &d = c->desc;
	for (unsigned k = 0; k < d.npz_resident; k++)
		for (unsigned j = 0; j < d.npy; j++)
			memcpy(&c->grid[((size_t)(d.plane0 + k) * d.npy + j) * d.npx], F[k][j], (size_t)d.npx * sizeof(sample_t));
	c->uploaded = true;
	c->counted = false;
	return MC33HIP_OK;
}

int mc33hip_count(mc33hip_ctx *c, double iso, const mc33hip_range *r, mc33hip_counts *out) {
	if (!c || !r || !c->uploaded || r->z_begin >= r->z_end || r->z_end > c->desc.nz_total) return MC33HIP_EINVAL;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[2]++;
	c->range = *r; c->iso = iso; c->counted = false;
	emu_surface s;
	if (run(c, 0u, &s) != MC33HIP_OK) return MC33HIP_ERUNTIME;
	memset(&c->counts, 0, sizeof c->counts);
	c->counts.nV = s.nV; c->counts.nT = s.nT;
	emu_free(&s);
	if (out) *out = c->counts;
	c->counted = true;
	return MC33HIP_OK;
}

int mc33hip_set_id_base(mc33hip_ctx *c, unsigned int id_base) {
	if (!c || !c->counted) return MC33HIP_EINVAL;
	c->range.id_base = id_base;
	return MC33HIP_OK;
}

static int emit_into(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT) {
	if (!c->counted) return MC33HIP_EINVAL;
	if (capV < c->counts.nV || capT < c->counts.nT) return MC33HIP_ECAPACITY;
	emu_surface s;
	if (run(c, c->range.id_base, &s) != MC33HIP_OK) return MC33HIP_ERUNTIME;
	if (s.nV != c->counts.nV || s.nT != c->counts.nT) { emu_free(&s); return MC33HIP_ERUNTIME; }
	if (s.nV) { memcpy(dV, s.V, (size_t)s.nV * 12); memcpy(dN, s.N, (size_t)s.nV * 12); }
	if (s.nT) memcpy(dT, s.T, (size_t)s.nT * 12);
	emu_free(&s);
	return MC33HIP_OK;
}

int mc33hip_emit(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT) {
	if (!c) return MC33HIP_EINVAL;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[3]++;
	return emit_into(c, dV, dN, dT, capV, capT);
}

int mc33hip_emit_download(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT, void *hV, void *hN, void *hT) {
	if (!c) return MC33HIP_EINVAL;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[4]++;
	const int rc = emit_into(c, dV, dN, dT, capV, capT);
	if (rc != MC33HIP_OK) return rc;
	if (c->counts.nV) { memcpy(hV, dV, (size_t)c->counts.nV * 12); memcpy(hN, dN, (size_t)c->counts.nV * 12); }
	if (c->counts.nT) memcpy(hT, dT, (size_t)c->counts.nT * 12);
	return MC33HIP_OK;
}

int mc33hip_extract(mc33hip_ctx *c, double iso, const mc33hip_range *r, void *dV, void *dN, void *dT, unsigned long long capV,
                    unsigned long long capT, mc33hip_counts *out) {
	const int rc = mc33hip_count(c, iso, r, out);
	if (rc != MC33HIP_OK) return rc;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[5]++;
	return emit_into(c, dV, dN, dT, capV, capT);  // (MC33HIP_ECAPACITY with the counts in *out, like the device)
}

int mc33hip_sweep_many(mc33hip_ctx *c, const double *isos, int n, const mc33hip_range *r) {
	if (!c || !isos || n < 1 || n > 8 || !r) return MC33HIP_EINVAL;
	g_calls[6]++;
	return MC33HIP_OK;  // (an optimisation of the device: nothing to do here)
}

int mc33hip_download_many(mc33hip_ctx *c, int n, void *const *dst, const void *const *src, const size_t *bytes, int) {
	if (!c) return MC33HIP_EINVAL;
	std::lock_guard<std::mutex> g(g_lock);
	g_calls[7]++;
	for (int k = 0; k < n; k++)
		if (bytes[k]) memcpy(dst[k], src[k], bytes[k]);
	return MC33HIP_OK;
}

int mc33hip_device_alloc(mc33hip_ctx *c, void **p, size_t bytes) {
	if (!c || !p) return MC33HIP_EINVAL;
	*p = malloc(bytes ? bytes : 16);
	return *p ? MC33HIP_OK : MC33HIP_ENOMEM;
}
int mc33hip_device_free(mc33hip_ctx *c, void *p) {
	if (!c) return MC33HIP_EINVAL;
	free(p);
	return MC33HIP_OK;
}

}  // extern "C"
