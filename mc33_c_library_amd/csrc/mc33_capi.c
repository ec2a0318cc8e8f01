/* mc33_capi.c -- host layer in C: the reference's public API (include/marching_cubes_33.h) on top of the
 * device-level C ABI (include/mc33_hip.h).  No HIP types here; everything GPU-side happens behind
 * mc33hip_*.  "MC:" = reference source/marching_cubes_33.c, "UTIL:" = reference source/MC33_util_grd.c.
 *
 * Behaviour kept from the reference:
 *   - create_MC33 snapshots N, r0, d (MC:1758-1782) and returns NULL on failure (MC:1753-1757);
 *   - calculate_isosurface returns a caller-owned `surface` whose T, V, N, color and the struct itself
 *     are five separate malloc blocks (MC:84-92 frees them one by one); an empty result is a zeroed,
 *     non-NULL surface (MC:1880-1883); failure is NULL with M->memoryfault = 1 (MC:1884-1887);
 *   - every vertex gets DefaultColorMC (MC:1875-1877); S->user is left alone (MC:1873 copies 52 bytes).
 * Deliberate difference: the grid is copied to HBM in create_MC33 and stays resident.  A caller that
 * rewrites G->F between calls sets MC33_HIP_REUPLOAD=1 (re-upload before every extraction).
 */
#define _DEFAULT_SOURCE /* madvise, clock_gettime */
#include <malloc.h> /* malloc_usable_size: asked only about blocks this library allocated itself */
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#include <sys/mman.h>

#include "../../include/marching_cubes_33.h"
#include "../../include/mc33_hip.h"

/* layout contract with programs compiled against the reference header */
#ifdef GRD_ORTHOGONAL /* measured against the reference header compiled with -DGRD_ORTHOGONAL */
_Static_assert(sizeof(_GRD) == 256 && offsetof(_GRD, d) == 48 && offsetof(_GRD, periodic) == 84 &&
               offsetof(_GRD, internal_data) == 88 && offsetof(_GRD, title) == 92, "_GRD layout differs from the reference (GRD_ORTHOGONAL)");
#else
_Static_assert(sizeof(_GRD) == 416 && offsetof(_GRD, d) == 48 && offsetof(_GRD, nonortho) == 96 &&
               offsetof(_GRD, internal_data) == 252, "_GRD layout differs from the reference");
#endif
_Static_assert(sizeof(surface) == 64 && offsetof(surface, iso) == 48 && offsetof(surface, user) == 56,
               "surface layout differs from the reference");
#ifdef GRD_ORTHOGONAL
#define MC33_MATS 0 /* bytes of _A, A_ missing from MC33 */
#else
#define MC33_MATS 144
#endif
#if GRD_TYPE_SIZE == 8 /* measured against the reference header compiled with -DGRD_TYPE_SIZE=8 */
_Static_assert(sizeof(MC33) == 200 + MC33_MATS && offsetof(MC33, memoryfault) == 56 && offsetof(MC33, O) == 72 && offsetof(MC33, nx) == 136 &&
               offsetof(MC33, store) == 152 && offsetof(MC33, Dx) == 160 + MC33_MATS, "MC33 layout differs from the reference (double build)");
#else
_Static_assert(sizeof(MC33) == 160 + MC33_MATS && offsetof(MC33, memoryfault) == 52 && offsetof(MC33, nx) == 96 &&
               offsetof(MC33, store) == 112 && offsetof(MC33, Dx) == 120 + MC33_MATS, "MC33 layout differs from the reference");
#endif

#ifndef DEFAULT_SURFACE_COLOR
#define DEFAULT_SURFACE_COLOR 0xff5c5c5c /* grey, 0xAABBGGRR (MC:76-78) */
#endif
#ifndef MC33_NORMAL_NEG
#define MC33_NORMAL_NEG 0 /* 1: front and back exchanged, reference source/libMC33.c:20-22 (the _nneg flavour of the library) */
#endif
int DefaultColorMC = (int)DEFAULT_SURFACE_COLOR;

/* One z-slab of the grid on one device (the same cut as mc33_c_library_amd/slabs.py makes per rank): cell slices
 * [z_begin, z_end), one ghost slice below (its vertices belong to the slab underneath, but the slab's triangles refer to them),
 * resident sample planes [p_lo, p_hi] = the cells' own, one above for the central differences of the normals (MC:888-890,
 * 1036-1038), the ghost slice's lower plane and one below that for vertices on grid points (MC:643-647).  Without
 * MC33_HIP_DEVICES there is one slab: the whole grid on the current device. */
#define MC33_MAX_DEVICES 16
struct staging {     /* device staging of a result, grown on demand; two sets so that calculate_isosurfaces can */
	void *dV, *dN, *dT; /* download one surface while the next is being extracted                           */
	unsigned long long capV, capT;
};
struct mc33_private_s;
typedef struct {
	struct mc33_private_s *owner;
	mc33hip_ctx *ctx;
	int device;
	unsigned z_begin, z_end, ghost, p_lo, p_hi;
	struct staging set[2];
	mc33hip_counts cnt; /* of the count in flight */
	MC33_real iso;
	int rc;
	struct mc33_surface *out;   /* the surface being filled, and where this slab's vertices / triangles begin in it */
	unsigned long long vbase, tbase;
} mc33_slab;

/* private object: the public MC33 first, so callers can keep treating the pointer as MC33* */
typedef struct mc33_private_s {
	MC33 pub;
	unsigned long long magic;
	int nslab;           /* 1, or the number of devices named by MC33_HIP_DEVICES */
	mc33_slab slab[MC33_MAX_DEVICES];
	_GRD *grid;          /* for MC33_HIP_REUPLOAD */
	int reupload;
	int grid_dirty;      /* MC33_grid_changed: the caller rewrote samples of G->F, upload them before the next extraction */
	int inclined;        /* G->nonortho at create time: the MC33_spnC store */
	double grd_A[9], grd_Ai[9];
} mc33_private;
#define MC33_MAGIC 0x4D43333348495031ull /* "MC33HIP1" */

static mc33_private *priv(MC33 *M) {
	mc33_private *p = (mc33_private *)M;
	return (p && p->magic == MC33_MAGIC) ? p : 0;
}

/* runs fn on every slab: the slabs of one device one after the other, the devices side by side - the device-level calls block
 * (count: until the counters are on the host; emit: the runtime's copies into pageable memory return when the data has arrived),
 * so every device gets a thread and its link is busy at the same time as the others'.  (Slabs that share a device - the
 * rehearsal on a one-GPU machine - gain nothing from threads of their own: eight threads copying from one GPU into pieces of the
 * same arrays took 10 ms where one takes 3.5, profiles/r05_capi_walls.txt.) */
struct slab_group {
	void *(*fn)(void *);
	mc33_slab *slab[MC33_MAX_DEVICES];
	int n;
};
static void *slab_group_run(void *arg) {
	struct slab_group *g = (struct slab_group *)arg;
	for (int k = 0; k != g->n; k++)
		g->fn(g->slab[k]);
	return 0;
}
static void for_each_slab(mc33_private *p, void *(*fn)(void *)) {
	struct slab_group grp[MC33_MAX_DEVICES];
	int ngrp = 0;
	for (int k = 0; k != p->nslab; k++) {
		int g = 0;
		while (g != ngrp && grp[g].slab[0]->device != p->slab[k].device) g++;
		if (g == ngrp) { grp[ngrp].fn = fn; grp[ngrp].n = 0; ngrp++; }
		grp[g].slab[grp[g].n++] = &p->slab[k];
	}
	pthread_t th[MC33_MAX_DEVICES];
	int started[MC33_MAX_DEVICES];
	for (int g = 1; g < ngrp; g++)
		started[g] = pthread_create(&th[g], 0, slab_group_run, &grp[g]) == 0;
	slab_group_run(&grp[0]); /* the first device's on the calling thread */
	for (int g = 1; g < ngrp; g++) {
		if (started[g]) pthread_join(th[g], 0);
		else slab_group_run(&grp[g]);
	}
}

static void *slab_create(void *arg) {
	mc33_slab *s = (mc33_slab *)arg;
	mc33_private *p = s->owner;
	const _GRD *G = p->grid;
	mc33hip_grid_desc d;
	memset(&d, 0, sizeof d);
	d.npx = G->N[0] + 1; d.npy = G->N[1] + 1; d.npz_resident = s->p_hi - s->p_lo + 1;
	d.plane0 = s->p_lo; d.nz_total = G->N[2];
	for (int j = 0; j != 3; j++) { d.r0[j] = G->r0[j]; d.d[j] = G->d[j]; }
	d.sample_bytes = (int)sizeof(GRD_data_type);
	d.device = s->device;
	s->rc = mc33hip_create(&s->ctx, &d);
	if (s->rc == MC33HIP_OK) s->rc = mc33hip_set_normal_neg(s->ctx, MC33_NORMAL_NEG);
	if (s->rc == MC33HIP_OK && p->nslab > 1) s->rc = mc33hip_own_stream(s->ctx); /* several contexts: each on a stream of its own */
	if (s->rc == MC33HIP_OK) s->rc = mc33hip_upload_rows(s->ctx, (const void *const *const *)(G->F + s->p_lo));
	return 0;
}

static void *slab_upload(void *arg) {
	mc33_slab *s = (mc33_slab *)arg;
	s->rc = mc33hip_upload_rows(s->ctx, (const void *const *const *)(s->owner->grid->F + s->p_lo));
	return 0;
}

/* MC33_HIP_DEVICES: "all", or a comma-separated list of HIP device ordinals - the z-slabs of the grid go to these devices in this
 * order.  An ordinal may appear several times (several slabs on one GPU: how the path is rehearsed on a one-GPU machine).
 * Returns the number of entries (0: not set - one slab on the current device). */
static int parse_devices(int *dev) {
	const char *e = getenv("MC33_HIP_DEVICES");
	if (!e || !*e)
		return 0;
	const int have = mc33hip_device_count();
	if (have <= 0)
		return 0;
	int n = 0;
	if (!strcmp(e, "all")) {
		for (; n < have && n < MC33_MAX_DEVICES; n++) dev[n] = n;
		return n;
	}
	while (*e && n < MC33_MAX_DEVICES) {
		char *end = 0;
		const long v = strtol(e, &end, 10);
		if (end == e || v < 0 || v >= have)
			return -1; /* not a list of devices of this machine */
		dev[n++] = (int)v;
		e = end;
		while (*e == ',' || *e == ' ') e++;
	}
	return *e ? -1 : n;
}

static int g_objects; /* MC33 objects alive (the host block cache is trimmed when the last one goes) */
static void cache_trim(size_t keep);
static size_t cache_limit(int *is_set);

MC33 *create_MC33(_GRD *G) {
	if (!G || !G->F)
		return 0;
	mc33_private *p = (mc33_private *)calloc(1, sizeof *p);
	if (!p)
		return 0;
	p->magic = MC33_MAGIC;
	MC33 *M = &p->pub;
	M->nx = G->N[0]; M->ny = G->N[1]; M->nz = G->N[2];
	M->F = (const GRD_data_type ***)G->F;
	for (int j = 0; j != 3; j++) { /* MC:1779-1782 */
		M->O[j] = (MC33_real)G->r0[j];
		M->D[j] = (MC33_real)G->d[j];
	}
#ifndef GRD_ORTHOGONAL
	if (G->nonortho) { /* MC:1763-1770: the matrices MC33_spnC multiplies with */
		for (int j = 0; j != 3; j++)
			for (int i = 0; i != 3; i++) {
				M->_A[j][i] = G->_A[j][i] * G->d[i];
				M->A_[j][i] = G->A_[j][i] / G->d[j];
			}
		p->inclined = 1;
		memcpy(p->grd_A, G->_A, sizeof p->grd_A);
		memcpy(p->grd_Ai, G->A_, sizeof p->grd_Ai);
	} else
#endif
	if (G->d[0] != G->d[1] || G->d[1] != G->d[2]) { /* MC:1772-1775 */
		M->ca = (MC33_real)(G->d[2] / G->d[0]);
		M->cb = (MC33_real)(G->d[2] / G->d[1]);
	}
	const char *e = getenv("MC33_HIP_REUPLOAD");
	p->reupload = e && *e && *e != '0';
	p->grid = G;
	__atomic_add_fetch(&g_objects, 1, __ATOMIC_RELAXED);
	int dev[MC33_MAX_DEVICES];
	int n = parse_devices(dev);
	if (G->N[0] < 1 || G->N[1] < 1 || G->N[2] < 1 || n < 0) {
		free_MC33(M);
		return 0;
	}
	if (n == 0) { n = 1; dev[0] = -1; } /* the current device */
	if ((unsigned)n > G->N[2]) n = (int)G->N[2]; /* a slab is at least one cell slice */
	p->nslab = n;
	for (int k = 0; k != n; k++) { /* even split along z (slabs.py: Slab) */
		mc33_slab *s = &p->slab[k];
		s->owner = p;
		s->device = dev[k];
		s->z_begin = (unsigned)((unsigned long long)k * G->N[2] / (unsigned)n);
		s->z_end = (unsigned)((unsigned long long)(k + 1) * G->N[2] / (unsigned)n);
		s->ghost = k ? 1u : 0u;
		s->p_lo = s->z_begin >= s->ghost + 1u ? s->z_begin - s->ghost - 1u : 0u;
		s->p_hi = s->z_end + 1u <= G->N[2] ? s->z_end + 1u : G->N[2];
	}
	for_each_slab(p, slab_create); /* the 1 / n of G->F of every device goes over its own link */
	for (int k = 0; k != n; k++)
		if (p->slab[k].rc != MC33HIP_OK) {
			free_MC33(M);
			return 0;
		}
	return M;
}

void free_MC33(MC33 *M) {
	mc33_private *p = priv(M);
	if (!p)
		return;
	for (int q = 0; q != p->nslab; q++) {
		mc33_slab *s = &p->slab[q];
		if (!s->ctx)
			continue;
		for (int k = 0; k != 2; k++) {
			if (s->set[k].dV) mc33hip_device_free(s->ctx, s->set[k].dV);
			if (s->set[k].dN) mc33hip_device_free(s->ctx, s->set[k].dN);
			if (s->set[k].dT) mc33hip_device_free(s->ctx, s->set[k].dT);
		}
		mc33hip_destroy(s->ctx);
	}
	p->magic = 0;
	free(p);
	if (__atomic_sub_fetch(&g_objects, 1, __ATOMIC_RELAXED) == 0)
		cache_trim(cache_limit(0)); /* no extractor left: what free_surface_memory kept beyond the plain limit goes back */
}

/* --- mult_Abf (reference header :186-191, MC33_util_grd.c:86-114) --------------------------------- */
void _multTSA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int t) {
	if (t) { /* rows of A^T, last first: c may alias b */
		c[2] = A[0][2] * b[0] + A[1][2] * b[1] + A[2][2] * b[2];
		c[1] = A[0][1] * b[0] + A[1][1] * b[1];
		c[0] = A[0][0] * b[0];
	} else {
		c[0] = A[0][0] * b[0] + A[0][1] * b[1] + A[0][2] * b[2];
		c[1] = A[1][1] * b[1] + A[1][2] * b[2];
		c[2] = A[2][2] * b[2];
	}
}

void _multA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int t) {
	const int r = t ? 3 : 1, s = t ? 1 : 3; /* element (i, j) of the matrix applied = ((const double *)A)[i*s + j*r] */
	const double *a = &A[0][0];
	const double c0 = a[0] * b[0] + a[r] * b[1] + a[2 * r] * b[2];
	const double c1 = a[s] * b[0] + a[s + r] * b[1] + a[s + 2 * r] * b[2];
	c[2] = a[2 * s] * b[0] + a[2 * s + r] * b[1] + a[2 * s + 2 * r] * b[2];
	c[0] = c0;
	c[1] = c1;
}

void (*mult_Abf)(const double (*)[3], MC33_real *, MC33_real *, int) = _multA_bf;

static int refresh_grid(mc33_private *p) {
	if (p->inclined) { /* the reference calls through mult_Abf for every vertex (MC:608, 612) */
		if (mult_Abf != _multA_bf && mult_Abf != _multTSA_bf)
			return MC33HIP_EINVAL; /* a caller-supplied function cannot run on the GPU */
		for (int k = 0; k != p->nslab; k++)
			if (mc33hip_set_inclined(p->slab[k].ctx, p->grd_A, p->grd_Ai, mult_Abf == _multTSA_bf) != MC33HIP_OK)
				return MC33HIP_EINVAL;
	}
	if (!p->reupload && !p->grid_dirty)
		return 0;
	p->grid_dirty = 0;
	for_each_slab(p, slab_upload);
	for (int k = 0; k != p->nslab; k++)
		if (p->slab[k].rc != MC33HIP_OK)
			return p->slab[k].rc;
	return MC33HIP_OK;
}

/* Extension (not in the reference, which reads G->F anew on every call, MC:1792, 1832-1868): tells the extractor that
 * samples of the grid it was created from have been rewritten.  The next size_of_isosurface / calculate_isosurface(s)
 * uploads G->F again first.  The _GRD and its rows must still be alive then, as for the reference's own M->F. */
void MC33_grid_changed(MC33 *M) {
	mc33_private *p = priv(M);
	if (p)
		p->grid_dirty = 1;
}

/* count pass of one slab (blocks until its counters are on the host) */
static void *slab_count(void *arg) {
	mc33_slab *s = (mc33_slab *)arg;
	mc33hip_range r;
	r.z_begin = s->z_begin; r.z_end = s->z_end; r.ghost_below = s->ghost; r.id_base = 0;
	memset(&s->cnt, 0, sizeof s->cnt);
	s->rc = mc33hip_count(s->ctx, (double)s->iso, &r, &s->cnt);
	return 0;
}

/* Counts of every slab, side by side; the totals in *tot.  Vertex ids and triangle slots of slab k begin at the sums over the
 * slabs below it (the reference numbers vertices in sweep order, so z-slabs concatenate). */
static int count_slabs(mc33_private *p, MC33_real iso, mc33hip_counts *tot) {
	memset(tot, 0, sizeof *tot);
	int rc = refresh_grid(p);
	if (rc != MC33HIP_OK)
		return rc;
	for (int k = 0; k != p->nslab; k++) p->slab[k].iso = iso;
	for_each_slab(p, slab_count);
	for (int k = 0; k != p->nslab; k++) {
		if (p->slab[k].rc != MC33HIP_OK)
			return p->slab[k].rc;
		tot->nV += p->slab[k].cnt.nV; tot->nT += p->slab[k].cnt.nT; tot->active_cells += p->slab[k].cnt.active_cells;
	}
	return (tot->nV > 0xFFFFFFFFull || tot->nT > 0xFFFFFFFFull) ? MC33HIP_EOVERFLOW : MC33HIP_OK;
}

unsigned long long size_of_isosurface(MC33 *M, MC33_real iso, unsigned int *nV, unsigned int *nT) {
	mc33_private *p = priv(M);
	mc33hip_counts cnt;
	memset(&cnt, 0, sizeof cnt);
	if (p) {
		M->iso = iso;
		if (count_slabs(p, iso, &cnt) != MC33HIP_OK)
			memset(&cnt, 0, sizeof cnt);
	}
	if (nV) *nV = (unsigned int)cnt.nV;
	if (nT) *nT = (unsigned int)cnt.nT;
	/* MC:1939 */
	return cnt.nV * (6 * sizeof(MC33_real) + sizeof(int)) + cnt.nT * (3 * sizeof(int)) + sizeof(surface);
}

static int ensure_staging(mc33_slab *s, struct staging *g, unsigned long long nV, unsigned long long nT) {
	if (g->capV < nV) {
		if (g->dV) mc33hip_device_free(s->ctx, g->dV);
		if (g->dN) mc33hip_device_free(s->ctx, g->dN);
		g->dV = g->dN = 0; g->capV = 0;
		unsigned long long cap = nV + nV / 8 + 1024;
		if (mc33hip_device_alloc(s->ctx, &g->dV, cap * 3 * sizeof(MC33_real)) != MC33HIP_OK) return -1;
		if (mc33hip_device_alloc(s->ctx, &g->dN, cap * 12) != MC33HIP_OK) return -1;
		g->capV = cap;
	}
	if (g->capT < nT) {
		if (g->dT) mc33hip_device_free(s->ctx, g->dT);
		g->dT = 0; g->capT = 0;
		unsigned long long cap = nT + nT / 8 + 1024;
		if (mc33hip_device_alloc(s->ctx, &g->dT, cap * 12) != MC33HIP_OK) return -1;
		g->capT = cap;
	}
	return 0;
}

/* GPU part of one surface of calculate_isosurfaces (one slab: the whole grid): the surface of `iso` into staging set g (device
 * memory), its sizes into *cnt */
static int extract_to_staging(mc33_private *p, struct staging *g, MC33_real iso, mc33hip_counts *cnt) {
	mc33_slab *s = &p->slab[0];
	mc33hip_range r;
	r.z_begin = 0; r.z_end = p->pub.nz; r.ghost_below = 0; r.id_base = 0;
	memset(cnt, 0, sizeof *cnt);
	int rc = refresh_grid(p);
	if (rc != MC33HIP_OK)
		return rc;
	/* one pass with the staging buffers of an earlier call; if they are too small the counts come back
	 * anyway, the buffers grow and only the emit pass is repeated */
	rc = mc33hip_extract(s->ctx, iso, &r, g->dV, g->dN, g->dT, g->capV, g->capT, cnt);
	if (rc == MC33HIP_ECAPACITY) {
		rc = ensure_staging(s, g, cnt->nV, cnt->nT) ? MC33HIP_ENOMEM : mc33hip_emit(s->ctx, g->dV, g->dN, g->dT, g->capV, g->capT);
		/* mc33hip_emit only enqueues: the set must be complete before anybody reads it - the helper thread of
		 * calculate_isosurfaces copies on a stream of its own, which is not ordered after this one */
		if (rc == MC33HIP_OK)
			rc = mc33hip_synchronize(s->ctx);
	}
	return rc;
}

/* One array of a caller-owned surface: plain free() releases it (MC:84-92).  Large ones start on a 2 MiB boundary and
 * ask for transparent huge pages: a 1024^3 surface is 200 MB, and touching it for the first time in 4 KiB pages (the
 * copy from the GPU, the colour fill, the munmap in free) cost more than the extraction itself - measured on the GPU box
 * (tools/d2h_probe.hip, profiles/r05_d2h_probe.txt): a 94 MB array arrives in 1.76 ms (56 GB/s, the rate of the link - pinning
 * the block with hipHostRegister first gains nothing) when its pages are mapped, in 4 - 5 ms when they are fresh, and giving
 * them back to the kernel costs another 4 - 6 ms; the 15 MB colour fill 0.16 against 3.0 ms.
 *
 * So free_surface_memory keeps large blocks for the next surface instead of releasing them.  How much it keeps: as much as the
 * surface it is releasing holds (an unmodified viewer loop - calculate_isosurface, draw, free_surface_memory, next isovalue:
 * reference GLUT_example/TestMC33_glut.c:421-458 - then always finds the blocks of the surface before), and at least 64 MB.
 * Blocks of earlier, larger surfaces that do not fit under that bound any more are freed, and when the last MC33 object is
 * destroyed everything beyond 64 MB goes back.  MC33_HOST_CACHE_MB in the environment replaces the rule by a fixed bound
 * (0 = keep nothing).
 *
 * Only blocks this library allocated are ever looked at: every large block handed out is entered in a small table
 * (address, size); free_surface_memory looks the pointer up there, and an address it does not know is simply passed to
 * free().  A table hit is double-checked with the allocator (a caller may have freed our block and got the same
 * address back for a smaller one of its own): it counts as ours only if the allocator still holds at least the
 * recorded size there.  The blocks are ordinary malloc blocks at all times. */
#define HUGE_PAGE ((size_t)2 << 20)
#define CACHE_MIN_BLOCK (2 * HUGE_PAGE)
#define CACHE_SLOTS 16
#define LIVE_SLOTS 256
#define CACHE_DEFAULT ((size_t)64 << 20)
static struct {
	void *p;
	size_t cap;
	unsigned long long age; /* (cache only) when it was put back: the oldest goes first */
} g_cache[CACHE_SLOTS], g_live[LIVE_SLOTS]; /* kept for reuse / handed out and still with the caller */
static size_t g_cache_bytes;
static unsigned long long g_cache_clock;
static pthread_mutex_t g_cache_lock = PTHREAD_MUTEX_INITIALIZER;

/* the plain bound: MC33_HOST_CACHE_MB, 64 MB when it is not set (*is_set says which) */
static size_t cache_limit(int *is_set) {
	const char *e = getenv("MC33_HOST_CACHE_MB");
	if (is_set) *is_set = e && *e;
	return e && *e ? (size_t)strtoull(e, 0, 10) << 20 : CACHE_DEFAULT;
}

/* (lock held) frees kept blocks, oldest first, until at most `keep` bytes are left */
static void cache_trim_locked(size_t keep) {
	while (g_cache_bytes > keep) {
		int old = -1;
		for (int k = 0; k != CACHE_SLOTS; k++)
			if (g_cache[k].p && (old < 0 || g_cache[k].age < g_cache[old].age))
				old = k;
		if (old < 0)
			break;
		free(g_cache[old].p);
		g_cache_bytes -= g_cache[old].cap;
		g_cache[old].p = 0;
	}
}
static void cache_trim(size_t keep) {
	pthread_mutex_lock(&g_cache_lock);
	cache_trim_locked(keep);
	pthread_mutex_unlock(&g_cache_lock);
}

/* (lock held) remember a large block that goes to the caller; when the table is full the block is simply not known
 * later and will be freed like a foreign one */
static void live_add(void *p, size_t cap) {
	for (int k = 0; k != LIVE_SLOTS; k++)
		if (!g_live[k].p || g_live[k].p == p) {
			g_live[k].p = p;
			g_live[k].cap = cap;
			return;
		}
}

/* Fresh large blocks - pages the process has never touched - are populated by several threads at once before the copies from the
 * GPU are started (populate_fresh).  The reference's own viewers KEEP every surface they make (GLUT_example/TestMC33_glut.c:421-458
 * appends to a list), so for them every call lands in fresh pages: populated by the copy itself, one fault after the other, a
 * 1024^3 surface costs 8 ms of page faults; profiles/r05_capi_walls.txt. */
struct fresh_block { char *p; size_t n; };
static __thread struct fresh_block t_fresh[4];
static __thread int t_nfresh;

/* a block of at least `bytes` bytes; *cap = what it can really hold */
static void *surface_block(size_t bytes, size_t *cap) {
	void *p = 0;
	if (bytes >= CACHE_MIN_BLOCK) {
		int best = -1;
		pthread_mutex_lock(&g_cache_lock);
		for (int k = 0; k != CACHE_SLOTS; k++) /* best fit, never more than twice the need */
			if (g_cache[k].p && g_cache[k].cap >= bytes && g_cache[k].cap <= 2 * bytes && (best < 0 || g_cache[k].cap < g_cache[best].cap))
				best = k;
		if (best >= 0) {
			p = g_cache[best].p;
			*cap = g_cache[best].cap;
			g_cache[best].p = 0;
			g_cache_bytes -= *cap;
			live_add(p, *cap);
		}
		pthread_mutex_unlock(&g_cache_lock);
		if (p)
			return p;
		const size_t rounded = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
		if (posix_memalign(&p, HUGE_PAGE, rounded) == 0) {
			(void)madvise(p, rounded, MADV_HUGEPAGE);
			*cap = rounded;
			if (t_nfresh < 4) { t_fresh[t_nfresh].p = (char *)p; t_fresh[t_nfresh].n = bytes; t_nfresh++; }
			pthread_mutex_lock(&g_cache_lock);
			live_add(p, rounded);
			pthread_mutex_unlock(&g_cache_lock);
			return p;
		}
	}
	p = malloc(bytes ? bytes : 1);
	*cap = bytes;
	return p;
}

/* (lock held) is p a large block of ours that the allocator still holds?  Its capacity, and the table entry is given up */
static size_t live_take(void *p) {
	if (!p)
		return 0;
	for (int k = 0; k != LIVE_SLOTS; k++)
		if (g_live[k].p == p) {
			const size_t cap = g_live[k].cap;
			g_live[k].p = 0;
			return malloc_usable_size(p) >= cap ? cap : 0;
		}
	return 0;
}

/* The counterpart used by free_surface_memory and adjustvectorlenght_s: the n arrays of ONE surface go back together.  Those
 * that are large blocks of ours are kept for the next surface as far as the bound allows - the bound being what these very
 * blocks hold, or 64 MB if that is more (MC33_HOST_CACHE_MB, when set, instead) - the others are freed. */
static void surface_blocks_release(void **blk, int n) {
	size_t cap[8], mine = 0;
	int is_set = 0;
	size_t limit = cache_limit(&is_set);
	pthread_mutex_lock(&g_cache_lock);
	for (int k = 0; k != n; k++) {
		cap[k] = live_take(blk[k]);
		mine += cap[k];
	}
	if (!is_set && mine > limit)
		limit = mine;
	cache_trim_locked(limit > mine ? limit - mine : 0); /* older blocks make room for these */
	for (int k = 0; k != n; k++) {
		if (!cap[k] || g_cache_bytes + cap[k] > limit)
			continue;
		for (int q = 0; q != CACHE_SLOTS; q++)
			if (!g_cache[q].p) {
				g_cache[q].p = blk[k];
				g_cache[q].cap = cap[k];
				g_cache[q].age = ++g_cache_clock;
				g_cache_bytes += cap[k];
				blk[k] = 0;
				break;
			}
	}
	pthread_mutex_unlock(&g_cache_lock);
	for (int k = 0; k != n; k++)
		free(blk[k]);
}
static void surface_block_release(void *p) { surface_blocks_release(&p, 1); }

/* the four arrays of a surface with nV vertices and nT triangles (caller-owned malloc blocks, MC:84-92); 0 when memory is short */
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23 /* Linux 5.14 */
#endif
static void *populate_thread(void *arg) {
	struct fresh_block *b = (struct fresh_block *)arg;
	if (madvise(b->p, b->n, MADV_POPULATE_WRITE) != 0) /* (older kernels: touch every page - nothing has been written into the block yet) */
		for (size_t off = 0; off < b->n; off += 4096) ((volatile char *)b->p)[off] = 0;
	return 0;
}
/* the fresh blocks surface_alloc has just made (this thread's list), in pieces of at most 32 MB, a thread per piece (16 at most) */
static void populate_fresh(void) {
	struct fresh_block piece[16];
	pthread_t th[16];
	int n = 0;
	for (int k = 0; k != t_nfresh; k++)
		for (size_t off = 0; off < t_fresh[k].n && n != 16; off += (size_t)32 << 20) {
			piece[n].p = t_fresh[k].p + off;
			piece[n].n = t_fresh[k].n - off < ((size_t)32 << 20) ? t_fresh[k].n - off : (size_t)32 << 20;
			n++;
		}
	t_nfresh = 0;
	int started[16];
	for (int k = 1; k < n; k++) started[k] = pthread_create(&th[k], 0, populate_thread, &piece[k]) == 0;
	if (n) populate_thread(&piece[0]);
	for (int k = 1; k < n; k++) {
		if (started[k]) pthread_join(th[k], 0);
		else populate_thread(&piece[k]);
	}
}

static surface *surface_alloc(size_t nV, size_t nT, MC33_real iso) {
	surface *S = (surface *)malloc(sizeof(surface));
	if (!S)
		return 0;
	t_nfresh = 0;
	memset(S, 0, sizeof(surface)); /* (an empty surface is exactly this, MC:1880-1883) */
	if (!nV)
		return S;
	size_t capV = 0, capN = 0, capT = 0, capC = 0;
	S->V = (MC33_real(*)[3])surface_block(nV * 3 * sizeof(MC33_real), &capV);
	S->N = (float(*)[3])surface_block(nV * 3 * sizeof(float), &capN);
	S->T = (unsigned int(*)[3])surface_block((nT ? nT : 1) * 3 * sizeof(int), &capT);
	const int fresh_vnt = t_nfresh; /* (the colour block is populated by whoever fills it) */
	S->color = (int *)surface_block(nV * sizeof(int), &capC);
	t_nfresh = fresh_vnt;
	if (!S->V || !S->N || !S->T || !S->color) {
		t_nfresh = 0;
		free_surface_memory(S);
		return 0;
	}
	S->nV = (unsigned int)nV; S->nT = (unsigned int)nT;
	/* capacities in elements, as the reference keeps them (MC:94-127 shrinks arrays whose cap exceeds the count) */
	size_t cv = capV / (3 * sizeof(MC33_real)), ct = capT / (3 * sizeof(int));
	if (capN / (3 * sizeof(float)) < cv) cv = capN / (3 * sizeof(float));
	if (capC / sizeof(int) < cv) cv = capC / sizeof(int);
	S->capv = (unsigned int)(cv > 0xFFFFFFFFu ? 0xFFFFFFFFu : cv);
	S->capt = (unsigned int)(ct > 0xFFFFFFFFu ? 0xFFFFFFFFu : ct);
	S->iso = iso;
	return S;
}

static void fill_color(surface *S) { /* MC:1875-1877 */
	const int col = DefaultColorMC;
	int *c = S->color;
	for (size_t k = 0, n = S->nV; k != n; k++)
		c[k] = col;
}
static void *fill_color_thread(void *arg) {
	fill_color((surface *)arg);
	return 0;
}

/* host part of calculate_isosurfaces: a caller-owned `surface` filled from staging set g.
 * concurrent: copy on the side stream, beside whatever the context is computing */
static surface *surface_from_staging(mc33_private *p, const struct staging *g, const mc33hip_counts *cnt, MC33_real iso, int concurrent) {
	surface *S = surface_alloc((size_t)cnt->nV, (size_t)cnt->nT, iso);
	if (!S || !S->nV)
		return S;
	populate_fresh();
	void *const dst[3] = {S->V, S->N, S->T};
	const void *const src[3] = {g->dV, g->dN, g->dT};
	const size_t bytes[3] = {(size_t)S->nV * 3 * sizeof(MC33_real), (size_t)S->nV * 12, (size_t)S->nT * 12};
	if (mc33hip_download_many(p->slab[0].ctx, 3, dst, src, bytes, concurrent) != MC33HIP_OK) {
		free_surface_memory(S);
		return 0;
	}
	fill_color(S);
	return S;
}

/* emit of one slab at its global base, its arrays copied straight to their place in the caller's blocks */
static void *slab_emit(void *arg) {
	mc33_slab *s = (mc33_slab *)arg;
	struct staging *g = &s->set[0];
	surface *S = s->out;
	s->rc = MC33HIP_OK;
	if (!s->cnt.nV && !s->cnt.nT)
		return 0;
	if (ensure_staging(s, g, s->cnt.nV, s->cnt.nT)) { s->rc = MC33HIP_ENOMEM; return 0; }
	if ((s->rc = mc33hip_set_id_base(s->ctx, (unsigned int)s->vbase)) != MC33HIP_OK) return 0;
	s->rc = mc33hip_emit_download(s->ctx, g->dV, g->dN, g->dT, g->capV, g->capT, S->V + s->vbase, S->N + s->vbase, S->T + s->tbase);
	const int w = mc33hip_download_wait(s->ctx); /* (also after a failure: nothing may still be writing into the blocks when they are released) */
	if (s->rc == MC33HIP_OK) s->rc = w;
	return 0;
}

/* MC:1816-1889.  Count on every slab (one, unless MC33_HIP_DEVICES names several devices) -> the sizes of the surface and where
 * each slab's part begins -> the caller's arrays (recycled blocks, see above) -> every slab emits its vertices and triangles at
 * its global base and copies them straight to their place in those arrays, each array as soon as the passes that write it are
 * through, over its own device's link: no device-to-device exchange, nothing to concatenate -> the colours are filled in by a
 * helper thread while the copies run. */
static double now_ms(void) {
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

surface *calculate_isosurface(MC33 *M, MC33_real iso) {
	mc33_private *p = priv(M);
	if (!p)
		return 0;
	M->nT = M->nV = 0;
	M->memoryfault = 0;
	M->iso = iso;
	mc33hip_counts tot;
	static int trace = -1; /* MC33_CAPI_TRACE=1: where the wall time of a call goes (stderr) */
	if (trace < 0) trace = getenv("MC33_CAPI_TRACE") != 0;
	double t[6] = {0, 0, 0, 0, 0, 0};
	if (trace) t[0] = now_ms();
	const int crc = count_slabs(p, iso, &tot);
	if (trace) t[1] = now_ms();
	surface *S = crc == MC33HIP_OK ? surface_alloc((size_t)tot.nV, (size_t)tot.nT, iso) : 0;
	if (trace) t[2] = now_ms();
	if (S && S->nV) {
		int ok = 1;
		unsigned long long vb = 0, tb = 0;
		for (int k = 0; k != p->nslab; k++) {
			mc33_slab *s = &p->slab[k];
			s->out = S; s->vbase = vb; s->tbase = tb;
			vb += s->cnt.nV; tb += s->cnt.nT;
		}
		pthread_t ct;
		const int helper = S->nV >= 65536u && pthread_create(&ct, 0, fill_color_thread, S) == 0; /* 16 MB at 1024^3: 0.8 ms beside 3.4 ms of copies */
		populate_fresh(); /* (blocks that did not come from the cache: their pages, before the copies need them) */
		if (trace) t[3] = now_ms();
		for_each_slab(p, slab_emit);
		if (trace) t[4] = now_ms();
		if (helper) pthread_join(ct, 0);
		else fill_color(S);
		for (int k = 0; k != p->nslab; k++)
			if (p->slab[k].rc != MC33HIP_OK)
				ok = 0;
		if (trace) {
			t[5] = now_ms();
			fprintf(stderr, "[mc33 capi] count %.3f  blocks %.3f  helper + fresh pages %.3f  emit + copies %.3f  colours joined %.3f  total %.3f ms\n", t[1] - t[0], t[2] - t[1],
			        t[3] - t[2], t[4] - t[3], t[5] - t[4], t[5] - t[0]);
		}
		if (!ok) {
			free_surface_memory(S);
			S = 0;
		}
	}
	if (!S) {
		M->memoryfault = 1;
		return 0;
	}
	if (S->nV) { /* mirror of the copy MC:1873 makes into the MC33 object's public prefix */
		M->T = S->T; M->V = S->V; M->N = S->N; M->color = S->color;
		M->nT = S->nT; M->capt = S->capt; M->capv = S->capv;
	}
	return S;
}

/* --- extension: several isovalues of the resident grid ---------------------------------------------------- */
struct download_job {
	mc33_private *p;
	const struct staging *g;
	mc33hip_counts cnt;
	MC33_real iso;
	surface *S;
};

static void *download_thread(void *arg) {
	struct download_job *j = (struct download_job *)arg;
	j->S = surface_from_staging(j->p, j->g, &j->cnt, j->iso, 1);
	return 0;
}

unsigned int calculate_isosurfaces(MC33 *M, const MC33_real *iso, unsigned int n, surface **out) {
	mc33_private *p = priv(M);
	unsigned int done = 0;
	if (!out)
		return 0;
	for (unsigned int k = 0; k != n; k++)
		out[k] = 0;
	if (!p || !iso)
		return 0;
	M->memoryfault = 0;
	if (p->nslab > 1) { /* several devices: every surface by itself (each call already runs its slabs side by side) */
		for (unsigned int k = 0; k != n; k++)
			if ((out[k] = calculate_isosurface(M, iso[k])) != 0)
				done++;
		M->memoryfault = done != n;
		return done;
	}
	if (refresh_grid(p) != MC33HIP_OK)
		return 0;
	struct download_job job;
	pthread_t th;
	int running = 0;
	unsigned int running_k = 0;
	for (unsigned int k = 0; k != n; k++) {
		if (k % 8 == 0 && n - k > 1) { /* the sweeps of the next (up to) 8 isovalues in one or two passes over the grid */
			double many[8];
			const unsigned int m = n - k < 8 ? n - k : 8;
			mc33hip_range r;
			r.z_begin = 0; r.z_end = M->nz; r.ghost_below = 0; r.id_base = 0;
			for (unsigned int q = 0; q != m; q++) many[q] = iso[k + q];
			if (!p->reupload) /* (the re-upload of every call would drop the sweeps made ahead) */
				(void)mc33hip_sweep_many(p->slab[0].ctx, many, (int)m, &r); /* (on failure the single calls sweep for themselves) */
		}
		/* surface k is computed into set k&1 while the helper thread copies surface k-1 out of the other set */
		struct staging *g = &p->slab[0].set[k & 1];
		mc33hip_counts cnt;
		M->iso = iso[k];
		const int rc = extract_to_staging(p, g, iso[k], &cnt);
		if (running) {
			pthread_join(th, 0);
			running = 0;
			out[running_k] = job.S;
		}
		if (rc != MC33HIP_OK)
			continue;
		job.p = p; job.g = g; job.cnt = cnt; job.iso = iso[k]; job.S = 0;
		if (pthread_create(&th, 0, download_thread, &job) == 0) {
			running = 1;
			running_k = k;
		} else
			out[k] = surface_from_staging(p, g, &cnt, iso[k], 0);
	}
	if (running) {
		pthread_join(th, 0);
		out[running_k] = job.S;
	}
	for (unsigned int k = 0; k != n; k++) {
		if (out[k]) done++;
		else M->memoryfault = 1;
	}
	return done;
}

void free_surface_memory(surface *S) { /* MC:84-92 */
	if (S) {
		void *blk[4] = {S->T, S->V, S->N, S->color};
		surface_blocks_release(blk, 4);
		free(S);
	}
}

void adjustvectorlenght_s(surface *S) { /* MC:94-127: shrink the four arrays to nV / nT elements */
	if (!S)
		return;
	if (S->capv > S->nV) {
		void *c = malloc(sizeof(int) * (size_t)S->nV), *n = malloc(3 * sizeof(float) * (size_t)S->nV),
		     *v = malloc(3 * sizeof(MC33_real) * (size_t)S->nV);
		if (!c || !n || !v) { free(c); free(n); free(v); return; }
		memcpy(c, S->color, sizeof(int) * (size_t)S->nV);
		memcpy(n, S->N, 3 * sizeof(float) * (size_t)S->nV);
		memcpy(v, S->V, 3 * sizeof(MC33_real) * (size_t)S->nV);
		surface_block_release(S->color); surface_block_release(S->N); surface_block_release(S->V);
		S->color = (int *)c; S->N = (float(*)[3])n; S->V = (MC33_real(*)[3])v;
		S->capv = S->nV;
	}
	if (S->capt > S->nT) {
		void *t = malloc(3 * sizeof(int) * (size_t)S->nT);
		if (!t) return;
		memcpy(t, S->T, 3 * sizeof(int) * (size_t)S->nT);
		surface_block_release(S->T);
		S->T = (unsigned int(*)[3])t;
		S->capt = S->nT;
	}
}

/* ---- grid container helpers (UTIL:125-169, 585-686), needed by callers that build grids ---------- */
/* cell angles 90 degrees, identity cell matrices (the members do not exist in GRD_ORTHOGONAL builds) */
static void set_orthogonal(_GRD *Z) {
#ifndef GRD_ORTHOGONAL
	Z->nonortho = 0;
	memset(Z->_A, 0, sizeof Z->_A);
	memset(Z->A_, 0, sizeof Z->A_);
	for (int i = 0; i != 3; i++) {
		Z->Ang[i] = 90.0f;
		Z->_A[i][i] = Z->A_[i][i] = 1.0;
	}
#else
	(void)Z;
#endif
}

void free_memory_grd(_GRD *Z) { /* UTIL:125-145 */
	if (!Z)
		return;
	if (Z->F) {
		for (unsigned int k = 0; k <= Z->N[2] && Z->F[k]; k++) {
			if (Z->internal_data)
				for (unsigned int j = 0; j <= Z->N[1]; j++)
					free(Z->F[k][j]);
			free(Z->F[k]);
		}
		free(Z->F);
	}
	free(Z);
}

int alloc_F(_GRD *Z) { /* UTIL:147-169: every row is its own allocation */
	const size_t np = (size_t)Z->N[2] + 1, nr = (size_t)Z->N[1] + 1, nc = (size_t)Z->N[0] + 1;
	Z->F = (GRD_data_type ***)calloc(np + 1, sizeof(void *)); /* one spare NULL terminates partial grids */
	if (!Z->F)
		return -1;
	Z->internal_data = 1;
	for (size_t k = 0; k != np; k++) {
		Z->F[k] = (GRD_data_type **)calloc(nr, sizeof(void *));
		if (!Z->F[k])
			return -1;
		for (size_t j = 0; j != nr; j++)
			if (!(Z->F[k][j] = (GRD_data_type *)malloc(nc * sizeof(GRD_data_type))))
				return -1;
	}
	return 0;
}

_GRD *grid_from_data_pointer(unsigned int Nx, unsigned int Ny, unsigned int Nz, GRD_data_type *data) { /* UTIL:585-627 */
	if (!data || !Nx || !Ny || !Nz)
		return 0;
	_GRD *Z = (_GRD *)calloc(1, sizeof(_GRD));
	if (!Z)
		return 0;
	Z->F = (GRD_data_type ***)calloc((size_t)Nz + 1, sizeof(void *));
	if (!Z->F) { free(Z); return 0; }
	Z->N[0] = Nx - 1; Z->N[1] = Ny - 1; Z->N[2] = Nz - 1;
	for (unsigned int k = 0; k != Nz; k++) {
		Z->F[k] = (GRD_data_type **)malloc((size_t)Ny * sizeof(void *));
		if (!Z->F[k]) { free_memory_grd(Z); return 0; }
		for (unsigned int j = 0; j != Ny; j++)
			Z->F[k][j] = data + ((size_t)k * Ny + j) * Nx;
	}
	for (int i = 0; i != 3; i++) {
		Z->L[i] = (float)Z->N[i];
		Z->d[i] = 1.0;
		Z->r0[i] = 0.0;
	}
	set_orthogonal(Z);
	return Z;
}

_GRD *generate_grid_from_fn(double xi, double yi, double zi, double xf, double yf, double zf, double dx, double dy,
                            double dz, double (*fn)(double x, double y, double z)) { /* UTIL:630-686 */
	if (dx <= 0 || dy <= 0 || dz <= 0 || xi == xf || yi == yf || zi == zf)
		return 0;
	double lo[3] = {xi, yi, zi}, hi[3] = {xf, yf, zf}, st[3] = {dx, dy, dz};
	_GRD *Z = (_GRD *)calloc(1, sizeof(_GRD));
	if (!Z)
		return 0;
	for (int i = 0; i != 3; i++) {
		if (lo[i] > hi[i]) { double t = lo[i]; lo[i] = hi[i]; hi[i] = t; }
		if (hi[i] - lo[i] < st[i]) st[i] = hi[i] - lo[i];
		Z->N[i] = (unsigned int)(int)((hi[i] - lo[i]) / st[i] + 0.5); /* intervals, UTIL:649-651 */
		Z->d[i] = st[i];
		Z->r0[i] = lo[i];
	}
	if (alloc_F(Z)) { free_memory_grd(Z); return 0; }
	if (fn) { /* coordinates advance by repeated addition, UTIL:660-672 */
		double z = lo[2];
		for (unsigned int k = 0; k <= Z->N[2]; k++, z += st[2]) {
			double y = lo[1];
			for (unsigned int j = 0; j <= Z->N[1]; j++, y += st[1]) {
				double x = lo[0];
				GRD_data_type *row = Z->F[k][j];
				for (unsigned int i = 0; i <= Z->N[0]; i++, x += st[0])
					row[i] = (GRD_data_type)fn(x, y, z);
			}
		}
	}
	for (int i = 0; i != 3; i++) {
		Z->L[i] = (float)(Z->N[i] * Z->d[i]);
	}
	set_orthogonal(Z);
	return Z;
}
