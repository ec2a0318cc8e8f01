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
#define _DEFAULT_SOURCE /* madvise */
#include <malloc.h> /* malloc_usable_size: asked only about blocks this library allocated itself */
#include <pthread.h>
#include <stddef.h>
#include <stdlib.h>
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

/* private object: the public MC33 first, so callers can keep treating the pointer as MC33* */
typedef struct {
	MC33 pub;
	unsigned long long magic;
	mc33hip_ctx *ctx;
	_GRD *grid;          /* for MC33_HIP_REUPLOAD */
	struct staging {     /* device staging of a result, grown on demand; two sets so that calculate_isosurfaces can */
		void *dV, *dN, *dT; /* download one surface while the next is being extracted                           */
		unsigned long long capV, capT;
	} set[2];
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
	mc33hip_grid_desc d;
	memset(&d, 0, sizeof d);
	d.npx = G->N[0] + 1; d.npy = G->N[1] + 1; d.npz_resident = G->N[2] + 1;
	d.plane0 = 0; d.nz_total = G->N[2];
	for (int j = 0; j != 3; j++) { d.r0[j] = G->r0[j]; d.d[j] = G->d[j]; }
	d.sample_bytes = (int)sizeof(GRD_data_type);
	d.device = -1;
	const char *e = getenv("MC33_HIP_REUPLOAD");
	p->reupload = e && *e && *e != '0';
	p->grid = G;
	if (G->N[0] < 1 || G->N[1] < 1 || G->N[2] < 1 || mc33hip_create(&p->ctx, &d) != MC33HIP_OK ||
	    mc33hip_set_normal_neg(p->ctx, MC33_NORMAL_NEG) != MC33HIP_OK ||
	    mc33hip_upload_rows(p->ctx, (const void *const *const *)G->F) != MC33HIP_OK) {
		free_MC33(M);
		return 0;
	}
	return M;
}

void free_MC33(MC33 *M) {
	mc33_private *p = priv(M);
	if (!p)
		return;
	if (p->ctx) {
		for (int k = 0; k != 2; k++) {
			if (p->set[k].dV) mc33hip_device_free(p->ctx, p->set[k].dV);
			if (p->set[k].dN) mc33hip_device_free(p->ctx, p->set[k].dN);
			if (p->set[k].dT) mc33hip_device_free(p->ctx, p->set[k].dT);
		}
		mc33hip_destroy(p->ctx);
	}
	p->magic = 0;
	free(p);
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
		if (mc33hip_set_inclined(p->ctx, p->grd_A, p->grd_Ai, mult_Abf == _multTSA_bf) != MC33HIP_OK)
			return MC33HIP_EINVAL;
	}
	if (!p->reupload && !p->grid_dirty)
		return 0;
	p->grid_dirty = 0;
	return mc33hip_upload_rows(p->ctx, (const void *const *const *)p->grid->F);
}

/* Extension (not in the reference, which reads G->F anew on every call, MC:1792, 1832-1868): tells the extractor that
 * samples of the grid it was created from have been rewritten.  The next size_of_isosurface / calculate_isosurface(s)
 * uploads G->F again first.  The _GRD and its rows must still be alive then, as for the reference's own M->F. */
void MC33_grid_changed(MC33 *M) {
	mc33_private *p = priv(M);
	if (p)
		p->grid_dirty = 1;
}

unsigned long long size_of_isosurface(MC33 *M, MC33_real iso, unsigned int *nV, unsigned int *nT) {
	mc33_private *p = priv(M);
	mc33hip_counts cnt;
	mc33hip_range r;
	memset(&cnt, 0, sizeof cnt);
	if (p) {
		r.z_begin = 0; r.z_end = M->nz; r.ghost_below = 0; r.id_base = 0;
		M->iso = iso;
		if (refresh_grid(p) != MC33HIP_OK || mc33hip_count(p->ctx, iso, &r, &cnt) != MC33HIP_OK)
			memset(&cnt, 0, sizeof cnt);
	}
	if (nV) *nV = (unsigned int)cnt.nV;
	if (nT) *nT = (unsigned int)cnt.nT;
	/* MC:1939 */
	return cnt.nV * (6 * sizeof(MC33_real) + sizeof(int)) + cnt.nT * (3 * sizeof(int)) + sizeof(surface);
}

static int ensure_staging(mc33_private *p, struct staging *g, unsigned long long nV, unsigned long long nT) {
	if (g->capV < nV) {
		if (g->dV) mc33hip_device_free(p->ctx, g->dV);
		if (g->dN) mc33hip_device_free(p->ctx, g->dN);
		g->dV = g->dN = 0; g->capV = 0;
		unsigned long long cap = nV + nV / 8 + 1024;
		if (mc33hip_device_alloc(p->ctx, &g->dV, cap * 3 * sizeof(MC33_real)) != MC33HIP_OK) return -1;
		if (mc33hip_device_alloc(p->ctx, &g->dN, cap * 12) != MC33HIP_OK) return -1;
		g->capV = cap;
	}
	if (g->capT < nT) {
		if (g->dT) mc33hip_device_free(p->ctx, g->dT);
		g->dT = 0; g->capT = 0;
		unsigned long long cap = nT + nT / 8 + 1024;
		if (mc33hip_device_alloc(p->ctx, &g->dT, cap * 12) != MC33HIP_OK) return -1;
		g->capT = cap;
	}
	return 0;
}

/* GPU part of calculate_isosurface: the surface of `iso` into staging set g (device memory), its sizes into *cnt */
static int extract_to_staging(mc33_private *p, struct staging *g, MC33_real iso, mc33hip_counts *cnt) {
	mc33hip_range r;
	r.z_begin = 0; r.z_end = p->pub.nz; r.ghost_below = 0; r.id_base = 0;
	memset(cnt, 0, sizeof *cnt);
	int rc = refresh_grid(p);
	if (rc != MC33HIP_OK)
		return rc;
	/* one pass with the staging buffers of an earlier call; if they are too small the counts come back
	 * anyway, the buffers grow and only the emit pass is repeated */
	rc = mc33hip_extract(p->ctx, iso, &r, g->dV, g->dN, g->dT, g->capV, g->capT, cnt);
	if (rc == MC33HIP_ECAPACITY) {
		rc = ensure_staging(p, g, cnt->nV, cnt->nT) ? MC33HIP_ENOMEM : mc33hip_emit(p->ctx, g->dV, g->dN, g->dT, g->capV, g->capT);
		/* mc33hip_emit only enqueues: the set must be complete before anybody reads it - the helper thread of
		 * calculate_isosurfaces copies on a stream of its own, which is not ordered after this one */
		if (rc == MC33HIP_OK)
			rc = mc33hip_synchronize(p->ctx);
	}
	return rc;
}

/* One array of a caller-owned surface: plain free() releases it (MC:84-92).  Large ones start on a 2 MiB boundary and
 * ask for transparent huge pages: a 1024^3 surface is 200 MB, and touching it for the first time in 4 KiB pages (the
 * copy from the GPU, the colour fill, the munmap in free) cost more than the extraction itself.
 *
 * free_surface_memory may keep large blocks for the next surface instead of releasing them (pages that are already
 * mapped and already known to the GPU driver make calculate_isosurface + free_surface_memory at 1024^3 about three
 * times faster than fresh memory).  What it keeps is bounded by MC33_HOST_CACHE_MB: default 64 (one array of a
 * 512^3-class surface; a drop-in library should not sit on a gigabyte after `free`), 0 = keep nothing; callers that
 * extract large surfaces in a loop set it to a few times the surface size.
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
static struct {
	void *p;
	size_t cap;
} g_cache[CACHE_SLOTS], g_live[LIVE_SLOTS]; /* kept for reuse / handed out and still with the caller */
static size_t g_cache_bytes;
static pthread_mutex_t g_cache_lock = PTHREAD_MUTEX_INITIALIZER;

static size_t cache_limit(void) {
	const char *e = getenv("MC33_HOST_CACHE_MB");
	return (size_t)(e ? strtoull(e, 0, 10) : 64ull) << 20;
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

/* the counterpart used by free_surface_memory: keep a large block of ours for the next surface, or free it */
static void surface_block_release(void *p) {
	if (!p)
		return;
	size_t cap = 0;
	pthread_mutex_lock(&g_cache_lock);
	for (int k = 0; k != LIVE_SLOTS; k++)
		if (g_live[k].p == p) {
			cap = g_live[k].cap;
			g_live[k].p = 0;
			break;
		}
	if (cap && malloc_usable_size(p) >= cap && g_cache_bytes + cap <= cache_limit())
		for (int k = 0; k != CACHE_SLOTS; k++)
			if (!g_cache[k].p) {
				g_cache[k].p = p;
				g_cache[k].cap = cap;
				g_cache_bytes += cap;
				p = 0;
				break;
			}
	pthread_mutex_unlock(&g_cache_lock);
	free(p);
}

/* host part: a caller-owned `surface` (five malloc blocks, MC:84-92) filled from staging set g.
 * concurrent: copy on the side stream, beside whatever the context is computing */
static surface *surface_from_staging(mc33_private *p, const struct staging *g, const mc33hip_counts *cnt, MC33_real iso, int concurrent) {
	surface *S = (surface *)malloc(sizeof(surface));
	if (!S)
		return 0;
	if (!cnt->nV) { /* MC:1880-1883 */
		memset(S, 0, sizeof(surface));
		return S;
	}
	const size_t nV = (size_t)cnt->nV, nT = (size_t)cnt->nT;
	size_t capV = 0, capN = 0, capT = 0, capC = 0;
	S->V = (MC33_real(*)[3])surface_block(nV * 3 * sizeof(MC33_real), &capV);
	S->N = (float(*)[3])surface_block(nV * 3 * sizeof(float), &capN);
	S->T = (unsigned int(*)[3])surface_block((nT ? nT : 1) * 3 * sizeof(int), &capT);
	S->color = (int *)surface_block(nV * sizeof(int), &capC);
	void *const dst[3] = {S->V, S->N, S->T};
	const void *const src[3] = {g->dV, g->dN, g->dT};
	const size_t bytes[3] = {nV * 3 * sizeof(MC33_real), nV * 12, nT * 12};
	if (!S->V || !S->N || !S->T || !S->color || mc33hip_download_many(p->ctx, 3, dst, src, bytes, concurrent) != MC33HIP_OK) {
		surface_block_release(S->V); surface_block_release(S->N); surface_block_release(S->T); surface_block_release(S->color);
		free(S);
		return 0;
	}
	const int col = DefaultColorMC;
	for (size_t k = 0; k != nV; k++)
		S->color[k] = col;
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

surface *calculate_isosurface(MC33 *M, MC33_real iso) {
	mc33_private *p = priv(M);
	if (!p)
		return 0;
	M->nT = M->nV = 0;
	M->memoryfault = 0;
	M->iso = iso;
	mc33hip_counts cnt;
	surface *S = extract_to_staging(p, &p->set[0], iso, &cnt) == MC33HIP_OK ? surface_from_staging(p, &p->set[0], &cnt, iso, 0) : 0;
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
				(void)mc33hip_sweep_many(p->ctx, many, (int)m, &r); /* (on failure the single calls sweep for themselves) */
		}
		/* surface k is computed into set k&1 while the helper thread copies surface k-1 out of the other set */
		struct staging *g = &p->set[k & 1];
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
		surface_block_release(S->T); surface_block_release(S->V); surface_block_release(S->N); surface_block_release(S->color);
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
