/* mc33_hip.h -- device-level C ABI of the MI355X MC33 extractor (plain pointers and sizes only).
 *
 * These are the entry points the reference-compatible host layer (mc33_capi.c: create_MC33,
 * calculate_isosurface, size_of_isosurface, free_MC33 - reference include/marching_cubes_33.h:228-258)
 * is built on, and what a foreign-language binding would bind.  Each one names the reference code it
 * replaces ("MC:" = reference source/marching_cubes_33.c).
 *
 * All functions return 0 on success, a negative MC33HIP_E* code otherwise (no exceptions, no exit;
 * the reference's only failure convention is a NULL return, MC:1884-1887).
 */
#ifndef MC33_HIP_H
#define MC33_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MC33HIP_OK          0
#define MC33HIP_EINVAL     -1  /* bad argument                                   */
#define MC33HIP_ENOGPU     -2  /* no HIP device / runtime error at start-up       */
#define MC33HIP_ENOMEM     -3  /* host or device allocation failed                */
#define MC33HIP_ECAPACITY  -4  /* caller's output buffers are too small           */
#define MC33HIP_ERUNTIME   -5  /* a HIP call or kernel failed (see last_error)    */
#define MC33HIP_EOVERFLOW  -6  /* more than 2^32-1 vertices or triangles          */

typedef struct mc33hip_ctx mc33hip_ctx;

/* Geometry of the (slab of the) grid one context works on.  Replaces the part of create_MC33 that
 * snapshots _GRD (MC:1758-1782).  sample_bytes must match the library variant: libMC33_f32 4 (float),
 * libMC33_u8 1, libMC33_u16 2, libMC33_u32 4 (unsigned int), libMC33_f64 8 (double). */
typedef struct {
	unsigned int npx, npy;      /* points per row / rows per plane                                       */
	unsigned int npz_resident;  /* planes resident in this context                                        */
	unsigned int plane0;        /* global z index of the first resident plane (0 for a whole grid)        */
	unsigned int nz_total;      /* cell slices of the WHOLE grid (points along z - 1)                     */
	double r0[3], d[3];         /* origin and spacing of the WHOLE grid (_GRD.r0, _GRD.d)                 */
	int sample_bytes;
	int device;                 /* HIP device ordinal, -1: current device                                 */
} mc33hip_grid_desc;

/* Work range of one extraction: cell slices [z_begin, z_end) are emitted; when ghost_below is non-zero
 * slice z_begin-1 is classified too, so that ids of vertices on the slab interface - which belong to
 * the rank below - can be resolved.  Single GPU: {0, nz_total, 0, 0}. */
typedef struct {
	unsigned int z_begin, z_end;
	unsigned int ghost_below;
	unsigned int id_base;       /* number of vertices created by all slices below z_begin (global numbering) */
} mc33hip_range;

typedef struct {
	unsigned long long nV, nT;            /* vertices / triangles of slices [z_begin, z_end)                */
	unsigned long long nV_ghost, nT_ghost;/* ... of the ghost slice (0 without ghost)                       */
	unsigned long long active_cells;      /* cells cut by the surface in the classified slices             */
} mc33hip_counts;

typedef struct {
	float sweep_ms, scan_ms, emit_ms;     /* hipEvent time of each pass of the last extraction, on its stream */
	float total_ms;
	unsigned int sweep_launches;          /* >1 if the work-record buffer had to grow and the sweep re-ran     */
} mc33hip_timing;

int mc33hip_create(mc33hip_ctx **out, const mc33hip_grid_desc *desc);
void mc33hip_destroy(mc33hip_ctx *c);
const char *mc33hip_last_error(void);

/* Grid upload: replaces "M->F = G->F" (MC:1792) - the reference re-reads host memory on every call,
 * this library keeps a pitched copy in HBM.
 *   upload_rows : F[k][j] row pointers exactly as _GRD.F holds them (rows may be separate mallocs,
 *                 reference MC33_util_grd.c:147-169), k counts resident planes
 *   adopt_device: use a caller-owned device buffer in place (no copy); pitch/slice in samples      */
int mc33hip_upload_rows(mc33hip_ctx *c, const void *const *const *F);
int mc33hip_upload_contiguous(mc33hip_ctx *c, const void *host_samples);
int mc33hip_adopt_device(mc33hip_ctx *c, const void *device_samples, size_t pitch, size_t slice);

/* Inclined (non-orthogonal) grid: vertices and normals go through the cell matrices like MC33_spnC does
 * (reference marching_cubes_33.c:587-621).  grd_A / grd_Ai = _GRD._A / _GRD.A_ (3x3 row major, before the
 * scaling by d that create_MC33 applies, MC:1763-1770); triangular != 0 = the caller's mult_Abf is
 * _multTSA_bf (MC33_util_grd.c:86-97).  NULL matrices switch back to the orthogonal stores. */
int mc33hip_set_inclined(mc33hip_ctx *c, const double *grd_A, const double *grd_Ai, int triangular);

/* Orientation: on != 0 negates the normals and exchanges the first two indices of every triangle - what the reference
 * does when it is compiled with MC33_NORMAL_NEG 1 (reference source/libMC33.c:20-22, marching_cubes_33.c:509-513,
 * 1246-1250).  The libMC33_<type>_nneg.so flavour of the host layer switches it on in create_MC33. */
int mc33hip_set_normal_neg(mc33hip_ctx *c, int on);

/* Stream all work is enqueued on (a hipStream_t passed as void*; NULL = the default stream). */
int mc33hip_set_stream(mc33hip_ctx *c, void *hip_stream);
/* ... or a non-blocking stream of the context's own, for callers without a HIP runtime of their own that drive several contexts
 * side by side (create_MC33 with MC33_HIP_DEVICES: one context per z-slab). */
int mc33hip_own_stream(mc33hip_ctx *c);
/* HIP devices visible to this process (0: none, or no runtime). */
int mc33hip_device_count(void);

/* Count pass only: classification + prefix sums; the device twin of size_of_isosurface (MC:1892-1940).
 * Synchronises the stream.
 * Over the library's own copy of the grid (mc33hip_upload*), a count that has not served an emit yet is reused by the next
 * mc33hip_count of the same isovalue and range: size_of_isosurface followed by calculate_isosurface of that value streams the
 * volume once.  Used once only - an extraction repeated with the same isovalue does all of its work again - and never with a
 * buffer of the caller's (mc33hip_adopt_device), whose samples may change without this library knowing. */
int mc33hip_count(mc33hip_ctx *c, double iso, const mc33hip_range *range, mc33hip_counts *out);

/* Iso sweep over the resident grid (BASELINE.json configs[4]; the caller of calculate_isosurfaces): classifies the
 * samples against n <= 8 isovalues while streaming the volume ONCE per 4 isovalues, instead of once per isovalue as n
 * separate calls do (the reference re-reads all of F for every isovalue, MC:1832-1868).  Nothing is returned: the
 * mc33hip_count / mc33hip_extract calls that follow with one of these isovalues and the same range find the sweep
 * already made and only run the passes after it.  A sweep made ahead is used once and is dropped when the grid changes.
 * Asynchronous on the context's stream. */
int mc33hip_sweep_many(mc33hip_ctx *c, const double *isos, int n, const mc33hip_range *range);

/* The same, and everything else a count needs as well - record ranges, cell records, prefix sums - right behind each pass
 * over the grid, one launch of every kernel for the 4 isovalues of the pass, each isovalue into buffers of its own.  The
 * mc33hip_count calls that follow only fetch counters, mc33hip_emit / mc33hip_extract only emit - in any order, any number
 * of times, until the next mc33hip_prepare_many / mc33hip_sweep_many or a change of the grid.  For callers that need the
 * counts of ALL isovalues before the first emit: z-slabs over several GPUs exchange them in ONE collective per step and
 * then emit every isovalue at its global id base (mc33_c_library_amd/slabs.py: extract_slab_many).  On one GPU
 * mc33hip_sweep_many + mc33hip_extract per isovalue is faster: an emit right behind its own tail finds the records in the
 * last-level cache.  Falls back to mc33hip_sweep_many when device memory does not allow the buffers.
 * At most 8 isovalues per call (MC33HIP_EINVAL beyond).  A mc33hip_count / mc33hip_extract of an isovalue or range that was
 * NOT prepared works in the buffers of prepared isovalue #0 and drops it: its next count sweeps the grid again (same
 * results, one more pass over the volume). */
int mc33hip_prepare_many(mc33hip_ctx *c, const double *isos, int n, const mc33hip_range *range);

/* Global number of the first vertex of the range last counted (z-slab decomposition: known only after
 * the ranks have exchanged their counts).  Takes effect in the next mc33hip_emit. */
int mc33hip_set_id_base(mc33hip_ctx *c, unsigned int id_base);

/* Emit pass for the range last counted, into caller-owned DEVICE buffers:
 * V: capV x 3 MC33_real (float; double in libMC33_f64), N: capV x 3 floats; T: capT x 3 unsigned.  The isovalue
 * is passed as a double and used as MC33_real.  Replaces the vertex/triangle appends of MC33_findCase
 * (MC:780-1252).  Asynchronous on the context's stream. */
int mc33hip_emit(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT);

/* A z-slab's count, count exchange and emit with no host round trip in between (the reference has no such
 * step - its one process numbers the vertices as it sweeps, MC:1783-1808).  mc33hip_count_async enqueues what mc33hip_count
 * does and returns; mc33hip_counts_to_device leaves {vertices, triangles} of the counted range as two 64-bit integers in DEVICE
 * memory (for a collective to gather: torch.distributed.all_gather_into_tensor over RCCL); mc33hip_bases_from_table turns the
 * gathered table - rank r's pair at device_table[r * stride] - into this rank's vertex id base and, with concatenated != 0, the
 * rows at which it writes into output arrays shared by all ranks (0: arrays of its own); mc33hip_emit_at_device_bases is
 * mc33hip_emit with those words read on the device (capacities are checked there too).  All four only enqueue on the context's
 * stream.  mc33hip_count_finish waits and brings the counts to the host: MC33HIP_ECAPACITY when the work records did not fit
 * (room has been made: repeat the step with mc33hip_count) or the outputs were too small. */
int mc33hip_count_async(mc33hip_ctx *c, double iso, const mc33hip_range *range);
int mc33hip_counts_to_device(mc33hip_ctx *c, long long *device_dst);
int mc33hip_bases_from_table(mc33hip_ctx *c, const long long *device_table, int stride, int rank, int concatenated);
int mc33hip_emit_at_device_bases(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT);
int mc33hip_count_finish(mc33hip_ctx *c, mc33hip_counts *out);

/* mc33hip_emit with the copy to the HOST pipelined behind it (what calculate_isosurface does, reference MC:1869-1879:
 * the surface in the caller's malloc blocks): hT (nT x 3 unsigned) is copied on a stream of the context's own as soon as the
 * passes that write T have been through, hV (nV x 3 MC33_real) and hN (nV x 3 floats) as soon as theirs have - the copies of
 * one array run beside the kernels of the other.  nV, nT are those of the mc33hip_count before.  Only enqueues;
 * mc33hip_download_wait returns when all three arrays are in host memory. */
int mc33hip_emit_download(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT,
                          void *hV, void *hN, void *hT);
int mc33hip_download_wait(mc33hip_ctx *c);

/* Whole extraction (count + emit) with ONE synchronisation at the end; fails with MC33HIP_ECAPACITY
 * (and reports the needed sizes in *out) when the buffers are too small.  This is the path
 * calculate_isosurface (MC:1816-1889) and bench.py use. */
int mc33hip_extract(mc33hip_ctx *c, double iso, const mc33hip_range *range, void *dV, void *dN, void *dT,
                    unsigned long long capV, unsigned long long capT, mc33hip_counts *out);

/* hipEvent times of the last extraction; all zero unless the context was created with MC33_HIP_TIMING=1 (whole call)
 * or 2 (per pass) in the environment - the event records cost about 20 us per call.  Waits for a pending mc33hip_emit. */
int mc33hip_last_timing(mc33hip_ctx *c, mc33hip_timing *t);

/* Switches the hipEvent timing of the context at run time (0, 1, 2 as MC33_HIP_TIMING, which only sets the level a context
 * starts with).  bench.py times its steps at level 2 - the events are recorded, never waited for - and once more at 0. */
int mc33hip_set_timing(mc33hip_ctx *c, int level);

/* Measurement aid (bench.py's `roofline.read_ceiling`): a plain read-only kernel over the resident grid -
 * every 16-byte chunk once (every block a contiguous piece, 16 loads in flight per lane: the shape that reads fastest on this part),
 * nothing written - timed with hipEvents on the context's stream, reps launches after a warm-up.
 * *bytes / *ms_best is what a read stream reaches on this device, in this process, on this very buffer: the ceiling the
 * sweep (MC:1832-1868, which reads every sample once) is set beside.  Not part of any extraction. */
int mc33hip_probe_read(mc33hip_ctx *c, int reps, float *ms_best, float *ms_median, unsigned long long *bytes);

/* Waits until everything enqueued on the context's stream (mc33hip_emit in particular) has finished.  Needed before
 * the output buffers are read by anything that is not ordered after that stream - mc33hip_download_concurrent, another
 * stream, another process. */
int mc33hip_synchronize(mc33hip_ctx *c);

/* Device-to-host copy helper for callers without a HIP runtime of their own (blocking). */
int mc33hip_download(mc33hip_ctx *c, void *host_dst, const void *device_src, size_t bytes);
/* The same on a stream of the context's own that neither waits for nor delays the work queued by the other entry
 * points: the download of one result can run while the next extraction is computed (calculate_isosurfaces does
 * that from a helper thread).  The source must not be written meanwhile.  Blocking; callable from any thread. */
int mc33hip_download_concurrent(mc33hip_ctx *c, void *host_dst, const void *device_src, size_t bytes);
/* Several copies, one wait at the end (a surface is three arrays): concurrent = 0 orders them after the context's
 * stream like mc33hip_download, != 0 uses the side stream like mc33hip_download_concurrent. */
int mc33hip_download_many(mc33hip_ctx *c, int n, void *const *host_dst, const void *const *device_src, const size_t *bytes,
                          int concurrent);
/* Plain device allocations on the context's device (for language bindings). */
int mc33hip_device_alloc(mc33hip_ctx *c, void **dptr, size_t bytes);
int mc33hip_device_free(mc33hip_ctx *c, void *dptr);

#ifdef __cplusplus
}
#endif
#endif
