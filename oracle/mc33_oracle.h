/* mc33_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's calculate_isosurface path
 * (reference source/marching_cubes_33.c:1816-1889 sweep, :673-1253 MC33_findCase, :347-462
 * ambiguity tests, :485-585 vertex stores, :628-649 MC33_surfint).  It exists to CHECK the HIP
 * path; nothing in the product library may include, link or call it.  Pinned against the real
 * reference built by oracle/Makefile (oracle/_ref) in tests/test_oracle_vs_reference.py and
 * against the committed fixtures under tests/golden/.
 */
#ifndef MC33_ORACLE_H
#define MC33_ORACLE_H
#include <stdint.h>

#if defined(MC33_ORACLE_U16)
typedef uint16_t mc33o_sample; /* INTEGER_GRD, GRD_TYPE_SIZE 2 (marching_cubes_33.h:66-75) */
#elif defined(MC33_ORACLE_U8)
typedef uint8_t mc33o_sample;  /* INTEGER_GRD, GRD_TYPE_SIZE 1 (marching_cubes_33.h:75-76) */
#elif defined(MC33_ORACLE_U32)
typedef uint32_t mc33o_sample; /* INTEGER_GRD, GRD_TYPE_SIZE 4 (marching_cubes_33.h:68-72) */
#elif defined(MC33_ORACLE_F64)
typedef double mc33o_sample;   /* GRD_TYPE_SIZE 8: double grid AND double arithmetic (marching_cubes_33.h:80-82) */
#else
typedef float mc33o_sample;    /* default float grid (marching_cubes_33.h:84-85) */
#endif
#ifdef MC33_ORACLE_F64
typedef double mc33o_real;     /* MC33_real: corner values, interpolation, vertex positions, isovalue */
#else
typedef float mc33o_real;
#endif

typedef struct {
	uint32_t nV, nT;
	mc33o_real *V; /* nV x 3 */
	float *N;    /* nV x 3 */
	uint32_t *T; /* nT x 3 */
} mc33o_surface;

/* data: contiguous samples, x fastest, then y, then z; np* = POINT counts per axis.
 * r0/d: origin and spacing (double, as in _GRD).  Returns 0 on success, -1 on allocation failure.
 * exact_rsqrt is always used for normals (1.0f/sqrtf, marching_cubes_33.c:70-73). */
int mc33o_calculate_isosurface(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz,
                               const double r0[3], const double d[3], mc33o_real iso, mc33o_surface *out);
/* Inclined grid (_GRD.nonortho, MC33_spnC MC:587-621): grd_A / grd_Ai are _GRD._A / _GRD.A_ (3x3, row major);
 * triangular != 0 selects the _multTSA_bf form of mult_Abf (MC33_util_grd.c:86-97).  NULL matrices = the call above. */
int mc33o_calculate_isosurface_inclined(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz,
                                        const double r0[3], const double d[3], const double *grd_A, const double *grd_Ai,
                                        int triangular, mc33o_real iso, mc33o_surface *out);
void mc33o_free_surface(mc33o_surface *s);

/* Per-cell classification only (no geometry): for every cell writes the 8-bit sign index and the
 * offset of the chosen triangle pattern inside the table (0 for inactive cells).  Arrays have
 * (npx-1)*(npy-1)*(npz-1) entries, x fastest. */
int mc33o_classify(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz, mc33o_real iso,
                   uint8_t *index_out, uint16_t *pattern_out);

/* FNV-1a 64-bit over raw bytes (golden hashes). */
uint64_t mc33o_fnv1a64(const void *p, uint64_t nbytes);

/* Fill helpers shared by fixtures: cos x + cos y + cos z sampled the way generate_grid_from_fn
 * does it (accumulated coordinates, MC33_util_grd.c:660-672), libm cos. */
void mc33o_fill_cos_field(float *data, uint32_t n, double lo, double h);
#endif
