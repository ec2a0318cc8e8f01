/* marching_cubes_33.h -- public C API of the MI355X-native MC33 library (libMC33_{f32,f64,u8,u16,u32}.so).
 *
 * Binary- and source-compatible with the header of dvega68/MC33_c_library (reference
 * include/marching_cubes_33.h): same type names, same struct layouts (checked by static asserts in
 * mc33_capi.c against the offsets recorded in SURVEY.md 8(a)-11), same function names and argument
 * meaning.  A program written against the reference header links against this library unchanged;
 * calculate_isosurface runs on the GPU (see include/mc33_hip.h for the device-level entry points).
 *
 * Compile-time variants, as in the reference (reference header :57-88):
 *   default                         GRD_data_type = float,          MC33_real = float
 *   -DINTEGER_GRD -DGRD_TYPE_SIZE=1 GRD_data_type = unsigned char,  MC33_real = float
 *   -DINTEGER_GRD -DGRD_TYPE_SIZE=2 GRD_data_type = unsigned short, MC33_real = float
 *   -DINTEGER_GRD -DGRD_TYPE_SIZE=4 GRD_data_type = unsigned int,   MC33_real = float
 *   -DGRD_TYPE_SIZE=8               GRD_data_type = double,         MC33_real = double (vertices, isovalue)
 *   -DGRD_ORTHOGONAL (with any of the above): _GRD and MC33 without the inclined-grid members (reference
 *   header :116-120, :173-175); link the libMC33_<type>_ortho.so flavour.
 * One library per variant, like the reference's one-type-per-compile model.
 */
#ifndef marching_cubes_33_h
#define marching_cubes_33_h

#define MC33C_VERSION_MAJOR 5
#define MC33C_VERSION_MINOR 5

/* ---- sample type (GRD_data_type) and arithmetic / vertex type (MC33_real) of this build ------------------- */
#ifdef INTEGER_GRD
#  if !defined(GRD_TYPE_SIZE) || (GRD_TYPE_SIZE != 1 && GRD_TYPE_SIZE != 2 && GRD_TYPE_SIZE != 4)
#    error "INTEGER_GRD needs GRD_TYPE_SIZE 1, 2 or 4"
#  endif
#  if GRD_TYPE_SIZE == 1
#    define MC33_SAMPLE_C_TYPE unsigned char
#  elif GRD_TYPE_SIZE == 2
#    define MC33_SAMPLE_C_TYPE unsigned short
#  else
#    define MC33_SAMPLE_C_TYPE unsigned
#  endif
#  define MC33_REAL_C_TYPE float
#elif defined(GRD_TYPE_SIZE) && GRD_TYPE_SIZE == 8
#  define MC33_SAMPLE_C_TYPE double
#  define MC33_REAL_C_TYPE double
#else
#  undef GRD_TYPE_SIZE
#  define GRD_TYPE_SIZE 4
#  define MC33_SAMPLE_C_TYPE float
#  define MC33_REAL_C_TYPE float
#endif
typedef MC33_SAMPLE_C_TYPE GRD_data_type;
typedef MC33_REAL_C_TYPE MC33_real;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- _GRD: regular grid of samples F[k][j][i] (k: z, j: y, i: x); N[] intervals per axis, N[]+1 points -----
 * Byte layout of the reference's _GRD (reference header :111-124): 416 bytes, 256 with GRD_ORTHOGONAL. */
typedef struct mc33_grid {
	GRD_data_type ***F;        /* plane -> row -> samples; rows may be separate allocations              */
	unsigned N[3];             /* intervals in x, y, z                                                  */
	double r0[3];              /* origin                                                                */
	double d[3];               /* spacing                                                               */
	float L[3];                /* extent (unused by the isosurface path)                                */
#ifndef GRD_ORTHOGONAL
	float Ang[3];              /* cell angles in degrees                                                */
	int nonortho;              /* inclined grid: positions / normals go through _A / A_ (MC33_spnC)     */
	double _A[3][3];           /* fractional -> cartesian cell matrix (unit edges) ...                  */
	double A_[3][3];           /* ... and the matrix applied (transposed) to the gradients              */
#endif
	int periodic;
	int internal_data;         /* 1: rows were allocated by alloc_F and are freed by free_memory_grd    */
	char title[160];
} _GRD;

/* ---- surface: result of calculate_isosurface (reference header :133-152), 64 bytes ------------------------
 * T, V, N, color and the struct itself are five separate malloc blocks owned by the caller
 * (free_surface_memory).  MC33 starts with the same members. */
typedef unsigned mc33_triangle[3];
typedef MC33_real mc33_position[3];
typedef float mc33_normal[3];
#define MC33_SURFACE_MEMBERS                                                                     \
	mc33_triangle *T;  /* triangles: three vertex indices each                                */ \
	mc33_position *V;  /* vertex positions                                                    */ \
	mc33_normal *N;    /* unit normals                                                        */ \
	int *color;        /* one 0xAABBGGRR colour per vertex                                    */ \
	unsigned nV;                                                                                 \
	unsigned nT;                                                                                 \
	unsigned capt;     /* allocated triangles                                                 */ \
	unsigned capv;     /* allocated vertices                                                  */ \
	MC33_real iso;

typedef union mc33_user_data { /* 8 bytes free for the caller */
	void *p;
	long long ul;
	int i[2];
	short si[4];
	char c[8];
	float f[2];
	double df;
} mc33_user_data;

typedef struct mc33_surface {
	MC33_SURFACE_MEMBERS
	mc33_user_data user;
} surface;

/* ---- MC33: extraction object (reference header :154-179); 304-byte public part (344 double build; 144 less
 * with GRD_ORTHOGONAL).  This library allocates a larger private object whose first member is this struct;
 * the five id caches of the reference's sweep are unused (NULL), the GPU context hangs behind the public part. */
typedef struct mc33_extractor {
	MC33_SURFACE_MEMBERS
	int memoryfault;           /* non-zero after a failed calculate_isosurface (out of memory, GPU error) */
	const GRD_data_type ***F;
	MC33_real O[3];            /* origin, spacing as MC33_real                                           */
	MC33_real D[3];
	MC33_real ca;              /* d[2]/d[0], d[2]/d[1] (anisotropic spacing)                             */
	MC33_real cb;
	unsigned nx;
	unsigned ny;
	unsigned nz;
	unsigned (*store)(void *, MC33_real *);
#ifndef GRD_ORTHOGONAL
	double _A[3][3];           /* cell matrices scaled by the spacing (inclined grids)                   */
	double A_[3][3];
#endif
	unsigned **Dx;
	unsigned **Dy;
	unsigned **Ux;
	unsigned **Uy;
	unsigned **Lz;
} MC33;

/* colour given to every vertex, 0xAABBGGRR (reference header :181) */
extern int DefaultColorMC;

/* ---- isosurface path (reference header :228-258) --------------------------------------------------------- */
MC33 *create_MC33(_GRD *grid);                                   /* uploads the grid to HBM once          */
surface *calculate_isosurface(MC33 *extractor, MC33_real isovalue); /* GPU extraction; NULL on failure      */
unsigned long long size_of_isosurface(MC33 *extractor, MC33_real isovalue,
                                      unsigned *vertices, unsigned *triangles);
void free_MC33(MC33 *extractor);
void free_surface_memory(surface *s);
void adjustvectorlenght_s(surface *s);                           /* shrink the arrays to nV / nT          */

/* extension (not in the reference): several isovalues of the grid that is resident in HBM.
 * out[k] = what calculate_isosurface(M, iso[k]) would return (NULL where it failed; the caller frees each with
 * free_surface_memory).  The device-to-host copy of surface k runs beside the extraction of surface k+1.
 * Returns the number of surfaces produced. */
unsigned calculate_isosurfaces(MC33 *extractor, const MC33_real *isovalues, unsigned count, surface **out);

/* extension (not in the reference): the caller has rewritten samples of the grid `extractor` was created from.  The
 * reference reads G->F anew on every call (source/marching_cubes_33.c:1792, 1832-1868); this library keeps a copy in
 * HBM, and uploads G->F again before the next extraction after this call.  (MC33_HIP_REUPLOAD=1 in the environment does
 * that before EVERY extraction, for callers that cannot be changed.) */
void MC33_grid_changed(MC33 *extractor);

/* ---- inclined grids (reference header :186-191) ---------------------------------------------------------
 * c = A b (transposed == 0) or A^T b for a 3x3 matrix; _multTSA_bf assumes an upper triangular A.  A caller may
 * point mult_Abf at either; calculate_isosurface looks at the pointer when it is called and runs the matching
 * form on the GPU (any other function: NULL + memoryfault). */
void _multA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int transposed);
void _multTSA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int transposed);
extern void (*mult_Abf)(const double (*A)[3], MC33_real *b, MC33_real *c, int transposed);

/* ---- surface files (reference header :193-222), host C: csrc/mc33_surface_io.c --------------------------- */
int write_bin_s(surface *s, const char *path);     /* ".sup" / ".sud" binary container; 0 on success, -1 on failure */
surface *read_bin_s(const char *path);             /* NULL on failure; reads both precisions                         */
int write_txt_s(surface *s, const char *path);
int write_obj_s(surface *s, const char *path);     /* Wavefront OBJ with per-vertex normals                          */
int write_ply_s(surface *s, const char *path, const char *author, const char *object); /* ASCII PLY               */

/* ---- grid containers (reference header :263-329), host C --------------------------------------------------- */
int alloc_F(_GRD *grid);                            /* rows for N[] + 1 points per axis; 0 on success               */
void free_memory_grd(_GRD *grid);
_GRD *grid_from_data_pointer(unsigned points_x, unsigned points_y, unsigned points_z, GRD_data_type *samples);
_GRD *generate_grid_from_fn(double x0, double y0, double z0, double x1, double y1, double z1,
                            double step_x, double step_y, double step_z, double (*f)(double, double, double));

/* ---- grid files (reference header :263-311), host C: csrc/mc33_grid_io.c; NULL on failure ------------------ */
_GRD *read_grd(const char *path);                   /* DMol .grd text file (may describe an inclined cell)         */
_GRD *read_grd_binary(const char *path);            /* the library's own binary container ("_GRD")                 */
_GRD *read_scanfiles(const char *first_file, unsigned resolution, int swap_bytes); /* numbered res x res u16 slices */
_GRD *read_raw_file(const char *path, unsigned *points, int bytes_per_sample, int is_float); /* bare samples     */
_GRD *read_dat_file(const char *path);              /* u16 nx, ny, nz header + u16 samples                         */

#ifdef __cplusplus
}
#endif
#endif /* marching_cubes_33_h */
