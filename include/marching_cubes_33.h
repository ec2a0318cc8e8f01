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

#if defined(INTEGER_GRD)
typedef float MC33_real;
#  if GRD_TYPE_SIZE == 4
typedef unsigned int GRD_data_type;
#  elif GRD_TYPE_SIZE == 2
typedef unsigned short int GRD_data_type;
#  elif GRD_TYPE_SIZE == 1
typedef unsigned char GRD_data_type;
#  else
#    error "INTEGER_GRD needs GRD_TYPE_SIZE 1, 2 or 4"
#  endif
#elif defined(GRD_TYPE_SIZE) && GRD_TYPE_SIZE == 8
typedef double GRD_data_type;
typedef double MC33_real;
#else
typedef float GRD_data_type;
typedef float MC33_real;
#  undef GRD_TYPE_SIZE
#  define GRD_TYPE_SIZE 4
#endif


#ifdef __cplusplus
extern "C" {
#endif

/* Regular grid of samples F[k][j][i] (k: z, j: y, i: x), N[] intervals per axis, so N[]+1 points.
 * Layout identical to the reference's _GRD (reference header :111-124), 416 bytes. */
typedef struct {
	GRD_data_type ***F;      /* row pointers; rows may be separate allocations                      */
	unsigned int N[3];       /* intervals in x, y, z                                               */
	double r0[3], d[3];      /* origin, spacing                                                    */
	float L[3];              /* extent (unused by the isosurface path)                             */
#ifndef GRD_ORTHOGONAL
	float Ang[3];
	int nonortho;            /* inclined grid: positions / normals go through _A / A_ (MC33_spnC)  */
	double _A[3][3], A_[3][3]; /* fractional -> cartesian cell matrix (unit edges) and its inverse  */
#endif
	int periodic;
	int internal_data;       /* 1: rows were allocated by alloc_F and are freed by free_memory_grd */
	char title[160];
} _GRD;

/* Result of calculate_isosurface (reference header :133-152), 64 bytes.  T, V, N, color and the
 * struct itself are five separate malloc blocks owned by the caller (free_surface_memory). */
typedef struct {
	unsigned int (*T)[3];    /* triangles: three vertex indices each   */
	MC33_real (*V)[3];       /* vertex positions                        */
	float (*N)[3];           /* unit normals                            */
	int *color;              /* one 0xAABBGGRR colour per vertex        */
	unsigned int nV, nT;
	unsigned int capt, capv; /* allocated triangles / vertices          */
	MC33_real iso;
	union {
		void *p;
		long long ul;
		int i[2];
		short si[4];
		char c[8];
		float f[2];
		double df;
	} user;                  /* free for the caller                     */
} surface;

/* Extraction object (reference header :154-179), 304-byte public prefix.  This library allocates a
 * larger private object whose first member is this struct; Dx..Lz are unused (NULL) and the GPU
 * context hangs behind the public part. */
typedef struct {
	unsigned int (*T)[3];
	MC33_real (*V)[3];
	float (*N)[3];
	int *color;
	unsigned int nV, nT;
	unsigned int capt, capv;
	MC33_real iso;
	int memoryfault;         /* non-zero after a failed calculate_isosurface (out of memory, GPU error) */
	const GRD_data_type ***F;
	MC33_real O[3], D[3], ca, cb;
	unsigned int nx, ny, nz;
	unsigned int (*store)(void *, MC33_real *);
#ifndef GRD_ORTHOGONAL
	double _A[3][3], A_[3][3];
#endif
	unsigned int **Dx, **Dy, **Ux, **Uy, **Lz;
} MC33;

extern int DefaultColorMC;   /* colour given to every vertex, 0xAABBGGRR (reference header :181) */

/* c = A b (t == 0) or A^T b (t != 0) for a 3x3 matrix, used for inclined grids (reference header :186-191).
 * _multTSA_bf assumes an upper triangular A.  A caller may point mult_Abf at either; calculate_isosurface
 * looks at the pointer when it is called and runs the matching form on the GPU (any other function: NULL). */
void _multTSA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int t);
void _multA_bf(const double (*A)[3], MC33_real *b, MC33_real *c, int t);
extern void (*mult_Abf)(const double (*)[3], MC33_real *, MC33_real *, int);

/* --- isosurface path (reference header :228-258) ------------------------------------------------ */
MC33 *create_MC33(_GRD *G);                                  /* uploads the grid to HBM once        */
surface *calculate_isosurface(MC33 *M, MC33_real iso);       /* GPU extraction; NULL on failure     */
unsigned long long size_of_isosurface(MC33 *M, MC33_real iso, unsigned int *nV, unsigned int *nT);
void free_MC33(MC33 *M);
void free_surface_memory(surface *S);
void adjustvectorlenght_s(surface *S);

/* --- extension (not in the reference): several isovalues of the grid that is resident in HBM ------- */
/* out[k] = what calculate_isosurface(M, iso[k]) would return (NULL where it failed; caller frees each with
 * free_surface_memory).  The device-to-host copy of surface k runs beside the extraction of surface k+1.
 * Returns the number of surfaces produced. */
unsigned int calculate_isosurfaces(MC33 *M, const MC33_real *iso, unsigned int n, surface **out);

/* --- surface files (reference header :193-222), host C: csrc/mc33_surface_io.c -------------------- */
int write_bin_s(surface *S, const char *filename);   /* ".sup" binary container; 0 on success, -1 on failure */
surface *read_bin_s(const char *filename);           /* NULL on failure; also reads ".sud" (double) files      */
int write_txt_s(surface *S, const char *filename);
int write_obj_s(surface *S, const char *filename);   /* Wavefront OBJ with per-vertex normals                  */
int write_ply_s(surface *S, const char *filename, const char *author, const char *object); /* ASCII PLY       */

/* --- grid container helpers (reference header :263-329), host C ---------------------------------- */
void free_memory_grd(_GRD *Z);
int alloc_F(_GRD *Z);
_GRD *grid_from_data_pointer(unsigned int Nx, unsigned int Ny, unsigned int Nz, GRD_data_type *data);
_GRD *generate_grid_from_fn(double x_initial, double y_initial, double z_initial,
                            double x_final, double y_final, double z_final,
                            double x_step, double y_step, double z_step,
                            double (*fn)(double x, double y, double z));


/* --- grid file readers (reference header :263-311), host C: csrc/mc33_grid_io.c; NULL on failure --------- */
_GRD *read_grd(const char *filename);                  /* DMol .grd text file (may describe an inclined cell)  */
_GRD *read_grd_binary(const char *filename);           /* the library's own binary container ("_GRD")          */
_GRD *read_scanfiles(const char *filename, unsigned int res, int order); /* numbered res x res u16 slices       */
_GRD *read_raw_file(const char *filename, unsigned int *N, int byte, int isfloat); /* bare samples, N = points  */
_GRD *read_dat_file(const char *filename);             /* u16 nx, ny, nz header + u16 samples                  */

#ifdef __cplusplus
}
#endif
#endif /* marching_cubes_33_h */
