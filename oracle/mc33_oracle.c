/* mc33_oracle.c -- TEST INFRASTRUCTURE ONLY (see mc33_oracle.h).
 *
 * Serial restatement of the reference algorithm, written table-driven instead of as one large
 * switch: the twelve cube edges, their ownership rule, the candidate lists that decide where the id
 * of a vertex lying exactly on a grid point comes from, and the gradient stencils are DATA here.
 * Each table / function cites the reference lines it restates ("MC:" = source/marching_cubes_33.c).
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA, strict evaluation order) -- see oracle/Makefile.
 */
#include "mc33_oracle.h"
#include "mc33_oracle_lut.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NOID 0xFFFFFFFFu

/* ------------------------------------------------------------------------------------------------
 * Cube geometry (MC:332-341 vertices, MC:660-668 edges).
 * corner k -> offset (dx,dy,dz) inside the cell.
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t CORNER_OFF[8][3] = {
	{0, 0, 0}, {0, 1, 0}, {0, 1, 1}, {0, 0, 1}, {1, 0, 0}, {1, 1, 0}, {1, 1, 1}, {1, 0, 1}};

/* corner index from offsets */
static const uint8_t CORNER_AT[2][2][2] = {/* [dx][dy][dz] */
	{{0, 3}, {1, 2}}, {{4, 7}, {5, 6}}};

enum { AX_X = 0, AX_Y = 1, AX_Z = 2 };

/* kinds of id sources tried when an edge end point has v == 0 (MC:788-1224) */
enum { SRC_CELL = 1,  /* another edge of this cell, if already visited (p[e] != FF)            */
       SRC_SLOT = 2,  /* id stored for a neighbouring grid edge, under a sign guard            */
       SRC_BELOW = 3  /* the z-edge one slice below ("value of previous slice"): no write-back */ };
enum { NEED_X = 1, NEED_Y = 2, NEED_Z = 4, NEED_X1 = 8, NEED_Y1 = 16, NEED_NOT_X = 32, NEED_NOT_Y = 64 };
enum { GUARD_NONE = 0, GUARD_CORNER = 1, GUARD_SAMPLE = 2 };

typedef struct {
	uint8_t kind, edge, need, guard;
	int8_t gcorner, fx, fy, fz; /* guard: corner index, or sample offset from the cell origin */
	uint8_t axis;
	int8_t sx, sy, sz;          /* grid edge whose stored id is taken (offset of its base point) */
	uint8_t also;               /* this cell edge is marked visited with the same id (or 0xFF)  */
} id_source;

#define CELL_(e)                               {SRC_CELL, e, 0, GUARD_NONE, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF}
#define CELLN_(e, need)                        {SRC_CELL, e, need, GUARD_NONE, 0, 0, 0, 0, 0, 0, 0, 0, 0xFF}
#define SLOTV_(need, c, ax, sx, sy, sz, also)  {SRC_SLOT, 0, need, GUARD_CORNER, c, 0, 0, 0, ax, sx, sy, sz, also}
#define SLOTF_(need, fx, fy, fz, ax, sx, sy, sz) {SRC_SLOT, 0, need, GUARD_SAMPLE, 0, fx, fy, fz, ax, sx, sy, sz, 0xFF}
#define BELOW_(need, fx, fy, fz, sx, sy)       {SRC_BELOW, 0, need, GUARD_SAMPLE, 0, fx, fy, fz, AX_Z, sx, sy, -1, 0xFF}

enum { OWN_X0 = 1, OWN_Y0 = 2, OWN_Z0 = 4 };

typedef struct {
	uint8_t a, b;      /* end points; t is measured from a (MC:810, 847, ... 1212)              */
	uint8_t axis;      /* direction of the edge                                                  */
	uint8_t own;       /* the cell creates the vertex only if these cell coordinates are 0,      */
	                   /* otherwise it reads the id stored by an earlier cell    */
	uint8_t nA, nB;
	id_source A[6];    /* tried in order when v[a] == 0 */
	id_source B[2];    /* tried in order when v[b] == 0 */
} edge_rule;

/* MC:787-1224, one row per edge.  Slots are written as (axis, dx, dy, dz) of the grid edge's base
 * point relative to the cell origin: Dy[y][x] = (Y,0,0,0), Uy[y][x] = (Y,0,0,1), Dx[y][x] = (X,0,0,0),
 * Ux[y][x] = (X,0,0,1), Lz[y][x] = (Z,0,0,0). */
static const edge_rule EDGE[12] = {
	/* 0: MC:787-820 */
	{0, 1, AX_Y, OWN_X0 | OWN_Z0, 5, 2,
	 {CELL_(3), CELL_(8), SLOTV_(NEED_Y, 3, AX_Z, 0, 0, 0, 0xFF), SLOTV_(NEED_Y, 4, AX_X, 0, 0, 0, 0xFF),
	  SLOTF_(NEED_Y, 0, -1, 0, AX_Y, 0, -1, 0)},
	 {CELL_(9), CELL_(1)}},
	/* 1: MC:821-859 */
	{1, 2, AX_Z, OWN_X0, 5, 2,
	 {CELL_(0), CELL_(9), SLOTV_(NEED_Z, 0, AX_Y, 0, 0, 0, 0xFF), SLOTF_(NEED_Z | NEED_Y1, 0, 2, 0, AX_Y, 0, 1, 0),
	  BELOW_(NEED_Z, 0, 1, -1, 0, 1)},
	 {CELL_(10), CELL_(2)}},
	/* 2: MC:860-895 */
	{3, 2, AX_Y, OWN_X0, 5, 2,
	 {CELL_(3), CELL_(11), SLOTV_(NEED_Y, 0, AX_Z, 0, 0, 0, 0xFF), SLOTV_(NEED_Y, 7, AX_X, 0, 0, 1, 0xFF),
	  SLOTF_(NEED_Y, 0, -1, 1, AX_Y, 0, -1, 1)},
	 {CELL_(10), CELL_(1)}},
	/* 3: MC:896-930 */
	{0, 3, AX_Z, OWN_X0 | OWN_Y0, 5, 2,
	 {CELL_(0), CELL_(8), SLOTV_(NEED_Z, 1, AX_Y, 0, 0, 0, 0xFF), SLOTV_(NEED_Z, 4, AX_X, 0, 0, 0, 0xFF),
	  BELOW_(NEED_Z, 0, 0, -1, 0, 0)},
	 {CELL_(2), CELL_(11)}},
	/* 4: MC:931-968 */
	{4, 5, AX_Y, OWN_Z0, 5, 2,
	 {CELL_(8), SLOTV_(NEED_Y, 0, AX_X, 0, 0, 0, 0xFF), SLOTV_(NEED_Y, 7, AX_Z, 1, 0, 0, 0xFF),
	  SLOTF_(NEED_Y, 1, -1, 0, AX_Y, 1, -1, 0), SLOTF_(NEED_Y | NEED_X1, 2, 0, 0, AX_X, 1, 0, 0)},
	 {CELL_(5), CELL_(9)}},
	/* 5: MC:969-1003 */
	{5, 6, AX_Z, 0, 5, 0,
	 {SLOTV_(NEED_Z, 4, AX_Y, 1, 0, 0, 4), SLOTV_(NEED_Z, 1, AX_X, 0, 1, 0, 9),
	  SLOTF_(NEED_Z | NEED_X1, 2, 1, 0, AX_X, 1, 1, 0), SLOTF_(NEED_Z | NEED_Y1, 1, 2, 0, AX_Y, 1, 1, 0),
	  BELOW_(NEED_Z, 1, 1, -1, 1, 1)},
	 {{0}}},
	/* 6: MC:1004-1042 */
	{7, 6, AX_Y, 0, 5, 2,
	 {SLOTV_(NEED_Y, 3, AX_X, 0, 0, 1, 11), SLOTV_(NEED_Y, 4, AX_Z, 1, 0, 0, 7),
	  SLOTF_(NEED_Y, 1, -1, 1, AX_Y, 1, -1, 1), SLOTF_(NEED_Y | NEED_X1, 2, 0, 1, AX_X, 1, 0, 1),
	  CELLN_(11, NEED_NOT_Y)},
	 {CELL_(5), CELL_(10)}},
	/* 7: MC:1043-1081 */
	{4, 7, AX_Z, OWN_Y0, 6, 2,
	 {CELL_(8), CELL_(4), SLOTV_(NEED_Z, 0, AX_X, 0, 0, 0, 0xFF), SLOTV_(NEED_Z, 5, AX_Y, 1, 0, 0, 0xFF),
	  SLOTF_(NEED_Z | NEED_X1, 2, 0, 0, AX_X, 1, 0, 0), BELOW_(NEED_Z, 1, 0, -1, 1, 0)},
	 {CELL_(6), CELL_(11)}},
	/* 8: MC:1082-1115 */
	{0, 4, AX_X, OWN_Y0 | OWN_Z0, 5, 2,
	 {CELL_(3), CELL_(0), SLOTV_(NEED_X, 3, AX_Z, 0, 0, 0, 0xFF), SLOTV_(NEED_X, 1, AX_Y, 0, 0, 0, 0xFF),
	  SLOTF_(NEED_X, -1, 0, 0, AX_X, -1, 0, 0)},
	 {CELL_(4), CELL_(7)}},
	/* 9: MC:1116-1151 */
	{1, 5, AX_X, OWN_Z0, 4, 2,
	 {CELL_(0), SLOTV_(NEED_X, 0, AX_Y, 0, 0, 0, 0xFF), SLOTV_(NEED_X, 2, AX_Z, 0, 1, 0, 0xFF),
	  SLOTF_(NEED_X, -1, 1, 0, AX_X, -1, 1, 0)},
	 {CELL_(5), CELL_(4)}},
	/* 10: MC:1152-1188 */
	{2, 6, AX_X, 0, 4, 2,
	 {SLOTV_(NEED_X, 1, AX_Z, 0, 1, 0, 1), SLOTV_(NEED_X, 3, AX_Y, 0, 0, 1, 2),
	  SLOTF_(NEED_X, -1, 1, 1, AX_X, -1, 1, 1), CELLN_(2, NEED_NOT_X)},
	 {CELL_(5), CELL_(6)}},
	/* 11: MC:1189-1224 */
	{3, 7, AX_X, OWN_Y0, 5, 2,
	 {CELL_(3), CELL_(2), SLOTV_(NEED_X, 0, AX_Z, 0, 0, 0, 0xFF), SLOTV_(NEED_X, 2, AX_Y, 0, 0, 1, 0xFF),
	  SLOTF_(NEED_X, -1, 0, 1, AX_X, -1, 0, 1)},
	 {CELL_(6), CELL_(7)}},
};

/* ------------------------------------------------------------------------------------------------ */
typedef struct {
	const mc33o_sample *F;
	uint32_t px, py, pz; /* points per axis */
	uint32_t nx, ny, nz; /* cells per axis  */
	mc33o_real iso;
	mc33o_real O[3], D[3], ca, cb;
	int store_mode;      /* 0: spn0 (MC:485), 1: spnA (MC:518), 2: spnB (MC:551), 3: spnC (MC:587) */
	double A[3][3], Ai[3][3]; /* spnC: _A[j][i]*d[i] and A_[j][i]/d[j] (MC:1763-1770) */
	int triangular;      /* spnC: mult_Abf points at _multTSA_bf (UTIL:86-97) instead of _multA_bf (UTIL:99-112) */
	uint32_t nV, nT, capV, capT;
	mc33o_real *V;
	float *N;
	uint32_t *T;
	int fault;
	uint32_t *idX[2], *idY[2], *idZ; /* ids of x/y edges on the two live planes, z edges of the slice */
} oracle_ctx;

static inline uint32_t sign_of(mc33o_real f) { /* MC:402-408 */
#ifdef MC33_ORACLE_F64
	uint64_t u; memcpy(&u, &f, 8); return (uint32_t)(u >> 63);
#else
	uint32_t u; memcpy(&u, &f, 4); return u >> 31;
#endif
}

static inline mc33o_sample sample(const oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z) {
	return c->F[((size_t)z * c->py + y) * c->px + x];
}

/* difference of two samples as the reference's expression yields it: float / double for float / double grids,
 * int (then converted) for unsigned char / short grids (integer promotion), unsigned modulo 2^32 for
 * unsigned int grids (no promotion) */
static inline mc33o_real sdiff(mc33o_sample a, mc33o_sample b) {
#if defined(MC33_ORACLE_U16) || defined(MC33_ORACLE_U8)
	return (float)((int)a - (int)b);
#elif defined(MC33_ORACLE_U32)
	return (float)(uint32_t)(a - b);
#else
	return a - b;
#endif
}

static uint32_t *id_slot(oracle_ctx *c, int axis, uint32_t x, uint32_t y, uint32_t z) {
	switch (axis) {
	case AX_X: return c->idX[z & 1] + (size_t)y * c->nx + x;
	case AX_Y: return c->idY[z & 1] + (size_t)y * (c->nx + 1) + x;
	default:   return c->idZ + (size_t)y * (c->nx + 1) + x;
	}
}

/* c = A b or A^T b in double, rounded to MC33_real on assignment: the two forms of mult_Abf (UTIL:86-112) */
static void mat_vec(const double (*A)[3], mc33o_real *b, int transposed, int triangular) {
	if (triangular) {
		if (transposed) {
			b[2] = A[0][2] * b[0] + A[1][2] * b[1] + A[2][2] * b[2];
			b[1] = A[0][1] * b[0] + A[1][1] * b[1];
			b[0] = A[0][0] * b[0];
		} else {
			b[0] = A[0][0] * b[0] + A[0][1] * b[1] + A[0][2] * b[2];
			b[1] = A[1][1] * b[1] + A[1][2] * b[2];
			b[2] = A[2][2] * b[2];
		}
		return;
	}
	double u, v;
	if (transposed) {
		u = A[0][0] * b[0] + A[1][0] * b[1] + A[2][0] * b[2];
		v = A[0][1] * b[0] + A[1][1] * b[1] + A[2][1] * b[2];
		b[2] = A[0][2] * b[0] + A[1][2] * b[1] + A[2][2] * b[2];
	} else {
		u = A[0][0] * b[0] + A[0][1] * b[1] + A[0][2] * b[2];
		v = A[1][0] * b[0] + A[1][1] * b[1] + A[1][2] * b[2];
		b[2] = A[2][0] * b[0] + A[2][1] * b[1] + A[2][2] * b[2];
	}
	b[0] = u; b[1] = v;
}

/* append one vertex: r[0..2] grid-index position, r[3..5] gradient (MC:485-621) */
static uint32_t emit_vertex(oracle_ctx *c, mc33o_real *r) {
	uint32_t id = c->nV;
	if (id == c->capV) {
		uint32_t ncap = c->capV * 2;
		mc33o_real *nv = (mc33o_real *)realloc(c->V, (size_t)ncap * 3 * sizeof(mc33o_real));
		if (nv) c->V = nv;
		float *nn = (float *)realloc(c->N, (size_t)ncap * 3 * sizeof(float));
		if (nn) c->N = nn;
		if (!nv || !nn) { c->fault = 1; return 0; }
		c->capV = ncap;
	}
	c->nV++;
	mc33o_real *p = c->V + 3 * (size_t)id;
	if (c->store_mode == 0) {
		p[0] = r[0]; p[1] = r[1]; p[2] = r[2];
	} else if (c->store_mode == 3) { /* MC:607-612 */
		mat_vec(c->A, r, 0, c->triangular);
		for (int k = 0; k < 3; k++) p[k] = r[k] + c->O[k];
		mat_vec(c->Ai, r + 3, 1, c->triangular);
	} else {
		for (int k = 0; k < 3; k++) p[k] = r[k] * c->D[k] + c->O[k];
		if (c->store_mode == 2) { r[3] *= c->ca; r[4] *= c->cb; }
	}
	/* MC:510-515 with the exact form of invSqrt (MC:70-73): the sum is MC33_real, the root and the normal are float */
	float s = 1.0f / sqrtf((float)(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]));
	float *n = c->N + 3 * (size_t)id;
	n[0] = s * (float)r[3]; n[1] = s * (float)r[4]; n[2] = s * (float)r[5];
	return id;
}

/* vertex exactly on grid point (x,y,z): MC:628-649 */
static uint32_t emit_on_point(oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z) {
	mc33o_real r[6];
	const uint32_t q[3] = {x, y, z}, lim[3] = {c->nx, c->ny, c->nz};
	r[0] = (mc33o_real)x; r[1] = (mc33o_real)y; r[2] = (mc33o_real)z;
	for (int ax = 0; ax < 3; ax++) {
		uint32_t lo[3] = {x, y, z}, hi[3] = {x, y, z};
		mc33o_real w;
		if (q[ax] == 0) { hi[ax] = 1; w = 1.0f; }
		else if (q[ax] == lim[ax]) { lo[ax] = q[ax] - 1; w = 1.0f; }
		else { lo[ax] = q[ax] - 1; hi[ax] = q[ax] + 1; w = 0.5f; }
		mc33o_real dlt = sdiff(sample(c, lo[0], lo[1], lo[2]), sample(c, hi[0], hi[1], hi[2]));
		r[3 + ax] = (w == 1.0f) ? dlt : 0.5f * dlt;
	}
	return emit_vertex(c, r);
}

/* regular vertex on edge e of cell (x,y,z): MC:810-816 ... 1212-1220 and SURVEY Appendix C */
static uint32_t emit_on_edge(oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z, int e, const mc33o_real *v) {
	const edge_rule *E = &EDGE[e];
	const uint32_t cell[3] = {x, y, z}, lim[3] = {c->nx, c->ny, c->nz};
	const uint8_t *oa = CORNER_OFF[E->a], *ob = CORNER_OFF[E->b];
	mc33o_real r[6];
	mc33o_real t = v[E->a] / (v[E->a] - v[E->b]);
	for (int ax = 0; ax < 3; ax++) {
		if (ax == E->axis) {
			r[ax] = (mc33o_real)cell[ax] + t;
			r[3 + ax] = v[E->b] - v[E->a];
			continue;
		}
		r[ax] = (mc33o_real)(cell[ax] + oa[ax]);
		if (oa[ax] == 1 && cell[ax] + 1 < lim[ax]) {
			/* central differences around the edge, blended along it */
			uint32_t pa[3] = {x + oa[0], y + oa[1], z + oa[2]}, pb[3] = {x + ob[0], y + ob[1], z + ob[2]};
			uint32_t al[3], ah[3], bl[3], bh[3];
			memcpy(al, pa, sizeof pa); memcpy(ah, pa, sizeof pa);
			memcpy(bl, pb, sizeof pb); memcpy(bh, pb, sizeof pb);
			al[ax]--; ah[ax]++; bl[ax]--; bh[ax]++;
			mc33o_real da = sdiff(sample(c, al[0], al[1], al[2]), sample(c, ah[0], ah[1], ah[2]));
			mc33o_real db = sdiff(sample(c, bl[0], bl[1], bl[2]), sample(c, bh[0], bh[1], bh[2]));
			r[3 + ax] = 0.5f * (da * (1 - t) + db * t);
		} else {
			/* one-sided: difference of v across the cell at both end points */
			uint8_t a_lo[3] = {oa[0], oa[1], oa[2]}, a_hi[3] = {oa[0], oa[1], oa[2]};
			uint8_t b_lo[3] = {ob[0], ob[1], ob[2]}, b_hi[3] = {ob[0], ob[1], ob[2]};
			a_lo[ax] = 0; a_hi[ax] = 1; b_lo[ax] = 0; b_hi[ax] = 1;
			mc33o_real da = v[CORNER_AT[a_hi[0]][a_hi[1]][a_hi[2]]] - v[CORNER_AT[a_lo[0]][a_lo[1]][a_lo[2]]];
			mc33o_real db = v[CORNER_AT[b_hi[0]][b_hi[1]][b_hi[2]]] - v[CORNER_AT[b_lo[0]][b_lo[1]][b_lo[2]]];
			r[3 + ax] = da * (1 - t) + db * t;
		}
	}
	return emit_vertex(c, r);
}

/* cell-centre vertex (pattern nibble 0xC): MC:1225-1230 */
static uint32_t emit_centre(oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z, const mc33o_real *v) {
	mc33o_real r[6];
	r[0] = x + 0.5f; r[1] = y + 0.5f; r[2] = z + 0.5f;
	r[3] = v[4] + v[5] + v[6] + v[7] - v[0] - v[1] - v[2] - v[3];
	r[4] = v[1] + v[2] + v[5] + v[6] - v[0] - v[3] - v[4] - v[7];
	r[5] = v[2] + v[3] + v[6] + v[7] - v[0] - v[1] - v[4] - v[5];
	return emit_vertex(c, r);
}

/* ------------------------------------------------------------------------------------------------
 * Ambiguity tests (MC:347-386 face tests, MC:431-462 interior test)
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t FACE_PROD[6][4] = {/* v[a]*v[b] < v[c]*v[d] */
	{0, 5, 1, 4}, {1, 6, 2, 5}, {3, 6, 2, 7}, {0, 7, 3, 4}, {0, 2, 1, 3}, {4, 6, 5, 7}};
static const uint8_t FACE_MASK[6] = {0xCC, 0x66, 0x33, 0x99, 0xF0, 0x0F};
static const uint8_t FACE_DIAG_HI[6] = {0x84, 0x42, 0x12, 0x81, 0xA0, 0x0A}; /* diagonal containing v0 / v6 */
static const uint8_t FACE_DIAG_LO[6] = {0x48, 0x24, 0x21, 0x18, 0x50, 0x05};
static const uint8_t FACE_KEYBIT[6] = {0x80, 0x02, 0x02, 0x80, 0x80, 0x02};  /* v0 for faces 0,3,4; v6 for 1,2,5 */

static inline int face_less(int f, const mc33o_real *v) {
	const uint8_t *q = FACE_PROD[f];
	return v[q[0]] * v[q[1]] < v[q[2]] * v[q[3]];
}
/* MC:371-386 */
static inline unsigned face_test_one(int f, const mc33o_real *v) { return face_less(f, v) ? FACE_DIAG_LO[f] : FACE_DIAG_HI[f]; }
/* MC:347-367 */
static int face_tests(int *face, unsigned ind, const mc33o_real *v) {
	int sum = 0;
	for (int f = 0; f < 6; f++) {
		int r = 0;
		if (ind & FACE_KEYBIT[f]) {
			if ((ind & FACE_MASK[f]) == FACE_DIAG_HI[f]) r = face_less(f, v) ? -1 : 1;
		} else {
			if ((ind & FACE_MASK[f]) == FACE_DIAG_LO[f]) r = face_less(f, v) ? 1 : -1;
		}
		face[f] = r;
		sum += r;
	}
	return sum;
}
/* MC:431-462 */
static int interior_test(int s, int flag13, const mc33o_real *v) {
	mc33o_real a = v[4] - v[0], b = v[5] - v[1], cc = v[6] - v[2], d = v[7] - v[3];
	mc33o_real t = a * cc - b * d;
	if (sign_of(t)) { if (s & 1) return 0; }
	else if (!(s & 1) || t == 0) return 0;
	t = 0.5f * (v[3] * b - v[2] * a + v[1] * d - v[0] * cc) / t;
	if (t > 0 && t < 1) {
		a = v[0] + a * t; b = v[1] + b * t; cc = v[2] + cc * t; d = v[3] + d * t;
		cc *= a; d *= b;
		if (s & 1) { if (cc < d && !sign_of(d)) return (sign_of(b) == sign_of(v[s])) + flag13; }
		else { if (cc > d && !sign_of(cc)) return (sign_of(a) == sign_of(v[s])) + flag13; }
	}
	return 0;
}

/* MC:683-779: table word -> offset of the triangle pattern (the walk starts at offset+1, MC:781) */
static unsigned pattern_offset(unsigned i, const mc33o_real *v, unsigned *flip_m, unsigned *flip_n) {
	unsigned c, m, n;
	if (i & 0x80) { c = mc33o_lut[i ^ 0xFF]; m = (c & 0x800) == 0; n = !m; }
	else { c = mc33o_lut[i]; n = (c & 0x800) == 0; m = !n; }
	*flip_m = m; *flip_n = n;
	const unsigned k = c & 0x7FF, ci = m ? i : i ^ 0xFF;
	int f[6];
	switch (c >> 12) {
	case 0: return k;
	case 1: return (ci & face_test_one(k >> 2, v)) ? 183 + 2 * k : 159 + k;
	case 2: return interior_test(k, 0, v) ? 239 + 6 * k : 231 + 2 * k;
	case 3:
		if (ci & face_test_one(k % 6, v)) return 575 + 5 * k;
		return interior_test(k / 6, 0, v) ? 407 + 7 * k : 335 + 3 * k;
	case 4:
		switch (face_tests(f, ci, v)) {
		case -3: return 695 + 3 * k;
		case -1: return (f[4] + f[5] < 0 ? (f[0] + f[2] < 0 ? 759 : 799) : 719) + 5 * k;
		case 1: return (f[4] + f[5] < 0 ? 983 : (f[0] + f[2] < 0 ? 839 : 911)) + 9 * k;
		default: return interior_test(k >> 1, 0, v) ? 1095 + 9 * k : 1055 + 5 * k;
		}
	case 5:
		switch (face_tests(f, ci, v)) {
		case -2:
			if (k == 2 ? interior_test(0, 0, v) : (interior_test(0, 0, v) || interior_test(k ? 1 : 3, 0, v)))
				return 1213 + 8 * k;
			return 1189 + 4 * k;
		case 0: return (f[2 + k] < 0 ? 1261 : 1285) + 8 * k;
		default:
			if (k == 2 ? interior_test(1, 0, v) : (interior_test(2, 0, v) || interior_test(k ? 3 : 1, 0, v)))
				return 1237 + 8 * k;
			return 1201 + 4 * k;
		}
	case 6:
		switch (face_tests(f, ci, v)) {
		case -2: return interior_test((0xDA010C >> (2 * k)) & 3, 0, v) ? 1453 + 8 * k : 1357 + 4 * k;
		case 0: return (f[k >> 1] < 0 ? 1645 : 1741) + 8 * k;
		default: return interior_test((0xA7B7E5 >> (2 * k)) & 3, 0, v) ? 1549 + 8 * k : 1405 + 4 * k;
		}
	default: {
		int s = face_tests(f, 165, v);
		if (s < 0) s = -s;
		switch (s) {
		case 0: {
			unsigned kk = ((f[1] < 0) << 1) | (f[5] < 0);
			if (f[0] * f[1] == f[5]) return 2157 + 12 * kk;
			int r = interior_test((int)kk, 1, v);
			return (unsigned)(2285 + (r ? 10 * (int)kk - 40 * r : 6 * (int)kk));
		}
		case 2: {
			unsigned off = 1917 + 10 * ((f[0] < 0 ? f[2] > 0 : 12 + (f[2] < 0)) + (f[1] < 0 ? f[3] < 0 : 6 + (f[3] > 0)));
			if (f[4] > 0) off += 30;
			return off;
		}
		case 4: {
			unsigned kk = (unsigned)(21 + 11 * f[0] + 4 * f[1] + 3 * f[2] + 2 * f[3] + f[4]);
			if (kk >> 4) kk -= (kk & 32 ? 20 : 10);
			return 1845 + 3 * kk;
		}
		default: return (unsigned)(1839 + 2 * f[0]);
		}
	}
	}
}

/* ------------------------------------------------------------------------------------------------
 * One active cell: MC:673-1253
 * ---------------------------------------------------------------------------------------------- */
static int need_ok(const oracle_ctx *c, unsigned need, uint32_t x, uint32_t y, uint32_t z) {
	if ((need & NEED_X) && !x) return 0;
	if ((need & NEED_Y) && !y) return 0;
	if ((need & NEED_Z) && !z) return 0;
	if ((need & NEED_X1) && !(x + 1 < c->nx)) return 0;
	if ((need & NEED_Y1) && !(y + 1 < c->ny)) return 0;
	if ((need & NEED_NOT_X) && x) return 0;
	if ((need & NEED_NOT_Y) && y) return 0;
	return 1;
}

/* id of the vertex of cut edge e whose end point (corner `zc`) has v == 0; *keep_slot is set when
 * the id came from the slice below and the edge's own slot must not be rewritten (MC:836-838 etc.) */
static uint32_t id_for_vertex_on_corner(oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z, const mc33o_real *v,
                                        uint32_t *p, const id_source *list, int n, int zc, int *keep_slot) {
	for (int k = 0; k < n; k++) {
		const id_source *s = &list[k];
		if (!need_ok(c, s->need, x, y, z)) continue;
		if (s->kind == SRC_CELL) {
			if (p[s->edge] != NOID) return p[s->edge];
			continue;
		}
		unsigned pass;
		if (s->guard == GUARD_CORNER) pass = sign_of(v[s->gcorner]);
		else pass = sign_of(c->iso - (mc33o_real)sample(c, x + s->fx, y + s->fy, z + s->fz));
		if (!pass) continue;
		uint32_t id = *id_slot(c, s->axis, x + s->sx, y + s->sy, z + s->sz);
		if (s->also != 0xFF) p[s->also] = id;
		if (s->kind == SRC_BELOW) *keep_slot = 1;
		return id;
	}
	return emit_on_point(c, x + CORNER_OFF[zc][0], y + CORNER_OFF[zc][1], z + CORNER_OFF[zc][2]);
}

static void add_triangle(oracle_ctx *c, uint32_t a, uint32_t b, uint32_t d) {
	if (c->nT == c->capT) {
		uint32_t *nt = (uint32_t *)realloc(c->T, (size_t)c->capT * 2 * 3 * sizeof(uint32_t));
		if (!nt) { c->fault = 1; return; }
		c->T = nt; c->capT *= 2;
	}
	uint32_t *t = c->T + 3 * (size_t)c->nT++;
	t[0] = a; t[1] = b; t[2] = d;
}

static void process_cell(oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z, unsigned i, const mc33o_real *v) {
	uint32_t p[13];
	for (int k = 0; k < 13; k++) p[k] = NOID;
	unsigned m, n;
	const unsigned short *pat = mc33o_lut + pattern_offset(i, v, &m, &n);
	unsigned word;
	do {
		word = *(++pat);
		uint32_t ti[3];
		unsigned w = word;
		for (int k = 3; k;) {
			unsigned e = w & 0xF;
			w >>= 4;
			if (p[e] == NOID) {
				if (e == 12) {
					p[12] = emit_centre(c, x, y, z, v);
				} else {
					const edge_rule *E = &EDGE[e];
					const int creator = !((E->own & OWN_X0) && x) && !((E->own & OWN_Y0) && y) && !((E->own & OWN_Z0) && z);
					const uint32_t bx = x + CORNER_OFF[E->a][0], by = y + CORNER_OFF[E->a][1], bz = z + CORNER_OFF[E->a][2];
					uint32_t *slot = id_slot(c, E->axis, bx, by, bz);
					if (!creator) {
						p[e] = *slot;
					} else {
						int keep = 0;
						if (v[E->a] == 0) p[e] = id_for_vertex_on_corner(c, x, y, z, v, p, E->A, E->nA, E->a, &keep);
						else if (v[E->b] == 0) p[e] = id_for_vertex_on_corner(c, x, y, z, v, p, E->B, E->nB, E->b, &keep);
						else p[e] = emit_on_edge(c, x, y, z, (int)e, v);
						if (!keep) *slot = p[e];
					}
				}
			}
			ti[--k] = p[e];
		}
		if (ti[0] != ti[1] && ti[0] != ti[2] && ti[1] != ti[2]) /* MC:1235 */
			add_triangle(c, ti[n], ti[m], ti[2]);               /* MC:1249 (MC33_NORMAL_NEG 0) */
	} while (word >> 12); /* top nibble 0 marks the last triangle (MC:780) */
}

/* ------------------------------------------------------------------------------------------------
 * Sweep over all cells, x fastest (MC:1816-1889)
 * ---------------------------------------------------------------------------------------------- */
static unsigned cell_values(const oracle_ctx *c, uint32_t x, uint32_t y, uint32_t z, mc33o_real *v) {
	unsigned i = 0;
	for (int k = 0; k < 8; k++) {
		v[k] = c->iso - (mc33o_real)sample(c, x + CORNER_OFF[k][0], y + CORNER_OFF[k][1], z + CORNER_OFF[k][2]); /* MC:1840-1855 */
		i |= sign_of(v[k]) << (7 - k);                                                              /* MC:1846-1859 */
	}
	return i;
}

int mc33o_calculate_isosurface(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz,
                               const double r0[3], const double d[3], mc33o_real iso, mc33o_surface *out) {
	return mc33o_calculate_isosurface_inclined(data, npx, npy, npz, r0, d, 0, 0, 0, iso, out);
}

int mc33o_calculate_isosurface_inclined(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz,
                                        const double r0[3], const double d[3], const double *grd_A, const double *grd_Ai,
                                        int triangular, mc33o_real iso, mc33o_surface *out) {
	oracle_ctx c;
	memset(&c, 0, sizeof c);
	memset(out, 0, sizeof *out);
	if (!data || npx < 2 || npy < 2 || npz < 2) return -1;
	c.F = data; c.px = npx; c.py = npy; c.pz = npz;
	c.nx = npx - 1; c.ny = npy - 1; c.nz = npz - 1;
	c.iso = iso;
	/* store selection: MC:1772-1782 */
	if (d[0] != d[1] || d[1] != d[2]) { c.store_mode = 2; c.ca = (mc33o_real)(d[2] / d[0]); c.cb = (mc33o_real)(d[2] / d[1]); }
	else c.store_mode = (d[0] == 1 && r0[0] == 0 && r0[1] == 0 && r0[2] == 0) ? 0 : 1;
	if (grd_A && grd_Ai) { /* G->nonortho: MC:1763-1770 */
		c.store_mode = 3;
		c.triangular = triangular;
		for (int j = 0; j < 3; j++)
			for (int i = 0; i < 3; i++) {
				c.A[j][i] = grd_A[3 * j + i] * d[i];
				c.Ai[j][i] = grd_Ai[3 * j + i] / d[j];
			}
	}
	for (int k = 0; k < 3; k++) { c.O[k] = (mc33o_real)r0[k]; c.D[k] = (mc33o_real)d[k]; }
	c.capV = c.capT = 4096;
	c.V = (mc33o_real *)malloc((size_t)c.capV * 3 * sizeof(mc33o_real));
	c.N = (float *)malloc((size_t)c.capV * 12);
	c.T = (uint32_t *)malloc((size_t)c.capT * 12);
	size_t nX = (size_t)(c.ny + 1) * c.nx, nY = (size_t)c.ny * (c.nx + 1), nZ = (size_t)(c.ny + 1) * (c.nx + 1);
	for (int k = 0; k < 2; k++) {
		c.idX[k] = (uint32_t *)malloc(nX * 4);
		c.idY[k] = (uint32_t *)malloc(nY * 4);
	}
	c.idZ = (uint32_t *)malloc(nZ * 4);
	int ok = c.V && c.N && c.T && c.idX[0] && c.idX[1] && c.idY[0] && c.idY[1] && c.idZ;
	if (ok) {
		mc33o_real v[8];
		for (uint32_t z = 0; z < c.nz; z++)
			for (uint32_t y = 0; y < c.ny; y++)
				for (uint32_t x = 0; x < c.nx; x++) {
					unsigned i = cell_values(&c, x, y, z, v);
					if (i != 0 && i != 0xFF) process_cell(&c, x, y, z, i, v); /* MC:1860-1862 */
				}
	}
	for (int k = 0; k < 2; k++) { free(c.idX[k]); free(c.idY[k]); }
	free(c.idZ);
	if (!ok || c.fault) { free(c.V); free(c.N); free(c.T); return -1; }
	out->nV = c.nV; out->nT = c.nT; out->V = c.V; out->N = c.N; out->T = c.T;
	return 0;
}

void mc33o_free_surface(mc33o_surface *s) {
	if (!s) return;
	free(s->V); free(s->N); free(s->T);
	memset(s, 0, sizeof *s);
}

int mc33o_classify(const mc33o_sample *data, uint32_t npx, uint32_t npy, uint32_t npz, mc33o_real iso,
                   uint8_t *index_out, uint16_t *pattern_out) {
	oracle_ctx c;
	memset(&c, 0, sizeof c);
	if (!data || npx < 2 || npy < 2 || npz < 2) return -1;
	c.F = data; c.px = npx; c.py = npy; c.pz = npz;
	c.nx = npx - 1; c.ny = npy - 1; c.nz = npz - 1;
	c.iso = iso;
	mc33o_real v[8];
	size_t q = 0;
	for (uint32_t z = 0; z < c.nz; z++)
		for (uint32_t y = 0; y < c.ny; y++)
			for (uint32_t x = 0; x < c.nx; x++, q++) {
				unsigned i = cell_values(&c, x, y, z, v), m, n;
				index_out[q] = (uint8_t)i;
				pattern_out[q] = (i != 0 && i != 0xFF) ? (uint16_t)pattern_offset(i, v, &m, &n) : 0;
			}
	return 0;
}

uint64_t mc33o_fnv1a64(const void *p, uint64_t nbytes) {
	const unsigned char *b = (const unsigned char *)p;
	uint64_t h = 0xcbf29ce484222325ull;
	for (uint64_t k = 0; k < nbytes; k++) { h ^= b[k]; h *= 0x100000001b3ull; }
	return h;
}

void mc33o_fill_cos_field(float *data, uint32_t n, double lo, double h) {
	double *cs = (double *)malloc(sizeof(double) * n);
	double a = lo;
	for (uint32_t k = 0; k < n; k++) { cs[k] = cos(a); a += h; } /* MC33_util_grd.c:660-672 */
	size_t q = 0;
	for (uint32_t k = 0; k < n; k++)
		for (uint32_t j = 0; j < n; j++)
			for (uint32_t i = 0; i < n; i++) data[q++] = (float)(cs[i] + cs[j] + cs[k]);
	free(cs);
}
