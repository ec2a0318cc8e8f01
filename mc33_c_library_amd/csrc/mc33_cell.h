// mc33_cell.h -- per-cell Marching Cubes 33 logic of the MI355X path, written so that every cell
// can be processed independently of the serial sweep the reference uses.
//
// The reference (source/marching_cubes_33.c, "MC:") numbers vertices in the order its z->y->x sweep
// first creates them and shares ids between neighbouring cells through five rolling 2-D caches
// (MC:1783-1808).  Here the same numbering is obtained without a sweep:
//
//   * every cut grid edge has exactly one OWNER cell - the first cell of the sweep that contains it
//      - and the owner alone decides what the edge's vertex id is;
//   * a cell's decision for each of its edges (its "plan") is a pure function of the grid - the 8 corner
//     values, the boundary flags and a few neighbouring samples: either a NEW vertex, whose id is
//     (#vertices created by earlier cells) + (rank of first appearance in this cell's pattern walk),
//     or an ALIAS of another grid edge (only when a corner value equals the isovalue exactly,
//     MC:788-1224), which is followed to that edge's owner until a NEW vertex is reached.
//
// The functions are __host__ __device__: the HIP kernels (mc33_kernels.hip) run them on the GPU, and
// tests/host_emu builds them with g++ so the formulation can be checked against the oracle on a box
// without a GPU.  They are NOT a CPU fallback: the product library only calls them from device code.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MC33_HD __host__ __device__ inline
#else
#define MC33_HD inline
#endif

namespace mc33 {

// ---------------------------------------------------------------------------------------------------
// Problem description shared by all kernels
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t SEG_CELLS = 256;  // cells of one row segment (one wave: 64 lanes x 4 samples)
constexpr uint32_t NO_ID = 0xFFFFFFFFu;

// MC33_real of the reference (marching_cubes_33.h:66-88): corner values, interpolation, vertex positions and the
// isovalue are float, except in the double build (GRD_TYPE_SIZE 8) where they are double.  Normals are always float.
#ifdef MC33_REAL_DOUBLE
typedef double real_t;
#else
typedef float real_t;
#endif

struct Params {
	uint32_t nx, ny, nz;  // cells per axis (= _GRD.N, marching_cubes_33.h:113)
	uint32_t nseg;        // row segments per row = ceil(nx / SEG_CELLS)
	uint32_t zs;          // first cell slice this launch covers (row-segment records are relative to it)
	real_t iso;
	real_t O[3], D[3];     // MC33_real copies of r0, d (MC:1779-1782)
	real_t ca, cb;         // MC:1773-1774
	int32_t store_mode;   // 0: MC33_spn0 (MC:485), 1: MC33_spnA (MC:518), 2: MC33_spnB (MC:551), 3: MC33_spnC (MC:587)
	int32_t triangular;   // spnC: mult_Abf is _multTSA_bf (UTIL:86-97) rather than _multA_bf (UTIL:99-112)
	int32_t negzero_iso;  // the isovalue is -0.0: a sample equal to it gives v = -0, "zero AND negative", and the degenerate-vertex
	                      // rules may then point at grid edges whose owner cell is not cut at all (see edge_vertex_id)
	int32_t normal_neg;   // the reference's MC33_NORMAL_NEG (libMC33.c:20-22): normals negated (MC:509-513), first two
	                      // indices of every triangle exchanged (MC:1246-1250)
	double A[9], Ai[9];   // spnC: M->_A, M->A_ (row major; MC:1763-1770)
};

struct Tables {
	const uint16_t *lut;        // MC33_all_tables data (mc33_lut_data.h)
	const uint32_t *rule_words; // mc33_rules_data.h
	const uint8_t *rule_index;  // [12][4]
};

template <typename T>
struct __attribute__((packed, aligned(sizeof(T)))) SamplePair {  // two consecutive samples: one 8-byte (4-byte) load
	T a, b;
};

template <typename T>
struct GridView {  // pitched copy of _GRD.F in HBM: sample (x,y,z) at p[(z-z0)*slice + y*pitch + x]
	const T *p;
	uint32_t pitch;  // samples per row (>= nx+1)
	uint32_t z0;     // global index of the first resident plane (0 unless the volume is z-slabbed over GPUs)
	uint64_t slice;  // samples per plane
	MC33_HD T at(uint32_t x, uint32_t y, uint32_t z) const {
#ifdef MC33_BOUNDS_HOOK  // tests/host_emu only: record reads outside the resident planes
		MC33_BOUNDS_HOOK(x, y, z);
#endif
		return p[(uint64_t)(z - z0) * slice + (uint64_t)y * pitch + x];
	}
	MC33_HD SamplePair<T> pair(uint32_t x, uint32_t y, uint32_t z) const {  // samples x and x+1 of a row
#ifdef MC33_BOUNDS_HOOK
		MC33_BOUNDS_HOOK(x + 1, y, z);
#endif
		return *(const SamplePair<T> *)(p + ((uint64_t)(z - z0) * slice + (uint64_t)y * pitch + x));
	}
};

MC33_HD float sample_diff(float a, float b) { return a - b; }
MC33_HD double sample_diff(double a, double b) { return a - b; }
// unsigned short promotes to int in the reference's expressions (e.g. MC:851); the difference meets
// a float operand only afterwards
MC33_HD float sample_diff(uint16_t a, uint16_t b) { return (float)((int)a - (int)b); }
MC33_HD float sample_diff(uint8_t a, uint8_t b) { return (float)((int)a - (int)b); }
// unsigned int does not promote: the difference wraps modulo 2^32 before it becomes a float
MC33_HD float sample_diff(uint32_t a, uint32_t b) { return (float)(uint32_t)(a - b); }

// v = iso - F as the reference's CPU code gets it (MC:1840-1855).  Only a NaN sample needs care: x86 (and ARM) hand
// back that NaN with ITS sign, so signbf(v) is the sign bit of the sample, while the GPU's subtract negates the second
// operand first and returns the NaN with the sign flipped.  The reference's sign classification of a NaN sample is
// therefore "the NaN's own sign bit" and that is what every pass here reproduces (tools/nan_check.py).
MC33_HD float iso_diff(float iso, float f) { return f != f ? f : iso - f; }
MC33_HD double iso_diff(double iso, double f) { return f != f ? f : iso - f; }

MC33_HD uint32_t sign_of(float f) { return __builtin_bit_cast(uint32_t, f) >> 31; }  // MC:406-408
MC33_HD uint32_t sign_of(double f) { return (uint32_t)(__builtin_bit_cast(uint64_t, f) >> 63); }  // MC:402-404

// per-thread arrays living wherever the caller wants them (the stack on the host, an LDS column on
// the device so that run-time indexing never turns into scratch memory)
struct VRef {
	real_t *p;
	int stride;
	MC33_HD real_t &operator[](int k) const { return p[k * stride]; }
};
struct URef {
	uint32_t *p;
	int stride;
	MC33_HD uint32_t &operator[](int k) const { return p[k * stride]; }
};

// ---------------------------------------------------------------------------------------------------
// Cube geometry (MC:332-341 vertices, MC:660-668 edges), nibble-packed: lookups are shifts
// ---------------------------------------------------------------------------------------------------
MC33_HD uint32_t corner_code(uint32_t k) { return (0x57314620u >> (4 * k)) & 7u; }   // k -> dx | dy<<1 | dz<<2
MC33_HD uint32_t corner_at(uint32_t code) { return (0x62735140u >> (4 * code)) & 7u; }  // inverse
// edge e: first end point a (t is measured from a: MC:810,847,883,...), second end point b, direction
MC33_HD uint32_t edge_a(uint32_t e) { return (uint32_t)(0x321047540310ull >> (4 * e)) & 15u; }
MC33_HD uint32_t edge_b(uint32_t e) { return (uint32_t)(0x765476653221ull >> (4 * e)) & 15u; }
MC33_HD uint32_t edge_axis(uint32_t e) { return (uint32_t)(0x000021212121ull >> (4 * e)) & 15u; }
// owner rule: bit0 "only if x==0", bit1 "only if y==0", bit2 "only if z==0"
// (MC:788, 822, 861, 897, 932, 1044, 1083, 1117, 1190)
MC33_HD uint32_t edge_own(uint32_t e) { return (uint32_t)(0x204620043115ull >> (4 * e)) & 15u; }

// grid edge (axis, base point) -> its owner cell and the edge's index inside the owner
struct OwnerRef {
	uint32_t x, y, z, e;
};
MC33_HD OwnerRef owner_of(uint32_t axis, uint32_t px, uint32_t py, uint32_t pz) {
	OwnerRef o;
	uint32_t d0, d1;
	if (axis == 0) {
		o.x = px; o.y = py ? py - 1 : 0; o.z = pz ? pz - 1 : 0;
		d0 = py - o.y; d1 = pz - o.z;
		o.e = d1 ? (d0 ? 10u : 11u) : (d0 ? 9u : 8u);
	} else if (axis == 1) {
		o.x = px ? px - 1 : 0; o.y = py; o.z = pz ? pz - 1 : 0;
		d0 = px - o.x; d1 = pz - o.z;
		o.e = d0 ? (d1 ? 6u : 4u) : (d1 ? 2u : 0u);
	} else {
		o.x = px ? px - 1 : 0; o.y = py ? py - 1 : 0; o.z = pz;
		d0 = px - o.x; d1 = py - o.y;
		o.e = d0 ? (d1 ? 5u : 7u) : (d1 ? 1u : 3u);
	}
	return o;
}

// the 8 corner values v[k] = iso - F (MC:1840-1855) and the sign index (MC:1846-1859)
template <typename T>
MC33_HD uint32_t load_cell(const GridView<T> &G, real_t iso, uint32_t x, uint32_t y, uint32_t z, const VRef &v) {
	uint32_t i = 0;
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t c = corner_code(k);
		const real_t d = iso_diff(iso, (real_t)G.at(x + (c & 1), y + ((c >> 1) & 1), z + (c >> 2)));
		v[k] = d;
		i |= sign_of(d) << (7 - k);
	}
	return i;
}

// ---------------------------------------------------------------------------------------------------
// Ambiguity tests (MC:347-386 face tests, MC:431-462 interior test) and case selection (MC:683-779)
// ---------------------------------------------------------------------------------------------------
// The corner values come either through a VRef (any memory) or as a Corner8 (registers: every index below is static
// except the one of corner_value, which is a chain of selects there).
struct Corner8 {
	real_t a[8];
	MC33_HD real_t operator[](int k) const { return a[k]; }
};
MC33_HD real_t corner_value(const VRef &v, int s) { return v[s]; }
MC33_HD real_t corner_value(const Corner8 &v, int s) {
	// (copies first: `c ? v.a[3] : v.a[2]` selects between two lvalues, i.e. between two ADDRESSES, and the load through the
	// selected address is what keeps the values out of registers)
	const real_t a0 = v.a[0], a1 = v.a[1], a2 = v.a[2], a3 = v.a[3], a4 = v.a[4], a5 = v.a[5], a6 = v.a[6], a7 = v.a[7];
	const real_t lo = (s & 2) ? ((s & 1) ? a3 : a2) : ((s & 1) ? a1 : a0);
	const real_t hi = (s & 2) ? ((s & 1) ? a7 : a6) : ((s & 1) ? a5 : a4);
	return (s & 4) ? hi : lo;
}

template <typename V>
MC33_HD bool face_less(int f, const V &v) {
	switch (f) {
	case 0: return v[0] * v[5] < v[1] * v[4];
	case 1: return v[1] * v[6] < v[2] * v[5];
	case 2: return v[3] * v[6] < v[2] * v[7];
	case 3: return v[0] * v[7] < v[3] * v[4];
	case 4: return v[0] * v[2] < v[1] * v[3];
	default: return v[4] * v[6] < v[5] * v[7];
	}
}
// (register-held values: all six comparisons with static indices, the face picks its bit - a run-time face number must not
// become an index into the values, which would move them to scratch memory)
MC33_HD bool face_less(int f, const Corner8 &v) {
	const uint32_t bits = (uint32_t)(v.a[0] * v.a[5] < v.a[1] * v.a[4]) | (uint32_t)(v.a[1] * v.a[6] < v.a[2] * v.a[5]) << 1 |
	                      (uint32_t)(v.a[3] * v.a[6] < v.a[2] * v.a[7]) << 2 | (uint32_t)(v.a[0] * v.a[7] < v.a[3] * v.a[4]) << 3 |
	                      (uint32_t)(v.a[0] * v.a[2] < v.a[1] * v.a[3]) << 4 | (uint32_t)(v.a[4] * v.a[6] < v.a[5] * v.a[7]) << 5;
	return (bits >> f) & 1u;
}
// per face f (byte f): index mask, the diagonal through v0 (faces 0,3,4) / v6 (faces 1,2,5), the other one
MC33_HD uint32_t face_mask(int f) { return (uint32_t)(0x0FF0993366CCull >> (8 * f)) & 0xFFu; }
MC33_HD uint32_t face_diag_hi(int f) { return (uint32_t)(0x0AA081124284ull >> (8 * f)) & 0xFFu; }
MC33_HD uint32_t face_diag_lo(int f) { return (uint32_t)(0x055018212448ull >> (8 * f)) & 0xFFu; }

template <typename V>
MC33_HD uint32_t face_test_one(int f, const V &v) { return face_less(f, v) ? face_diag_lo(f) : face_diag_hi(f); }  // MC:371-386

// results of the six face tests (-1, 0, 1 each), two bits per face so that picking one by a run-time number is a shift
struct FaceSigns {
	uint32_t bits;  // (r + 1) << 2 f
	int sum;
	MC33_HD int operator[](int f) const { return (int)((bits >> (2 * f)) & 3u) - 1; }
};
template <typename V>
MC33_HD FaceSigns face_tests(uint32_t ind, const V &v) {  // MC:347-367
	FaceSigns fs{0u, 0};
	for (int f = 0; f < 6; f++) {
		const uint32_t key = (f == 0 || f == 3 || f == 4) ? 0x80u : 0x02u;
		int r = 0;
		if (ind & key) {
			if ((ind & face_mask(f)) == face_diag_hi(f)) r = face_less(f, v) ? -1 : 1;
		} else {
			if ((ind & face_mask(f)) == face_diag_lo(f)) r = face_less(f, v) ? 1 : -1;
		}
		fs.bits |= (uint32_t)(r + 1) << (2 * f);
		fs.sum += r;
	}
	return fs;
}

template <typename V>
MC33_HD int interior_test(int s, int flag13, const V &v) {  // MC:431-462
	real_t a = v[4] - v[0], b = v[5] - v[1], c = v[6] - v[2], d = v[7] - v[3];
	real_t t = a * c - b * d;
	if (sign_of(t)) {
		if (s & 1) return 0;
	} else if (!(s & 1) || t == 0)
		return 0;
	t = 0.5f * (v[3] * b - v[2] * a + v[1] * d - v[0] * c) / t;
	if (t > 0 && t < 1) {
		a = v[0] + a * t; b = v[1] + b * t; c = v[2] + c * t; d = v[3] + d * t;
		c *= a; d *= b;
		if (s & 1) {
			if (c < d && !sign_of(d)) return (int)(sign_of(b) == sign_of(corner_value(v, s))) + flag13;
		} else {
			if (c > d && !sign_of(c)) return (int)(sign_of(a) == sign_of(corner_value(v, s))) + flag13;
		}
	}
	return 0;
}

// table word -> offset of the triangle pattern; the walk starts at offset+1 (MC:781).
// m,n: which of the first two triangle slots is written first (winding, MC:683-691, 1249)
template <typename V>
MC33_HD uint32_t pattern_offset_word(uint32_t c, uint32_t i, const V &v, uint32_t &m, uint32_t &n);
template <typename V>
MC33_HD uint32_t pattern_offset(const uint16_t *lut, uint32_t i, const V &v, uint32_t &m, uint32_t &n) {
	return pattern_offset_word(lut[(i & 0x80) ? (i ^ 0xFF) : i], i, v, m, n);
}
// (c: the table word of the sign index, tables[i] or tables[i ^ 0xFF] for i >= 128)
template <typename V>
MC33_HD uint32_t pattern_offset_word(uint32_t c, uint32_t i, const V &v, uint32_t &m, uint32_t &n) {
	if (i & 0x80) { m = (c & 0x800) == 0; n = !m; }
	else { n = (c & 0x800) == 0; m = !n; }
	const uint32_t k = c & 0x7FF, ci = m ? i : i ^ 0xFF;
	switch (c >> 12) {
	case 0: return k;                                                                    // cases 1,2,5,8,9,11,14
	case 1: return (ci & face_test_one((int)(k >> 2), v)) ? 183 + 2 * k : 159 + k;       // case 3
	case 2: return interior_test((int)k, 0, v) ? 239 + 6 * k : 231 + 2 * k;              // case 4
	case 3:                                                                              // case 6
		if (ci & face_test_one((int)(k % 6), v)) return 575 + 5 * k;
		return interior_test((int)(k / 6), 0, v) ? 407 + 7 * k : 335 + 3 * k;
	case 4: {                                                                            // case 7
		const FaceSigns f = face_tests(ci, v);
		switch (f.sum) {
		case -3: return 695 + 3 * k;
		case -1: return (f[4] + f[5] < 0 ? (f[0] + f[2] < 0 ? 759u : 799u) : 719u) + 5 * k;
		case 1: return (f[4] + f[5] < 0 ? 983u : (f[0] + f[2] < 0 ? 839u : 911u)) + 9 * k;
		default: return interior_test((int)(k >> 1), 0, v) ? 1095 + 9 * k : 1055 + 5 * k;
		}
	}
	case 5: {                                                                            // case 10
		const FaceSigns f = face_tests(ci, v);
		switch (f.sum) {
		case -2:
			if (k == 2 ? interior_test(0, 0, v) : (interior_test(0, 0, v) || interior_test(k ? 1 : 3, 0, v)))
				return 1213 + 8 * k;
			return 1189 + 4 * k;
		case 0: return (f[2 + (int)k] < 0 ? 1261u : 1285u) + 8 * k;
		default:
			if (k == 2 ? interior_test(1, 0, v) : (interior_test(2, 0, v) || interior_test(k ? 3 : 1, 0, v)))
				return 1237 + 8 * k;
			return 1201 + 4 * k;
		}
	}
	case 6: {                                                                            // case 12
		const FaceSigns f = face_tests(ci, v);
		switch (f.sum) {
		case -2: return interior_test((int)((0xDA010Cu >> (2 * k)) & 3), 0, v) ? 1453 + 8 * k : 1357 + 4 * k;
		case 0: return (f[(int)(k >> 1)] < 0 ? 1645u : 1741u) + 8 * k;
		default: return interior_test((int)((0xA7B7E5u >> (2 * k)) & 3), 0, v) ? 1549 + 8 * k : 1405 + 4 * k;
		}
	}
	default: {                                                                           // case 13
		const FaceSigns f = face_tests(165, v);
		const int s = f.sum < 0 ? -f.sum : f.sum;
		if (s == 0) {
			const int kk = ((f[1] < 0) << 1) | (f[5] < 0);
			if (f[0] * f[1] == f[5]) return (uint32_t)(2157 + 12 * kk);
			const int r = interior_test(kk, 1, v);
			return (uint32_t)(2285 + (r ? 10 * kk - 40 * r : 6 * kk));
		}
		if (s == 2) {
			uint32_t off = 1917 + 10 * ((f[0] < 0 ? (uint32_t)(f[2] > 0) : 12u + (f[2] < 0)) +
			                            (f[1] < 0 ? (uint32_t)(f[3] < 0) : 6u + (f[3] > 0)));
			if (f[4] > 0) off += 30;
			return off;
		}
		if (s == 4) {
			uint32_t kk = (uint32_t)(21 + 11 * f[0] + 4 * f[1] + 3 * f[2] + 2 * f[3] + f[4]);
			if (kk >> 4) kk -= (kk & 32 ? 20 : 10);
			return 1845 + 3 * kk;
		}
		return (uint32_t)(1839 + 2 * f[0]);
	}
	}
}

// ---------------------------------------------------------------------------------------------------
// The plan of one active cell: what each edge of its pattern resolves to
// ---------------------------------------------------------------------------------------------------
struct CellPlan {
	uint64_t rank;     // nibble per slot 0..12 (12 = cell centre): rank of the NEW vertex, 0xF = none
	uint32_t visited;  // slots that appear in the pattern (or were marked on the way, MC:973 etc.)
	uint32_t created;  // slots that create a vertex here (a slot may also share another slot's rank)
	uint32_t onpoint;  // subset of created: the vertex lies on a grid point (MC:628 MC33_surfint)
	uint32_t onb;      // ... and that grid point is the edge's second end point
	uint32_t tgt[3];   // byte per edge 0..11, meaningful when visited and rank==0xF: the grid edge whose
	                   // id is used: axis | (dx+1)<<2 | (dy+1)<<4 | (dz+1)<<6, base point = cell + (dx,dy,dz)
	uint16_t poff;     // pattern offset in the table
	uint8_t m, n;      // winding selectors
	uint8_t nnew;      // vertices created by this cell
	uint8_t ntri;      // triangles in the pattern (before the zero-area filter, MC:1235)
	uint8_t zmask;     // corners whose value is exactly 0 (bit k)
};

MC33_HD uint32_t plan_rank(const CellPlan &p, uint32_t e) { return (uint32_t)(p.rank >> (4 * e)) & 15u; }
MC33_HD void plan_set_rank(CellPlan &p, uint32_t e, uint32_t r) {
	p.rank = (p.rank & ~(15ull << (4 * e))) | ((uint64_t)r << (4 * e));
}
// (the word is picked by selects, not by an index: a run-time index into the three words would put the plan in scratch memory)
MC33_HD uint32_t tgt_byte(const uint32_t *tgt /*[3]*/, uint32_t e) {
	const uint32_t w = e < 4u ? tgt[0] : e < 8u ? tgt[1] : tgt[2];
	return (w >> (8 * (e & 3))) & 0xFFu;
}
MC33_HD uint32_t plan_tgt(const CellPlan &p, uint32_t e) { return tgt_byte(p.tgt, e); }
MC33_HD void plan_set_tgt(CellPlan &p, uint32_t e, uint32_t t) {
	const uint32_t sh = 8 * (e & 3), keep = ~(0xFFu << sh), put = t << sh;
	p.tgt[0] = e < 4u ? (p.tgt[0] & keep) | put : p.tgt[0];
	p.tgt[1] = (e >= 4u && e < 8u) ? (p.tgt[1] & keep) | put : p.tgt[1];
	p.tgt[2] = e >= 8u ? (p.tgt[2] & keep) | put : p.tgt[2];
}
MC33_HD uint32_t make_tgt(uint32_t axis, int dx, int dy, int dz) {
	return axis | (uint32_t)(dx + 1) << 2 | (uint32_t)(dy + 1) << 4 | (uint32_t)(dz + 1) << 6;
}

template <typename T>
MC33_HD void plan_visit(CellPlan &p, const Tables &tab, const Params &P, const GridView<T> &G,
                        uint32_t x, uint32_t y, uint32_t z, const VRef &v, uint32_t e) {
	const uint32_t bit = 1u << e;
	p.visited |= bit;
	if (e == 12) {  // MC:1225-1230
		plan_set_rank(p, 12, p.nnew++);
		p.created |= bit;
		return;
	}
	const uint32_t own = edge_own(e), a = edge_a(e), b = edge_b(e);
	if (((own & 1) && x) || ((own & 2) && y) || ((own & 4) && z)) {
		// an earlier cell owns this edge: the reference reads the id from its cache (e.g. MC:789)
		const uint32_t c = corner_code(a);
		plan_set_tgt(p, e, make_tgt(edge_axis(e), (int)(c & 1), (int)((c >> 1) & 1), (int)(c >> 2)));
		return;
	}
	const real_t va = v[(int)a], vb = v[(int)b];
	if (va == 0 || vb == 0) {
		// the vertex sits on a grid point: try the id sources in the reference's order (MC:791-808 ...)
		const uint8_t *ix = tab.rule_index + 4 * e;
		const uint32_t first = (va == 0) ? ix[0] : ix[2], count = (va == 0) ? ix[1] : ix[3];
		for (uint32_t q = 0; q < count; q++) {
			const uint32_t w = tab.rule_words[first + q];
			const uint32_t need = (w >> 2) & 0x7Fu;
			if (((need & 1) && !x) || ((need & 2) && !y) || ((need & 4) && !z) || ((need & 8) && !(x + 1 < P.nx)) ||
			    ((need & 16) && !(y + 1 < P.ny)) || ((need & 32) && x) || ((need & 64) && y))
				continue;
			const uint32_t kind = w & 3u, arg = (w >> 9) & 15u;
			if (kind == 1) {  // another edge of this cell, if already visited
				if (p.visited & (1u << arg)) {
					plan_set_rank(p, e, plan_rank(p, arg));
					plan_set_tgt(p, e, plan_tgt(p, arg));
					return;
				}
				continue;
			}
			uint32_t pass;
			if (w & (1u << 13)) {
				const int fx = (int)((w >> 14) & 3) - 1, fy = (int)((w >> 16) & 3) - 1, fz = (int)((w >> 18) & 3) - 1;
				pass = sign_of(iso_diff(P.iso, (real_t)G.at((uint32_t)((int)x + fx), (uint32_t)((int)y + fy), (uint32_t)((int)z + fz))));
			} else
				pass = sign_of(v[(int)arg]);
			if (!pass) continue;
			const uint32_t t = (w >> 20) & 0xFFu;  // axis | sx+1 | sy+1 | sz+1, same layout as make_tgt
			plan_set_tgt(p, e, t);
			const uint32_t also = w >> 28;
			if (also != 15u) {
				p.visited |= 1u << also;
				plan_set_tgt(p, also, t);
			}
			return;
		}
		plan_set_rank(p, e, p.nnew++);  // MC33_surfint (MC:628)
		p.created |= bit;
		p.onpoint |= bit;
		if (va != 0) p.onb |= bit;
		return;
	}
	plan_set_rank(p, e, p.nnew++);  // regular interpolated vertex (MC:810-816 ...)
	p.created |= bit;
}

template <typename T>
MC33_HD void plan_cell(CellPlan &p, const Tables &tab, const Params &P, const GridView<T> &G,
                       uint32_t x, uint32_t y, uint32_t z, uint32_t i, const VRef &v) {
	uint32_t m, n;
	p.poff = (uint16_t)pattern_offset(tab.lut, i, v, m, n);
	p.m = (uint8_t)m; p.n = (uint8_t)n;
	p.rank = ~0ull;
	p.visited = p.created = p.onpoint = p.onb = 0;
	p.tgt[0] = p.tgt[1] = p.tgt[2] = 0;
	p.nnew = 0; p.ntri = 0;
	uint32_t zm = 0;
	for (int k = 0; k < 8; k++) zm |= (uint32_t)(v[k] == 0) << k;
	p.zmask = (uint8_t)zm;
	uint32_t pos = p.poff, word;
	do {  // MC:780-784
		word = tab.lut[++pos];
		p.ntri++;
		uint32_t w = word;
		for (int k = 0; k < 3; k++) {
			const uint32_t e = w & 15u;
			w >>= 4;
			if (!(p.visited & (1u << e))) plan_visit(p, tab, P, G, x, y, z, v, e);
		}
	} while (word >> 12);
}

// absolute grid edge a plan target byte refers to
struct GridEdge {
	uint32_t axis, x, y, z;
};
MC33_HD GridEdge tgt_edge(uint32_t t, uint32_t x, uint32_t y, uint32_t z) {
	GridEdge g;
	g.axis = t & 3u;
	g.x = (uint32_t)((int)x + (int)((t >> 2) & 3) - 1);
	g.y = (uint32_t)((int)y + (int)((t >> 4) & 3) - 1);
	g.z = (uint32_t)((int)z + (int)((t >> 6) & 3) - 1);
	return g;
}

// Follow a grid edge to the cell that creates its vertex by PLANNING every owner on the way (the round-1 formulation of
// the count pass: no stored plans needed).  The kernels use chase_root / count_triangles_stored on the stored plans
// instead; root_of, slots_differ and count_triangles are kept as the independent statement the host emulator checks
// those against.  w: scratch for 8 values.
struct VertexKey {
	uint32_t x, y, z, rank;
};
template <typename T>
MC33_HD VertexKey root_of(const Tables &tab, const Params &P, const GridView<T> &G, GridEdge g, const VRef &w) {
	VertexKey key = {NO_ID, NO_ID, NO_ID, 15u};
	for (int hop = 0; hop < 64; hop++) {
		const OwnerRef o = owner_of(g.axis, g.x, g.y, g.z);
		const uint32_t i = load_cell(G, P.iso, o.x, o.y, o.z, w);
		if (i == 0 || i == 0xFF) return key;
		CellPlan q;
		plan_cell(q, tab, P, G, o.x, o.y, o.z, i, w);
		if (!(q.visited & (1u << o.e))) return key;
		const uint32_t r = plan_rank(q, o.e);
		if (r != 15u) {
			key.x = o.x; key.y = o.y; key.z = o.z; key.rank = r;
			return key;
		}
		g = tgt_edge(plan_tgt(q, o.e), o.x, o.y, o.z);
	}
	return key;
}

// do pattern slots ea and eb of this cell refer to different vertices?
template <typename T>
MC33_HD bool slots_differ(const CellPlan &p, const Tables &tab, const Params &P, const GridView<T> &G,
                          uint32_t x, uint32_t y, uint32_t z, uint32_t ea, uint32_t eb, const VRef &w) {
	const uint32_t ra = plan_rank(p, ea), rb = plan_rank(p, eb);
	if (ra != 15u && rb != 15u) return ra != rb;
	if (ra != 15u || rb != 15u) return true;  // one created here, the other by an earlier cell
	const uint32_t ta = plan_tgt(p, ea), tb = plan_tgt(p, eb);
	if (ta == tb) return false;
	// two different grid edges share a vertex only when it lies on a common end point with value 0
	const uint32_t ca = (1u << edge_a(ea)) | (1u << edge_b(ea)), cb = (1u << edge_a(eb)) | (1u << edge_b(eb));
	if (!(ca & cb & p.zmask)) return true;
	const VertexKey ka = root_of(tab, P, G, tgt_edge(ta, x, y, z), w);
	const VertexKey kb = root_of(tab, P, G, tgt_edge(tb, x, y, z), w);
	return !(ka.x == kb.x && ka.y == kb.y && ka.z == kb.z && ka.rank == kb.rank);
}

// number of triangles the cell appends (MC:1235 drops triangles with two equal vertex ids)
template <typename T>
MC33_HD uint32_t count_triangles(const CellPlan &p, const Tables &tab, const Params &P, const GridView<T> &G,
                                 uint32_t x, uint32_t y, uint32_t z, const VRef &w) {
	if (!p.zmask) return p.ntri;  // all vertices distinct
	uint32_t pos = p.poff, word, nt = 0;
	do {
		word = tab.lut[++pos];
		const uint32_t e0 = word & 15u, e1 = (word >> 4) & 15u, e2 = (word >> 8) & 15u;
		if (slots_differ(p, tab, P, G, x, y, z, e0, e1, w) && slots_differ(p, tab, P, G, x, y, z, e0, e2, w) &&
		    slots_differ(p, tab, P, G, x, y, z, e1, e2, w))
			nt++;
	} while (word >> 12);
	return nt;
}

// ---------------------------------------------------------------------------------------------------
// Vertex geometry.  r[0..2]: position in grid-index units, r[3..5]: gradient of v = iso - F
// ---------------------------------------------------------------------------------------------------
template <typename T>
MC33_HD void vertex_on_edge(const Params &P, const GridView<T> &G, uint32_t x, uint32_t y, uint32_t z, uint32_t e,
                            const VRef &v, real_t *r) {  // MC:810-816 ... 1212-1220, SURVEY.md Appendix C
	const uint32_t a = edge_a(e), b = edge_b(e), axis = edge_axis(e);
	const uint32_t ca = corner_code(a), cb = corner_code(b);
	const uint32_t cell[3] = {x, y, z}, lim[3] = {P.nx, P.ny, P.nz};
	const real_t va = v[(int)a], vb = v[(int)b];
	const real_t t = va / (va - vb);
	for (uint32_t ax = 0; ax < 3; ax++) {
		if (ax == axis) {
			r[ax] = (real_t)cell[ax] + t;
			r[3 + ax] = vb - va;
			continue;
		}
		const uint32_t off = (ca >> ax) & 1u;
		r[ax] = (real_t)(cell[ax] + off);
		if (off && cell[ax] + 1 < lim[ax]) {
			// central differences across the edge at both end points, blended along the edge
			uint32_t pa[3] = {x + (ca & 1), y + ((ca >> 1) & 1), z + (ca >> 2)};
			uint32_t pb[3] = {x + (cb & 1), y + ((cb >> 1) & 1), z + (cb >> 2)};
			uint32_t lo[3] = {pa[0], pa[1], pa[2]}, hi[3] = {pa[0], pa[1], pa[2]};
			lo[ax]--; hi[ax]++;
			const real_t da = sample_diff(G.at(lo[0], lo[1], lo[2]), G.at(hi[0], hi[1], hi[2]));
			lo[0] = hi[0] = pb[0]; lo[1] = hi[1] = pb[1]; lo[2] = hi[2] = pb[2];
			lo[ax]--; hi[ax]++;
			const real_t db = sample_diff(G.at(lo[0], lo[1], lo[2]), G.at(hi[0], hi[1], hi[2]));
			r[3 + ax] = 0.5f * (da * (1 - t) + db * t);
		} else {
			// one-sided: difference of v across the cell at both end points
			const uint32_t bitax = 1u << ax;
			const real_t da = v[(int)corner_at(ca | bitax)] - v[(int)corner_at(ca & ~bitax)];
			const real_t db = v[(int)corner_at(cb | bitax)] - v[(int)corner_at(cb & ~bitax)];
			r[3 + ax] = da * (1 - t) + db * t;
		}
	}
}

template <typename T>
MC33_HD void vertex_on_point(const Params &P, const GridView<T> &G, uint32_t x, uint32_t y, uint32_t z, real_t *r) {  // MC:628-649
	const uint32_t q[3] = {x, y, z}, lim[3] = {P.nx, P.ny, P.nz};
	r[0] = (real_t)x; r[1] = (real_t)y; r[2] = (real_t)z;
	for (uint32_t ax = 0; ax < 3; ax++) {
		uint32_t lo[3] = {x, y, z}, hi[3] = {x, y, z};
		bool half = false;
		if (q[ax] == 0) hi[ax] = 1;
		else if (q[ax] == lim[ax]) lo[ax] = q[ax] - 1;
		else { lo[ax] = q[ax] - 1; hi[ax] = q[ax] + 1; half = true; }
		const real_t d = sample_diff(G.at(lo[0], lo[1], lo[2]), G.at(hi[0], hi[1], hi[2]));
		r[3 + ax] = half ? 0.5f * d : d;
	}
}

MC33_HD void vertex_centre(uint32_t x, uint32_t y, uint32_t z, const VRef &v, real_t *r) {  // MC:1225-1230
	r[0] = (real_t)x + 0.5f; r[1] = (real_t)y + 0.5f; r[2] = (real_t)z + 0.5f;
	r[3] = v[4] + v[5] + v[6] + v[7] - v[0] - v[1] - v[2] - v[3];
	r[4] = v[1] + v[2] + v[5] + v[6] - v[0] - v[3] - v[4] - v[7];
	r[5] = v[2] + v[3] + v[6] + v[7] - v[0] - v[1] - v[4] - v[5];
}

MC33_HD float inv_sqrt_exact(float f) { return 1.0f / sqrtf(f); }  // MC:70-73 (the reference's portable form)

// b <- A b or A^T b (3x3, row major), products and sums in double, rounded to MC33_real on assignment: the two
// forms of the reference's mult_Abf (UTIL:86-112).  The triangular form skips the zero terms and writes its
// results one by one, which matters for the rounding of nothing but is kept for -0 / non-finite inputs.
MC33_HD void mat_vec(const double *A, real_t *b, bool transposed, bool triangular) {
	if (triangular) {
		if (transposed) {
			b[2] = (real_t)(A[2] * b[0] + A[5] * b[1] + A[8] * b[2]);
			b[1] = (real_t)(A[1] * b[0] + A[4] * b[1]);
			b[0] = (real_t)(A[0] * b[0]);
		} else {
			b[0] = (real_t)(A[0] * b[0] + A[1] * b[1] + A[2] * b[2]);
			b[1] = (real_t)(A[4] * b[1] + A[5] * b[2]);
			b[2] = (real_t)(A[8] * b[2]);
		}
		return;
	}
	double u, v;
	if (transposed) {
		u = A[0] * b[0] + A[3] * b[1] + A[6] * b[2];
		v = A[1] * b[0] + A[4] * b[1] + A[7] * b[2];
		b[2] = (real_t)(A[2] * b[0] + A[5] * b[1] + A[8] * b[2]);
	} else {
		u = A[0] * b[0] + A[1] * b[1] + A[2] * b[2];
		v = A[3] * b[0] + A[4] * b[1] + A[5] * b[2];
		b[2] = (real_t)(A[6] * b[0] + A[7] * b[1] + A[8] * b[2]);
	}
	b[0] = (real_t)u; b[1] = (real_t)v;
}

// world position and unit normal of vertex `id` (MC:485-621).  MODE: the store (Params::store_mode) when the caller knows it
// at compile time - the kernels that are bound by their instruction count are built once per store, and the two 3 x 3
// double matrices of MC33_spnC stay out of the registers of the other three; -1: looked up in P.
template <int MODE = -1>
MC33_HD void store_vertex(const Params &P, real_t *r, real_t *V, float *N, uint32_t id) {
	const int store_mode = MODE < 0 ? P.store_mode : MODE;
	real_t *p = V + 3 * (uint64_t)id;
	if (store_mode == 0) {
		p[0] = r[0]; p[1] = r[1]; p[2] = r[2];
	} else if (store_mode == 3) {  // MC:607-612
		mat_vec(P.A, r, false, P.triangular != 0);
		for (int k = 0; k < 3; k++) p[k] = r[k] + P.O[k];
		mat_vec(P.Ai, r + 3, true, P.triangular != 0);
	} else {
		for (int k = 0; k < 3; k++) p[k] = r[k] * P.D[k] + P.O[k];
		if (store_mode == 2) { r[3] *= P.ca; r[4] *= P.cb; }
	}
	// MC:510-515: the squared length is MC33_real, the inverse root and the normal are float
	float s = inv_sqrt_exact((float)(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]));
	if (P.normal_neg) s = -s;  // MC:509-513
	float *n = N + 3 * (uint64_t)id;
	n[0] = s * (float)r[3]; n[1] = s * (float)r[4]; n[2] = s * (float)r[5];
}

// ---------------------------------------------------------------------------------------------------
// Work records produced by the count pass and consumed by the emit pass
// ---------------------------------------------------------------------------------------------------
// One 16-byte entry per active cell, stored so that the entries of one row segment are contiguous and
// sorted by x.  Neighbours look each other up through the row-segment directory.
struct Entry {
	uint32_t w0;  // x within the segment (8) | sign index (8) | pattern offset (12) | nnew (4)
	uint32_t w1;  // vertex offset inside the row segment (16) | triangle offset (16)
	uint32_t w2;  // ranks of edges 0..7  (nibbles, 0xF = the cell creates no vertex for that edge)
	uint32_t w3;  // ranks of edges 8..11 (16) | triangles appended by the cell (4) << 16 | slow flag << 20 | tested flag << 21 |
	              // rank of the cell-centre vertex (4) << 24
};
constexpr uint32_t ENTRY_SLOW = 1u << 20;    // the generic per-cell code writes the cell (k_emit_slow)
constexpr uint32_t ENTRY_TESTED = 1u << 21;  // pattern chosen by the face / interior tests, written by the fast emit passes (see cell_is_tested)
constexpr uint32_t ENTRY_COUNT = 1u << 22;   // slow record whose triangles are still to be counted by vertex identity (k_slow_count)
MC33_HD Entry make_entry(uint32_t xl, uint32_t i, const CellPlan &p, uint32_t nt, uint32_t voff, uint32_t toff, bool slow) {
	Entry e;
	e.w0 = xl | i << 8 | (uint32_t)p.poff << 16 | (uint32_t)p.nnew << 28;
	e.w1 = voff | toff << 16;
	e.w2 = (uint32_t)p.rank;
	e.w3 = ((uint32_t)(p.rank >> 32) & 0xFFFFu) | nt << 16 | (slow ? ENTRY_SLOW : 0u) | plan_rank(p, 12) << 24;
	return e;
}
MC33_HD uint32_t entry_rank(const Entry &e, uint32_t edge) {
	return edge < 8 ? (e.w2 >> (4 * edge)) & 15u : (e.w3 >> (4 * (edge - 8))) & 15u;
}
MC33_HD uint32_t entry_nnew(const Entry &e) { return e.w0 >> 28; }
MC33_HD uint32_t entry_ntri(const Entry &e) { return (e.w3 >> 16) & 15u; }
MC33_HD uint32_t entry_rank_centre(const Entry &e) { return (e.w3 >> 24) & 15u; }

// Fast path.  A cell is FAST when it owns exactly the three edges meeting at its far corner (x,y,z >= 1),
// its sign index selects a pattern without tests (table word group 0: MC33 cases 1,2,5,8,9,11,14,
// MC:694-696) and none of its corners equals the isovalue.  Everything about such a cell follows from
// the 8-bit sign index alone, precomputed once per library load:
//   fast[i] = pattern offset (12) | triangles (4) << 12 | nnew (4) << 16 | rank of edge 5 (4) << 20 |
//             rank of edge 6 (4) << 24 | rank of edge 10 (4) << 28          (0xFFFFFFFF: not fast)
constexpr uint32_t FAST_NONE = 0xFFFFFFFFu;
MC33_HD Entry make_fast_entry(uint32_t xl, uint32_t i, uint32_t f, uint32_t voff, uint32_t toff) {
	Entry e;
	e.w0 = xl | i << 8 | (f & 0xFFFu) << 16 | ((f >> 16) & 15u) << 28;
	e.w1 = voff | toff << 16;
	e.w2 = 0xF00FFFFFu | ((f >> 20) & 15u) << 20 | ((f >> 24) & 15u) << 24;
	e.w3 = 0xF0FFu | ((f >> 28) & 15u) << 8 | ((f >> 12) & 15u) << 16 | 15u << 24;  // (no pattern without tests has a centre vertex)
	return e;
}
// placeholder written by the sweep for a cell the slow kernel will plan
MC33_HD Entry make_pending_entry(uint32_t xl, uint32_t i) {
	Entry e;
	e.w0 = xl | i << 8;
	e.w1 = 0;
	e.w2 = 0xFFFFFFFFu;
	e.w3 = 0xFFFFu | ENTRY_SLOW | 15u << 24;
	return e;
}

// Storage of the work records.  In registers a record is an Entry; in HBM it is split in two 8-byte halves kept in two
// arrays with the same index:
//   A: x in the segment (8) | sign index (8) | nnew (4) | triangles (4) | slow flag (1 << 24) | tested flag (1 << 25) | count flag (1 << 26)   ;   vertex offset (16) | triangle offset (16)
//   B: ranks of edges 0..7   ;   ranks of edges 8..11 (16) | pattern offset (12) << 16 | rank of the centre vertex (4) << 28
// Half B of a FAST record is a function of its sign index (fast_b_table): it is never written or read - a fast
// record costs 8 bytes of HBM traffic per pass instead of 16 (writes are what the passes after the sweep pay for most).
struct EntryA { uint32_t a0, a1; };
struct EntryB { uint32_t b0, b1; };
constexpr uint32_t ENTRYA_SLOW = 1u << 24, ENTRYA_TESTED = 1u << 25, ENTRYA_COUNT = 1u << 26;
MC33_HD EntryA entry_a(const Entry &e) {
	return EntryA{(e.w0 & 0xFFFFu) | (e.w0 >> 28) << 16 | ((e.w3 >> 16) & 15u) << 20 | ((e.w3 >> 20) & 7u) << 24, e.w1};
}
MC33_HD EntryB entry_b(const Entry &e) { return EntryB{e.w2, (e.w3 & 0xFFFFu) | ((e.w0 >> 16) & 0xFFFu) << 16 | (e.w3 >> 24) << 28}; }
MC33_HD Entry entry_join(const EntryA &a, const EntryB &b) {
	Entry e;
	e.w0 = (a.a0 & 0xFFFFu) | ((b.b1 >> 16) & 0xFFFu) << 16 | ((a.a0 >> 16) & 15u) << 28;
	e.w1 = a.a1;
	e.w2 = b.b0;
	e.w3 = (b.b1 & 0xFFFFu) | ((a.a0 >> 20) & 15u) << 16 | ((a.a0 >> 24) & 7u) << 20 | (b.b1 >> 28) << 24;
	return e;
}
MC33_HD uint32_t entrya_nnew(const EntryA &a) { return (a.a0 >> 16) & 15u; }
MC33_HD uint32_t entrya_ntri(const EntryA &a) { return (a.a0 >> 20) & 15u; }
// half B of the fast record of every sign index (zeros where the index is not fast)
inline void fast_b_table(const uint32_t *fast /*[256]*/, EntryB *out /*[256]*/) {
	for (uint32_t i = 0; i < 256; i++) {
		out[i] = EntryB{0, 0};
		if (fast[i] != FAST_NONE) out[i] = entry_b(make_fast_entry(0, i, fast[i], 0, 0));
	}
}
// a record from its halves: half B from the table unless k_slow_plan made it
MC33_HD Entry load_entry(const EntryA *ea, const EntryB *eb, const EntryB *fast_b, uint32_t ri) {
	const EntryA a = ea[ri];
	const EntryB b = (a.a0 & (ENTRYA_SLOW | ENTRYA_TESTED)) ? eb[ri] : fast_b[(a.a0 >> 8) & 0xFFu];
	return entry_join(a, b);
}

// Third part of a SLOW record (16 bytes, same index): what its plan says about the slots that take their vertex from
// elsewhere, so that nobody has to make the plan of a cell twice - the count pass and the emit pass follow chains of
// such references from record to record (chase_root) instead of re-planning every owner on the way.
//   tgt[3]: byte per edge 0..11, as CellPlan::tgt   ;   masks: visited (12) | created (12) << 12 | corners equal to the isovalue (8) << 24
//   (the centre slot 12 is visited and created exactly when it has a rank)
struct EntryC { uint32_t tgt[3], masks; };
MC33_HD EntryC entry_c(const CellPlan &p) {
	return EntryC{{p.tgt[0], p.tgt[1], p.tgt[2]}, (p.visited & 0xFFFu) | (p.created & 0xFFFu) << 12 | (uint32_t)p.zmask << 24};
}
// the plan back from the record (all but onpoint / onb, which plan_restore_points derives from the corner values)
MC33_HD void plan_restore(CellPlan &p, const uint16_t *lut, const Entry &en, const EntryC &c) {
	const uint32_t i = (en.w0 >> 8) & 0xFFu, r12 = entry_rank_centre(en), has12 = r12 != 15u ? 1u << 12 : 0u;
	p.rank = (uint64_t)en.w2 | (uint64_t)(en.w3 & 0xFFFFu) << 32 | (uint64_t)r12 << 48 | 0xFFF0000000000000ull;
	p.visited = (c.masks & 0xFFFu) | has12;
	p.created = ((c.masks >> 12) & 0xFFFu) | has12;
	p.onpoint = p.onb = 0;
	p.tgt[0] = c.tgt[0]; p.tgt[1] = c.tgt[1]; p.tgt[2] = c.tgt[2];
	p.poff = (uint16_t)((en.w0 >> 16) & 0xFFFu);
	p.n = (uint8_t)(((lut[(i & 0x80) ? (i ^ 0xFF) : i] >> 11) ^ (i >> 7) ^ 1u) & 1u);  // as pattern_offset
	p.m = (uint8_t)!p.n;
	p.nnew = (uint8_t)entry_nnew(en);
	p.ntri = 0;
	p.zmask = (uint8_t)(c.masks >> 24);
}
template <typename V>
MC33_HD void plan_restore_points(CellPlan &p, const V &v) {  // as plan_visit sets them (MC:628: the vertex lies on a grid point)
	for (uint32_t e = 0; e < 12; e++) {
		if (!(p.created & (1u << e))) continue;
		const real_t va = v[(int)edge_a(e)], vb = v[(int)edge_b(e)];
		if (va == 0 || vb == 0) {
			p.onpoint |= 1u << e;
			if (va != 0) p.onb |= 1u << e;
		}
	}
}

inline void build_fast_table(const uint16_t *lut, uint32_t *fast /*[256]*/) {
	Tables tab{lut, nullptr, nullptr};
	Params P{};
	P.nx = P.ny = P.nz = 1u << 20;
	GridView<real_t> G{nullptr, 0, 0, 0};
	for (uint32_t i = 0; i < 256; i++) {
		fast[i] = FAST_NONE;
		if (i == 0 || i == 255) continue;
		const uint32_t c = lut[(i & 0x80) ? (i ^ 0xFF) : i];
		if (c >> 12) continue;  // needs face / interior tests
		real_t vb[8];
		for (int k = 0; k < 8; k++) vb[k] = ((i >> (7 - k)) & 1) ? -1.0f : 1.0f;
		CellPlan p;
		plan_cell(p, tab, P, G, 1, 1, 1, i, VRef{vb, 1});  // interior cell, no corner equal to iso: no rule is consulted
		fast[i] = (uint32_t)p.poff | (uint32_t)p.ntri << 12 | (uint32_t)p.nnew << 16 | plan_rank(p, 5) << 20 |
		          plan_rank(p, 6) << 24 | plan_rank(p, 10) << 28;
	}
}

// What an interior cell without a corner equal to the isovalue makes of the pattern that starts at table offset p: it
// creates the vertices of its edges 5, 6, 10 and of its centre (slot 12) in the order the pattern names them (MC:780-784)
//   info[p] = rank of edge 5 | edge 6 << 4 | edge 10 << 8 | centre << 12 (15: not in the pattern) | triangles << 16 | new vertices << 20
// (filled for every offset; only pattern starts are ever looked up)
inline void build_pattern_info(const uint16_t *lut, uint32_t n, uint32_t *info) {
	for (uint32_t p = 0; p < n; p++) {
		uint32_t visited = 0, nnew = 0, ntri = 0, rank[13], pos = p, word = 0;
		for (int e = 0; e < 13; e++) rank[e] = 15u;
		do {
			if (++pos >= n) break;
			word = lut[pos];
			ntri++;
			uint32_t w = word;
			for (int k = 0; k < 3; k++, w >>= 4) {
				const uint32_t e = w & 15u;
				if (e > 12u || (visited & (1u << e))) continue;
				visited |= 1u << e;
				if (e == 5u || e == 6u || e == 10u || e == 12u) rank[e] = nnew++;
			}
		} while ((word >> 12) && ntri < 15u);
		info[p] = rank[5] | rank[6] << 4 | rank[10] << 8 | rank[12] << 12 | ntri << 16 | nnew << 20;
	}
}
// record of a TESTED cell from its pattern offset and info word (same fields as make_entry fills from a CellPlan)
MC33_HD Entry make_tested_entry(uint32_t xl, uint32_t i, uint32_t poff, uint32_t info, uint32_t voff, uint32_t toff) {
	Entry e;
	e.w0 = xl | i << 8 | poff << 16 | ((info >> 20) & 15u) << 28;
	e.w1 = voff | toff << 16;
	e.w2 = 0xF00FFFFFu | (info & 15u) << 20 | ((info >> 4) & 15u) << 24;
	e.w3 = 0xF0FFu | ((info >> 8) & 15u) << 8 | ((info >> 16) & 15u) << 16 | ENTRY_TESTED | ((info >> 12) & 15u) << 24;
	return e;
}

// A TESTED cell: its sign index needs the face / interior tests (so the sweep could not finish it), but it is an interior
// cell (it owns exactly edges 5, 6, 10 and, in some patterns, the centre vertex) and none of its corners equals the
// isovalue - no alias, every other edge's vertex is a regular vertex of the edge's owner.  Once k_slow_plan has chosen
// its pattern, the fast emit passes write it like a fast cell: pattern offset and ranks come from half B of the record.
MC33_HD bool cell_is_tested(const CellPlan &p, uint32_t x, uint32_t y, uint32_t z) { return x && y && z && !p.zmask; }

// Directory of the row segments.  A row segment = the cells (x in [256 s, 256 s + 256), y, z).  Records
// are STORED in the order [z][s][y] (the 63 rows a wave handles are contiguous: coalesced writes), while
// the prefix sums run over them in the reference's sweep order [z][y][s] (segment_sweep_to_store).
struct alignas(64) SegDir {
	// One cache line per row segment, laid out so that ONE 16-byte read answers "which record is cell x": for each of
	// the four 64-cell words k of the segment
	//   q[k] = { activity mask of the word (lo, hi), index of the first work record of the word's cells, nent }
	// nent = number of records of the whole segment (bit 31: the sweep left cells for the slow kernel).  Written only
	// for segments that hold records; nobody looks at the others.
	uint32_t q[4][4];
};
MC33_HD uint32_t segdir_first(const SegDir &d) { return d.q[0][2]; }
MC33_HD uint32_t segdir_nent(const SegDir &d) { return d.q[0][3]; }
MC33_HD SegDir make_segdir(uint32_t first, uint32_t nent_flags, const uint64_t *mask /*[4]*/) {
	SegDir d;
	uint32_t at = first;
	for (int k = 0; k < 4; k++) {
		d.q[k][0] = (uint32_t)mask[k]; d.q[k][1] = (uint32_t)(mask[k] >> 32); d.q[k][2] = at; d.q[k][3] = nent_flags;
		at += (uint32_t)__builtin_popcountll(mask[k]);
	}
	return d;
}
struct SegBase {
	uint32_t vbase, tbase;  // exclusive scans of the per-segment counts, in sweep order
};
constexpr uint32_t SEG_DIRTY = 1u << 31;

MC33_HD uint64_t segment_index(const Params &P, uint32_t x, uint32_t y, uint32_t z) {
	return ((uint64_t)(z - P.zs) * P.nseg + x / SEG_CELLS) * P.ny + y;
}
// position q in sweep order ([z][y][s]) -> storage index
MC33_HD uint64_t segment_sweep_to_store(const Params &P, uint64_t q) {
	const uint64_t zy = q / P.nseg, sg = q % P.nseg;
	const uint64_t z = zy / P.ny, y = zy % P.ny;
	return (z * P.nseg + sg) * P.ny + y;
}
struct SegCoord {
	uint32_t xbase, y, z;
};
MC33_HD SegCoord segment_coord(const Params &P, uint32_t s) {
	SegCoord c;
	c.y = s % P.ny;
	const uint32_t t = s / P.ny;
	c.xbase = (t % P.nseg) * SEG_CELLS;
	c.z = t / P.nseg + P.zs;
	return c;
}

template <typename T>
struct EmitCtx;
template <typename T>
MC33_HD Entry ctx_entry(const EmitCtx<T> &c, uint32_t ri);

template <typename T>
struct EmitCtx {
	Tables tab;
	Params P;
	GridView<T> G;
	const SegBase *seg_base;
	const SegDir *seg_dir;
	const EntryA *entries_a;    // work records, half A (all) and half B (slow records only)
	const EntryB *entries_b;
	const EntryC *entries_c;    // plans of the slow records
	const EntryB *fast_b;       // [256] half B of the fast records
	bool fast_b_in_lds;         // ... the table is a copy in LDS (the fast emit kernels): read as such, see fast_half_b
	const uint32_t *entry_seg;  // row segment of each entry
	real_t *V;
	float *N;
	uint32_t *Tri;
	// z-slab decomposition: cells below z_emit are ghosts (counted so that ids of the slab interface
	// can be looked up, but written by the rank below); local vertex k is stored at k - v_skip and is
	// known globally as k + id_delta; local triangle k is stored at k - t_skip.
	uint32_t z_emit, v_skip, t_skip, id_delta;
};

// Half B of a fast record from the table.  When the table sits in LDS it must be READ as LDS: through the generic pointer
// it is a flat load, which counts on both wait counters - every wait for it is a wait for all global loads in flight - and a
// `stored ? entries_b[i] : fast_b[k]` becomes ONE flat load through a selected pointer.  An LDS read and a global load
// behind a branch cannot be merged.
template <typename T>
MC33_HD EntryB fast_half_b(const EmitCtx<T> &c, uint32_t sign_index) {
#if defined(__HIP_DEVICE_COMPILE__)
	if (c.fast_b_in_lds) {
		typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
		lds_u32 p = (lds_u32)(const uint32_t *)c.fast_b + 2u * sign_index;
		return EntryB{p[0], p[1]};
	}
#endif
	return c.fast_b[sign_index];
}
template <typename T>
MC33_HD EntryB ctx_half_b(const EmitCtx<T> &c, const EntryA &a, uint32_t ri) {
	EntryB b = fast_half_b(c, (a.a0 >> 8) & 0xFFu);
	if (a.a0 & (ENTRYA_SLOW | ENTRYA_TESTED)) b = c.entries_b[ri];
	return b;
}
template <typename T>
MC33_HD Entry ctx_entry(const EmitCtx<T> &c, uint32_t ri) {
	const EntryA a = c.entries_a[ri];
	return entry_join(a, ctx_half_b(c, a, ri));
}

// per-segment counts packed in one word: vertices (<= 13*256) | triangles (<= 12*256) << 16
MC33_HD uint32_t seg_pack(uint32_t nv, uint32_t nt) { return nv | nt << 16; }

// work record of the active cell (x,y,z), through the directory (no search: the activity mask of the cell's word
// gives the rank of the cell among the word's records); w = the directory word q[xl >> 6] of the cell's row segment
struct DirWord {
	uint32_t mlo, mhi, first, nent;
};
template <typename T>
MC33_HD DirWord dir_word(const EmitCtx<T> &c, uint64_t s, uint32_t xl) {
	const uint32_t *q = c.seg_dir[s].q[xl >> 6];
	return DirWord{q[0], q[1], q[2], q[3]};
}
MC33_HD uint32_t record_rank(const DirWord &w, uint32_t xl) {
	const uint64_t m = (uint64_t)w.mhi << 32 | w.mlo;
	return w.first + (uint32_t)__builtin_popcountll(m & ((1ull << (xl & 63u)) - 1ull));
}
template <typename T>
MC33_HD uint32_t find_record(const EmitCtx<T> &c, uint64_t s, uint32_t xl) {
	const DirWord w = dir_word(c, s, xl);
	const uint64_t m = (uint64_t)w.mhi << 32 | w.mlo;
	if (!((m >> (xl & 63u)) & 1ull)) return NO_ID;
	return record_rank(w, xl);
}

// The vertex a grid edge resolves to, followed from record to record: the record of the cell that creates it, the
// vertex's rank among that cell's new vertices, and where that cell's vertices begin in its row segment.  rec == NO_ID:
// no vertex.  Every hop is a directory word + the owner's record (+ its stored plan if it is a slow one); no samples are
// read and no plan is made (round 1 re-planned every owner on the way: load_cell + plan_cell per hop).
struct RootRef {
	uint32_t rec, rank, seg, voff;
};
template <typename T>
MC33_HD RootRef chase_root(const EmitCtx<T> &c, GridEdge g, const VRef &w) {
	const RootRef none = {NO_ID, 15u, 0u, 0u};
	for (int hop = 0; hop < 64; hop++) {
		const OwnerRef o = owner_of(g.axis, g.x, g.y, g.z);
		const uint64_t s = segment_index(c.P, o.x, o.y, o.z);
		// Only with iso = -0.0 can a rule name an edge whose owner is not cut (for the reference that is a read of whatever
		// an earlier slice or call left in its id caches; here it is "no vertex").  Such a row segment may hold no record
		// at all, and its directory entry is then whatever an earlier extraction left: do not look at it.
		if (c.P.negzero_iso) {
			const uint32_t oi = load_cell(c.G, c.P.iso, o.x, o.y, o.z, w);
			if (oi == 0 || oi == 0xFF) return none;
		}
		const uint32_t ri = find_record(c, s, o.x % SEG_CELLS);
		if (ri == NO_ID) return none;
		const EntryA ea = c.entries_a[ri];
		const Entry e = entry_join(ea, ctx_half_b(c, ea, ri));
		const uint32_t r = entry_rank(e, o.e);
		if (r != 15u) return RootRef{ri, r, (uint32_t)s, ea.a1 & 0xFFFFu};
		if (!(ea.a0 & ENTRYA_SLOW)) return none;  // a fast or tested cell creates the vertices of its cut edges itself: the edge is not in its pattern
		const EntryC pc = c.entries_c[ri];
		if (!(pc.masks & (1u << o.e))) return none;  // (owned edges are 0..11)
		g = tgt_edge(tgt_byte(pc.tgt, o.e), o.x, o.y, o.z);
	}
	return none;
}
MC33_HD uint64_t root_key(const RootRef &r) { return r.rec == NO_ID ? ~0ull : (uint64_t)r.rec << 4 | r.rank; }

// The roots of SEVERAL pattern slots of one cell, the first hop of all of them made together (round 5).  A slow record walks up
// to nine foreign slots; chased one after the other that is nine times directory word -> owner's record (-> its half B) -> ...,
// some twenty dependent round trips per record, and a kernel of such threads does little but wait.  Where a slot ends at its FIRST
// owner - the cell that owns the grid edge created the vertex, the usual case next to a smooth surface - one hop settles it, so: the directory words of all wanted slots are asked for together, then the owners'
// records, then (slow / tested owners only) their halves B; a slot that is not settled by then - an alias: the owner points on -
// goes through chase_root as before.  Six slots at a time (the words of twelve would not fit the registers of the kernels
// that call this).  `want`: the slots to resolve (visited, no rank of their own, not the centre); self_s / self_ri: the cell's own
// row segment and record - addresses that are safe to read for the slots that are not wanted.
// sink(e, k, root): called once per wanted slot e (the k-th of its group), group by group; sink.group_done(): behind every group of six (a place to ask for
// what the group's results lead to - the emit pass fetches the six row-segment bases together there).
#if defined(__HIP_DEVICE_COMPILE__)
#define MC33_UNROLL _Pragma("unroll")
#else
#define MC33_UNROLL
#endif
template <typename T, typename Sink>
MC33_HD void roots_together(const EmitCtx<T> &c, const CellPlan &p, uint32_t x, uint32_t y, uint32_t z, uint32_t want, uint64_t self_s,
                            uint32_t self_ri, const VRef &w, Sink &sink) {
	MC33_UNROLL
	for (int g0 = 0; g0 < 12; g0 += 6) {
		if (!((want >> g0) & 63u)) continue;
		GridEdge ge[6];
		OwnerRef ow[6];
		uint32_t sg[6];
		DirWord dw[6];
		bool need[6];
		MC33_UNROLL
		for (int k = 0; k < 6; k++) {  // round 1: the directory words
			const uint32_t e = (uint32_t)(g0 + k);
			need[k] = ((want >> e) & 1u) != 0u;
			ge[k] = tgt_edge(plan_tgt(p, e), x, y, z);
			ow[k] = need[k] ? owner_of(ge[k].axis, ge[k].x, ge[k].y, ge[k].z) : OwnerRef{x, y, z, 0u};
			sg[k] = need[k] ? (uint32_t)segment_index(c.P, ow[k].x, ow[k].y, ow[k].z) : (uint32_t)self_s;
			dw[k] = dir_word(c, sg[k], ow[k].x % SEG_CELLS);
		}
		uint32_t ri[6];
		EntryA ea[6];
		MC33_UNROLL
		for (int k = 0; k < 6; k++) {  // round 2: the owners' records
			const uint32_t xl = ow[k].x % SEG_CELLS;
			const uint64_t m = (uint64_t)dw[k].mhi << 32 | dw[k].mlo;
			ri[k] = (need[k] && ((m >> (xl & 63u)) & 1ull)) ? record_rank(dw[k], xl) : NO_ID;
			ea[k] = c.entries_a[ri[k] == NO_ID ? self_ri : ri[k]];
		}
		EntryB eb[6];
		MC33_UNROLL
		for (int k = 0; k < 6; k++) {  // round 3: half B - from the sign index for a fast owner, stored for a tested or slow one
			eb[k] = fast_half_b(c, (ea[k].a0 >> 8) & 0xFFu);
			if (ri[k] != NO_ID && (ea[k].a0 & (ENTRYA_SLOW | ENTRYA_TESTED))) eb[k] = c.entries_b[ri[k]];
		}
		MC33_UNROLL
		for (int k = 0; k < 6; k++) {
			if (!need[k]) continue;
			RootRef r = {NO_ID, 15u, 0u, 0u};
			if (ri[k] != NO_ID) {
				const uint32_t rk = entry_rank(entry_join(ea[k], eb[k]), ow[k].e);
				if (rk != 15u) r = RootRef{ri[k], rk, sg[k], ea[k].a1 & 0xFFFFu};
				else if (ea[k].a0 & ENTRYA_SLOW) r = chase_root(c, ge[k], w);  // the owner points on: the walk, from the start (rare)
			}
			sink((uint32_t)(g0 + k), k, r);
		}
		sink.group_done();
	}
}

// Roots of the slots of one cell, each chased at most once (count pass: a slot is compared with up to 2 x 12 others).
// p == nullptr: nothing is remembered.
struct RootMemo {
	uint64_t *p;
	int stride;
	uint32_t have;
};
template <typename T>
MC33_HD uint64_t slot_root(const EmitCtx<T> &c, const CellPlan &p, uint32_t x, uint32_t y, uint32_t z, uint32_t e, const VRef &w, RootMemo &memo) {
	if (memo.p && (memo.have & (1u << e))) return memo.p[e * memo.stride];
	const uint64_t k = root_key(chase_root(c, tgt_edge(plan_tgt(p, e), x, y, z), w));
	if (memo.p) { memo.p[e * memo.stride] = k; memo.have |= 1u << e; }
	return k;
}
// do pattern slots ea and eb of this cell refer to different vertices?  (slots_differ on the stored plans)
template <typename T>
MC33_HD bool slots_differ_stored(const EmitCtx<T> &c, const CellPlan &p, uint32_t x, uint32_t y, uint32_t z, uint32_t ea, uint32_t eb,
                                 const VRef &w, RootMemo &memo) {
	const uint32_t ra = plan_rank(p, ea), rb = plan_rank(p, eb);
	if (ra != 15u && rb != 15u) return ra != rb;
	if (ra != 15u || rb != 15u) return true;  // one created here, the other by an earlier cell
	const uint32_t ta = plan_tgt(p, ea), tb = plan_tgt(p, eb);
	if (ta == tb) return false;
	// two different grid edges share a vertex only when it lies on a common end point with value 0
	const uint32_t ca = (1u << edge_a(ea)) | (1u << edge_b(ea)), cb = (1u << edge_a(eb)) | (1u << edge_b(eb));
	if (!(ca & cb & p.zmask)) return true;
	return slot_root(c, p, x, y, z, ea, w, memo) != slot_root(c, p, x, y, z, eb, w, memo);
}
// number of triangles the cell appends (MC:1235 drops triangles with two equal vertex ids), from the stored plans
// (self_s, self_ri: the cell's row segment and record, see roots_together; self_ri == NO_ID: every slot chased by itself, as until round 5)
template <typename T>
MC33_HD uint32_t count_triangles_stored(const EmitCtx<T> &c, const CellPlan &p, uint32_t x, uint32_t y, uint32_t z, const VRef &w, RootMemo &memo,
                                        uint64_t self_s = 0, uint32_t self_ri = NO_ID) {
	if (p.zmask && memo.p && self_ri != NO_ID && !c.P.negzero_iso) {
		// the slots whose identity can matter - foreign ones with an end point on a corner that equals the isovalue - resolved
		// together and put into the memo, so that the pair tests below find them there
		uint32_t want = 0;
		for (uint32_t e = 0; e < 12; e++)
			if ((p.visited & (1u << e)) && plan_rank(p, e) == 15u && (((1u << edge_a(e)) | (1u << edge_b(e))) & p.zmask)) want |= 1u << e;
		if (want) {
			struct MemoSink {
				RootMemo &m;
				MC33_HD void operator()(uint32_t e, int, const RootRef &r) { m.p[e * m.stride] = root_key(r); }
				MC33_HD void group_done() {}
			} sink{memo};
			roots_together(c, p, x, y, z, want, self_s, self_ri, w, sink);
			memo.have |= want;
		}
	}
	uint32_t pos = p.poff, word, nt = 0;
	do {
		word = c.tab.lut[++pos];
		const uint32_t e0 = word & 15u, e1 = (word >> 4) & 15u, e2 = (word >> 8) & 15u;
		if (!p.zmask || (slots_differ_stored(c, p, x, y, z, e0, e1, w, memo) && slots_differ_stored(c, p, x, y, z, e0, e2, w, memo) &&
		                 slots_differ_stored(c, p, x, y, z, e1, e2, w, memo)))
			nt++;
	} while (word >> 12);
	return nt;
}

// Emit pass for one SLOW record: writes its NEW vertices and its triangles.  The plan comes back from the record
// (plan_restore); v, w: 8-value scratch arrays; ids: 13-slot scratch.
template <typename T>
MC33_HD void emit_cell(const EmitCtx<T> &c, uint32_t entry_index, const VRef &v, const VRef &w, const URef &ids) {
	const Entry en = ctx_entry(c, entry_index);
	if (!(en.w3 & ENTRY_SLOW)) return;  // (a tested cell that k_slow_plan found on the slow list: the fast emit passes write it)
	const uint32_t s = c.entry_seg[entry_index];
	const SegCoord sc = segment_coord(c.P, s);
	const uint32_t y = sc.y, z = sc.z;
	if (z < c.z_emit) return;
	const uint32_t x = sc.xbase + (en.w0 & 0xFFu);
	const SegBase sb = c.seg_base[s];
	const uint32_t vbase = sb.vbase + (en.w1 & 0xFFFFu);
	uint32_t tpos = sb.tbase + (en.w1 >> 16) - c.t_skip;
	load_cell(c.G, c.P.iso, x, y, z, v);
	CellPlan p;
	plan_restore(p, c.tab.lut, en, c.entries_c[entry_index]);
	plan_restore_points(p, v);
	// ids of all slots of the pattern; NEW vertices are written on the way
	// (Round 5: the foreign slots resolved together first - roots_together, then the bases of their row segments together.
	// Bit-identical and slower: k_emit_slow 0.97 -> 1.39 ms on the 2 M slow records of a 1024^3 uchar field with an integer
	// isovalue - 142 registers instead of 88, three waves per SIMD instead of five, and on such a field most owners are slow
	// cells themselves that point on, so the first hop made for all slots at once is made again by the walk.  The count pass,
	// which resolves only the slots at a corner equal to the isovalue, keeps the batched form: the tail of that field 1.32 -> 1.23 ms,
	// profiles/r05_integer_iso.txt.)
	for (uint32_t e = 0; e < 13; e++) {
		if (!(p.visited & (1u << e))) continue;
		const uint32_t r = plan_rank(p, e);
		if (r != 15u) {
			ids[(int)e] = vbase + r;
			if (p.created & (1u << e)) {
				real_t g[6];
				if (e == 12) vertex_centre(x, y, z, v, g);
				else if (p.onpoint & (1u << e)) {
					const uint32_t cc = corner_code((p.onb & (1u << e)) ? edge_b(e) : edge_a(e));
					vertex_on_point(c.P, c.G, x + (cc & 1), y + ((cc >> 1) & 1), z + (cc >> 2), g);
				} else
					vertex_on_edge(c.P, c.G, x, y, z, e, v, g);
				store_vertex(c.P, g, c.V, c.N, vbase + r - c.v_skip);
			}
		} else {
			const RootRef root = chase_root(c, tgt_edge(plan_tgt(p, e), x, y, z), w);
			ids[(int)e] = root.rec == NO_ID ? NO_ID : c.seg_base[root.seg].vbase + root.voff + root.rank;
		}
	}
	RootMemo memo{nullptr, 0, 0u};
	uint32_t pos = p.poff, word;
	do {  // MC:780-784, 1235-1250
		word = c.tab.lut[++pos];
		uint32_t ti[3];
		const uint32_t e2 = word & 15u, e1 = (word >> 4) & 15u, e0 = (word >> 8) & 15u;
		ti[2] = ids[(int)e2];
		ti[1] = ids[(int)e1];
		ti[0] = ids[(int)e0];
		// MC:1235 on the ids - except for iso = -0.0, where ids may be "no vertex" (see chase_root): the triangle slots
		// were counted by vertex identity (count_triangles_stored), and the same test decides here, so that every counted slot is
		// written and the output is a function of the input alone
		const bool keep = c.P.negzero_iso ? (slots_differ_stored(c, p, x, y, z, e2, e1, w, memo) && slots_differ_stored(c, p, x, y, z, e2, e0, w, memo) &&
		                                     slots_differ_stored(c, p, x, y, z, e1, e0, w, memo))
		                                  : (ti[0] != ti[1] && ti[0] != ti[2] && ti[1] != ti[2]);
		if (keep) {
			uint32_t *t = c.Tri + 3 * (uint64_t)tpos++;
			const bool swap = (p.n != 0) != (c.P.normal_neg != 0);  // MC:1246-1250
			t[0] = (swap ? ti[1] : ti[0]) + c.id_delta; t[1] = (swap ? ti[0] : ti[1]) + c.id_delta; t[2] = ti[2] + c.id_delta;
		}
	} while (word >> 12);
}

// --- fast emit ---------------------------------------------------------------------------------------
// vertex of one of the three owned edges (5: z-edge, 6: y-edge, 10: x-edge, all through corner 6) of an
// interior cell; same arithmetic as vertex_on_edge, written out so that every index is static
template <typename T, int E>
MC33_HD void fast_owned_vertex(const EmitCtx<T> &c, uint32_t x, uint32_t y, uint32_t z, const real_t *v, uint32_t id) {
	const Params &P = c.P;
	const GridView<T> &G = c.G;
	real_t r[6];
	const real_t v6 = v[6];
	if (E == 5) {
		const real_t va = v[5], t = va / (va - v6);
		r[0] = (real_t)(x + 1); r[1] = (real_t)(y + 1); r[2] = (real_t)z + t;
		r[3] = (x + 1 < P.nx) ? 0.5f * (sample_diff(G.at(x, y + 1, z), G.at(x + 2, y + 1, z)) * (1 - t) +
		                               sample_diff(G.at(x, y + 1, z + 1), G.at(x + 2, y + 1, z + 1)) * t)
		                      : (v[5] - v[1]) * (1 - t) + (v6 - v[2]) * t;
		r[4] = (y + 1 < P.ny) ? 0.5f * (sample_diff(G.at(x + 1, y, z), G.at(x + 1, y + 2, z)) * (1 - t) +
		                               sample_diff(G.at(x + 1, y, z + 1), G.at(x + 1, y + 2, z + 1)) * t)
		                      : (v[5] - v[4]) * (1 - t) + (v6 - v[7]) * t;
		r[5] = v6 - va;
	} else if (E == 6) {
		const real_t va = v[7], t = va / (va - v6);
		r[0] = (real_t)(x + 1); r[1] = (real_t)y + t; r[2] = (real_t)(z + 1);
		r[3] = (x + 1 < P.nx) ? 0.5f * (sample_diff(G.at(x, y, z + 1), G.at(x + 2, y, z + 1)) * (1 - t) +
		                               sample_diff(G.at(x, y + 1, z + 1), G.at(x + 2, y + 1, z + 1)) * t)
		                      : (v[7] - v[3]) * (1 - t) + (v6 - v[2]) * t;
		r[4] = v6 - va;
		r[5] = (z + 1 < P.nz) ? 0.5f * (sample_diff(G.at(x + 1, y, z), G.at(x + 1, y, z + 2)) * (1 - t) +
		                               sample_diff(G.at(x + 1, y + 1, z), G.at(x + 1, y + 1, z + 2)) * t)
		                      : (v[7] - v[4]) * (1 - t) + (v6 - v[5]) * t;
	} else {
		const real_t va = v[2], t = va / (va - v6);
		r[0] = (real_t)x + t; r[1] = (real_t)(y + 1); r[2] = (real_t)(z + 1);
		r[3] = v6 - va;
		r[4] = (y + 1 < P.ny) ? 0.5f * (sample_diff(G.at(x, y, z + 1), G.at(x, y + 2, z + 1)) * (1 - t) +
		                               sample_diff(G.at(x + 1, y, z + 1), G.at(x + 1, y + 2, z + 1)) * t)
		                      : (v[2] - v[3]) * (1 - t) + (v6 - v[7]) * t;
		r[5] = (z + 1 < P.nz) ? 0.5f * (sample_diff(G.at(x, y + 1, z), G.at(x, y + 1, z + 2)) * (1 - t) +
		                               sample_diff(G.at(x + 1, y + 1, z), G.at(x + 1, y + 1, z + 2)) * t)
		                      : (v[2] - v[1]) * (1 - t) + (v6 - v[5]) * t;
	}
	store_vertex(P, r, c.V, c.N, id - c.v_skip);
}

// ---- fast emit, split in two passes so that each has a short dependency chain and few registers -------
// The samples the (up to three) owned vertices of a FAST record and the centre vertex of a TESTED one are made from:
// short runs along x - the 2x2 rows of the cell (x..x+2) and the four rows one step outside it (y+2 on both planes,
// plane z+2 on both rows) - 20 values.  Where an outer neighbour does not exist (far faces of the grid) the slot
// holds a sample that does, and the arithmetic below never looks at it.
//   F[row][col]: rows 0..3 = (y,z) (y+1,z) (y,z+1) (y+1,z+1); col = x, x+1, x+2 (x+1 again when x+2 is outside)
//   Y2[p][col] = (y+2 | y, z+p), Z2[q][col] = (y+q, z+2 | z); col = x, x+1
template <typename T>
struct FastSamples {
	T F[4][3];
	T Y2[2][2], Z2[2][2];
};
// ... fetched from the grid by the thread itself: 12 loads, all asked for together (the round-2 vertex pass; the host
// emulator; in the round-3 kernel the records whose rows are not staged in LDS)
template <typename T>
MC33_HD void fast_samples_direct(const GridView<T> &G, uint32_t x, uint32_t y, uint32_t z, bool xin, bool yin, bool zin, FastSamples<T> &S) {
	for (int r = 0; r < 4; r++) {
		const uint32_t yy = y + (r & 1), zz = z + (r >> 1);
		const SamplePair<T> q = G.pair(x, yy, zz);
		S.F[r][0] = q.a;
		S.F[r][1] = q.b;
		S.F[r][2] = G.at(xin ? x + 2 : x + 1, yy, zz);  // (x + 1 again on the last column: the same value as q.b, and a load without a branch -
		                                               // a conditional load made the compiler wait for each row's pair before asking for the next row)
	}
	for (int q = 0; q < 2; q++) {
		const SamplePair<T> yq = G.pair(x, yin ? y + 2 : y, z + q), zq = G.pair(x, y + q, zin ? z + 2 : z);
		S.Y2[q][0] = yq.a; S.Y2[q][1] = yq.b;
		S.Z2[q][0] = zq.a; S.Z2[q][1] = zq.b;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	{  // every sample fetched above is needed HERE: without a use outside the per-vertex branches the compiler moves the loads
		// a branch alone uses (half a pair, even) into that branch - a round trip of their own for the records that take it
		uint32_t acc = 0;
		for (int r = 0; r < 4; r++) acc += (uint32_t)S.F[r][0] + (uint32_t)S.F[r][1] + (uint32_t)S.F[r][2];
		for (int q = 0; q < 2; q++) acc += (uint32_t)S.Y2[q][0] + (uint32_t)S.Y2[q][1] + (uint32_t)S.Z2[q][0] + (uint32_t)S.Z2[q][1];
		asm volatile("" ::"v"(acc));
	}
#endif
}

// Vertices of one FAST / TESTED record from its samples: positions, normals, stores.  The arithmetic and its order are
// those of vertex_on_edge (MC:990-1000, 1029-1039, 1175-1185).  rN: rank of the vertex of edge N / of the centre among the
// vertices this cell creates (15: none); vbase: id of the cell's first vertex.
template <typename T, int MODE = -1>
MC33_HD void fast_vertices_compute(const EmitCtx<T> &c, uint32_t x, uint32_t y, uint32_t z, uint32_t vbase, uint32_t r5, uint32_t r6, uint32_t r10,
                                   uint32_t r12, const FastSamples<T> &S) {
	const Params &P = c.P;
	const bool xin = x + 1 < P.nx, yin = y + 1 < P.ny, zin = z + 1 < P.nz;  // the outer neighbours exist
	const T(&F)[4][3] = S.F;
	const T(&Y2)[2][2] = S.Y2;
	const T(&Z2)[2][2] = S.Z2;
	const real_t iso = P.iso;
	const real_t v1 = iso - (real_t)F[1][0], v2 = iso - (real_t)F[3][0], v3 = iso - (real_t)F[2][0];
	const real_t v4 = iso - (real_t)F[0][1], v5 = iso - (real_t)F[1][1], v6 = iso - (real_t)F[3][1], v7 = iso - (real_t)F[2][1];
	real_t r[6];
	if (r5 != 15u) {  // edge 5: (x+1, y+1, z) -> (x+1, y+1, z+1)
		const real_t t = v5 / (v5 - v6);
		r[0] = (real_t)(x + 1); r[1] = (real_t)(y + 1); r[2] = (real_t)z + t;
		r[3] = xin ? 0.5f * (sample_diff(F[1][0], F[1][2]) * (1 - t) + sample_diff(F[3][0], F[3][2]) * t)
		           : (v5 - v1) * (1 - t) + (v6 - v2) * t;
		r[4] = yin ? 0.5f * (sample_diff(F[0][1], Y2[0][1]) * (1 - t) + sample_diff(F[2][1], Y2[1][1]) * t)
		           : (v5 - v4) * (1 - t) + (v6 - v7) * t;
		r[5] = v6 - v5;
		store_vertex<MODE>(P, r, c.V, c.N, vbase + r5 - c.v_skip);
	}
	if (r6 != 15u) {  // edge 6: (x+1, y, z+1) -> (x+1, y+1, z+1)
		const real_t t = v7 / (v7 - v6);
		r[0] = (real_t)(x + 1); r[1] = (real_t)y + t; r[2] = (real_t)(z + 1);
		r[3] = xin ? 0.5f * (sample_diff(F[2][0], F[2][2]) * (1 - t) + sample_diff(F[3][0], F[3][2]) * t)
		           : (v7 - v3) * (1 - t) + (v6 - v2) * t;
		r[4] = v6 - v7;
		r[5] = zin ? 0.5f * (sample_diff(F[0][1], Z2[0][1]) * (1 - t) + sample_diff(F[1][1], Z2[1][1]) * t)
		           : (v7 - v4) * (1 - t) + (v6 - v5) * t;
		store_vertex<MODE>(P, r, c.V, c.N, vbase + r6 - c.v_skip);
	}
	if (r10 != 15u) {  // edge 10: (x, y+1, z+1) -> (x+1, y+1, z+1)
		const real_t t = v2 / (v2 - v6);
		r[0] = (real_t)x + t; r[1] = (real_t)(y + 1); r[2] = (real_t)(z + 1);
		r[3] = v6 - v2;
		r[4] = yin ? 0.5f * (sample_diff(F[2][0], Y2[1][0]) * (1 - t) + sample_diff(F[2][1], Y2[1][1]) * t)
		           : (v2 - v3) * (1 - t) + (v6 - v7) * t;
		r[5] = zin ? 0.5f * (sample_diff(F[1][0], Z2[1][0]) * (1 - t) + sample_diff(F[1][1], Z2[1][1]) * t)
		           : (v2 - v1) * (1 - t) + (v6 - v5) * t;
		store_vertex<MODE>(P, r, c.V, c.N, vbase + r10 - c.v_skip);
	}
	if (r12 != 15u) {  // centre vertex of a tested cell (MC:1225-1230)
		real_t vv[8] = {iso_diff(iso, (real_t)F[0][0]), iso_diff(iso, (real_t)F[1][0]), iso_diff(iso, (real_t)F[3][0]), iso_diff(iso, (real_t)F[2][0]),
		                iso_diff(iso, (real_t)F[0][1]), iso_diff(iso, (real_t)F[1][1]), iso_diff(iso, (real_t)F[3][1]), iso_diff(iso, (real_t)F[2][1])};
		vertex_centre(x, y, z, VRef{vv, 1}, r);
		store_vertex<MODE>(P, r, c.V, c.N, vbase + r12 - c.v_skip);
	}
}

// Vertices of one FAST record, one thread on its own: the samples by 12 loads of its own (fast_samples_direct), then the
// arithmetic.  (The host emulator's vertex pass; the GPU runs k_emit_vertices, which stages the rows of 64 records in LDS.)
template <typename T>
MC33_HD void emit_fast_vertices(const EmitCtx<T> &c, const Entry &en, uint32_t s) {
	const uint32_t r5 = (en.w2 >> 20) & 15u, r6 = (en.w2 >> 24) & 15u, r10 = (en.w3 >> 8) & 15u, r12 = entry_rank_centre(en);
	if ((r5 & r6 & r10 & r12) == 15u) return;  // the cell creates no vertex
	const SegCoord sc = segment_coord(c.P, s);
	const uint32_t y = sc.y, z = sc.z;
	if (z < c.z_emit) return;
	const uint32_t x = sc.xbase + (en.w0 & 0xFFu);
	const uint32_t vbase = c.seg_base[s].vbase + (en.w1 & 0xFFFFu);
	FastSamples<T> S;
	fast_samples_direct(c.G, x, y, z, x + 1 < c.P.nx, y + 1 < c.P.ny, z + 1 < c.P.nz, S);
	fast_vertices_compute(c, x, y, z, vbase, r5, r6, r10, r12, S);
}

// Triangles of one FAST (or TESTED) record.  ids: 13-slot scratch.
// The other nine edges belong to six neighbours; edge k of this cell is edge k' of
// its owner:  o0 (x-1,y,z-1): 0->6 | o1 (x-1,y,z): 1->5, 2->6 | o2 (x-1,y-1,z): 3->5 |
//             o3 (x,y,z-1): 4->6, 9->10 | o4 (x,y-1,z): 7->5, 11->10 | o5 (x,y-1,z-1): 8->10
// They sit in three other row segments - A = (y, z-1): o3 and, one record before it, o0; B = (y-1, z): o4, o2;
// C = (y-1, z-1): o5 - and o1 is the record right before this cell's own.  All directory records are fetched
// together (one 16-byte word each: the activity mask of the cell's 64-cell word and the index of its first record), then
// the (at most five) work records: two dependent round trips.  Neighbours that are not needed are redirected to this cell's own records (valid, cached
// addresses), so the loads carry no control flow.  The edges have no end point equal to the isovalue (they
// are edges of a fast cell), so their owners created regular vertices: no alias to follow.
template <typename T>
MC33_HD void fast_triangles_write(const EmitCtx<T> &c, const Entry &en, const Entry (&oe)[6], const uint32_t (&ovb)[6], const SegBase &sb, const URef &ids);
// Two records next to each other in ONE load (16 bytes; the records of x - 1 and x in a row segment are neighbours in the
// array): what costs the triangle pass is the number of load instructions of a wave - each looks up every distinct line its
// lanes touch - not their width.  (Round 3: 15 -> 12 load instructions per record, triangle pass 91 -> 88 us at 1024^3,
// 349 -> 343 us per isovalue at 2048 x 2048 x 1024.)
struct EntryA2 { EntryA lo, hi; };
MC33_HD EntryA2 entry_pair(const EntryA *p) {
	EntryA2 r;
	r.lo = p[0]; r.hi = p[1];
	return r;
}
struct SegBase2 { SegBase lo, hi; };  // ... and so the bases of the row segments y - 1 and y
MC33_HD SegBase2 seg_base_pair(const SegBase *p) {
	SegBase2 r;
	r.lo = p[0]; r.hi = p[1];
	return r;
}
// (before: the record before this one in the array, o1 when x - 1 is wanted - the caller has it from the load of `en` itself)
template <typename T>
MC33_HD void emit_fast_triangles(const EmitCtx<T> &c, const Entry &en, const EntryA &before, uint32_t s, uint32_t self_index, const URef &ids,
                                 const uint32_t *known_below = nullptr, uint32_t *keep_below = nullptr) {  // (developer experiment: the positions of x in segments A, B, C given / kept)
	// (a ghost slice of a z-slab; z = zs + s / (nseg ny), without the division)
	if (c.z_emit > c.P.zs && (uint64_t)s < (uint64_t)(c.z_emit - c.P.zs) * c.P.nseg * c.P.ny) return;
	const uint32_t xl = en.w0 & 0xFFu;
	const uint32_t i = (en.w0 >> 8) & 0xFFu;
#define MC33_SIDE(k) ((i >> (7 - (k))) & 1u)
	const bool cut0 = MC33_SIDE(0) != MC33_SIDE(1), cut1 = MC33_SIDE(1) != MC33_SIDE(2), cut2 = MC33_SIDE(3) != MC33_SIDE(2);
	const bool cut3 = MC33_SIDE(0) != MC33_SIDE(3), cut4 = MC33_SIDE(4) != MC33_SIDE(5), cut9 = MC33_SIDE(1) != MC33_SIDE(5);
	const bool cut7 = MC33_SIDE(4) != MC33_SIDE(7), cut11 = MC33_SIDE(3) != MC33_SIDE(7), cut8 = MC33_SIDE(0) != MC33_SIDE(4);
#undef MC33_SIDE
	const bool need[6] = {cut0, cut1 || cut2, cut3, cut4 || cut9, cut7 || cut11, cut8};
	// the bases of this row segment and of the one before it (y - 1: segment B) in one load
	// (asked for here, looked at only behind the neighbours' loads: a select on it up here made the directory words below
	// wait for it - a round trip of its own in every record's chain, round 4)
	const SegBase2 sp = seg_base_pair(c.seg_base + (s ? s - 1u : 0u));
	Entry oe[6];
	uint32_t ovb[6];
	if (xl != 0) {
		// segments A = (y, z-1), B = (y-1, z), C = (y-1, z-1) (index 0..2); x-1 lies in the same segments as x.  Everything is
		// written so that the loads of a round trip are asked for together: addresses by selects, never by branches (the
		// compiler waits at a branch for what it depends on), halves A of all records before any half B.
		const bool needseg[3] = {need[0] || need[3], need[2] || need[4], need[5]};
		const uint64_t dz = (uint64_t)c.P.nseg * c.P.ny;  // segment index: ((z - zs) nseg + seg) ny + y
		const uint64_t gs[3] = {(uint64_t)s - (needseg[0] ? dz : 0ull), (uint64_t)s - (needseg[1] ? 1ull : 0ull), (uint64_t)s - (needseg[2] ? dz + 1ull : 0ull)};
		DirWord sd[3] = {};
		uint32_t svb[3];
		if (!known_below)
			for (int g = 0; g < 3; g++) sd[g] = dir_word(c, gs[g], xl);  // round trip 1: one 16-byte directory word per segment ...
		// ... and the bases: B's came with this segment's; A's (y, z-1) and C's (y-1, z-1) are neighbours too
		const uint64_t sa = (uint64_t)s - ((needseg[0] || needseg[2]) ? dz : 0ull);
		const SegBase2 ap = seg_base_pair(c.seg_base + (sa ? sa - 1ull : 0ull));
		svb[0] = sa ? ap.hi.vbase : ap.lo.vbase;
		svb[1] = sp.lo.vbase;  // (looked at only when B is wanted: y >= 1 then, and s - 1 is B)
		svb[2] = ap.lo.vbase;  // (looked at only when C is wanted: y >= 1 then, and sa - 1 is C)
		EntryA oa[6];
		uint32_t oi[6];
		oi[1] = need[1] ? self_index - 1 : self_index;  // o1: the cell x-1 is active whenever needed ...
		oa[1] = need[1] ? before : entry_a(en);         // ... and its record the one before this one
		uint32_t below[3];
		for (int g = 0; g < 3; g++) below[g] = known_below ? known_below[g] : record_rank(sd[g], xl);  // the record of x in that segment (or where it would be)
		if (keep_below) for (int g = 0; g < 3; g++) keep_below[g] = below[g];
		// round trip 2: halves A of the records.  x - 1 is the record before x's position (it is active whenever needed): the two
		// of a segment come as a pair, from the record before x's on (x's own position when that is the array's first record)
		oi[3] = need[3] ? below[0] : self_index; oi[0] = need[0] ? below[0] - 1 : self_index;
		oi[4] = need[4] ? below[1] : self_index; oi[2] = need[2] ? below[1] - 1 : self_index;
		oi[5] = need[5] ? below[2] : self_index;
		const uint32_t pa = needseg[0] ? below[0] : self_index, pb = needseg[1] ? below[1] : self_index;  // (x's position in A, B; or a cached address)
		const EntryA2 qa = entry_pair(c.entries_a + (pa ? pa - 1u : 0u)), qb = entry_pair(c.entries_a + (pb ? pb - 1u : 0u));
		oa[5] = c.entries_a[oi[5]];
		oa[3] = pa ? qa.hi : qa.lo; oa[0] = qa.lo;  // (o0 is looked at only when needed: pa >= 1 then)
		oa[4] = pb ? qb.hi : qb.lo; oa[2] = qb.lo;
		// halves B: from the table for fast records; a third round trip only for neighbours that are tested or slow cells
		for (int o = 0; o < 6; o++)
			oe[o] = entry_join(oa[o], ctx_half_b(c, oa[o], oi[o]));
		ovb[0] = ovb[3] = svb[0]; ovb[2] = ovb[4] = svb[1]; ovb[5] = svb[2]; ovb[1] = s ? sp.hi.vbase : sp.lo.vbase;
	} else {
		// first cell of a row segment: the x-1 neighbours are the last cells of the previous segment (ny row segments back) -
		// six lookups, again in rounds: directory words and bases, then halves A, then halves B
		const uint32_t odx[6] = {1, 1, 1, 0, 0, 0}, ody[6] = {0, 0, 1, 0, 1, 1}, odz[6] = {1, 0, 0, 1, 0, 1};
		const uint64_t dz = (uint64_t)c.P.nseg * c.P.ny;
		uint64_t os[6];
		uint32_t oxl[6], oi[6];
		DirWord d[6];
		EntryA oa[6];
		for (int o = 0; o < 6; o++) {
			os[o] = (uint64_t)s - (need[o] ? odz[o] * dz + ody[o] + (odx[o] ? (uint64_t)c.P.ny : 0ull) : 0ull);
			oxl[o] = (need[o] && odx[o]) ? SEG_CELLS - 1u : xl;
			d[o] = dir_word(c, os[o], oxl[o]);
			ovb[o] = c.seg_base[os[o]].vbase;
		}
		for (int o = 0; o < 6; o++) {
			oi[o] = need[o] ? record_rank(d[o], oxl[o]) : self_index;
			oa[o] = c.entries_a[oi[o]];
		}
		for (int o = 0; o < 6; o++) oe[o] = entry_join(oa[o], ctx_half_b(c, oa[o], oi[o]));
	}
	const SegBase sb = s ? sp.hi : sp.lo;
	fast_triangles_write(c, en, oe, ovb, sb, ids);
}

// ... the second half: the triangles of the record from its six owner records oe[] (o0 .. o5 above; a record that is not
// needed may be anything) and the first vertex of each owner's row segment ovb[]; sb: the record's own segment.
template <typename T>
MC33_HD void fast_triangles_write(const EmitCtx<T> &c, const Entry &en, const Entry (&oe)[6], const uint32_t (&ovb)[6], const SegBase &sb, const URef &ids) {
	const uint32_t i = (en.w0 >> 8) & 0xFFu;
	uint32_t ob[6];
	for (int o = 0; o < 6; o++) ob[o] = ovb[o] + (oe[o].w1 & 0xFFFFu);
	ids[0] = ob[0] + entry_rank(oe[0], 6);
	ids[1] = ob[1] + entry_rank(oe[1], 5);  ids[2] = ob[1] + entry_rank(oe[1], 6);
	ids[3] = ob[2] + entry_rank(oe[2], 5);
	ids[4] = ob[3] + entry_rank(oe[3], 6);  ids[9] = ob[3] + entry_rank(oe[3], 10);
	ids[7] = ob[4] + entry_rank(oe[4], 5);  ids[11] = ob[4] + entry_rank(oe[4], 10);
	ids[8] = ob[5] + entry_rank(oe[5], 10);
	const uint32_t vbase = sb.vbase + (en.w1 & 0xFFFFu);
	uint32_t tpos = sb.tbase + (en.w1 >> 16) - c.t_skip;
	ids[5] = vbase + ((en.w2 >> 20) & 15u); ids[6] = vbase + ((en.w2 >> 24) & 15u); ids[10] = vbase + ((en.w3 >> 8) & 15u);
	ids[12] = vbase + entry_rank_centre(en);
	// winding (MC:683-691): n = 1 swaps the first two indices
	const uint32_t n = ((c.tab.lut[(i & 0x80) ? (i ^ 0xFF) : i] >> 11) ^ (i >> 7) ^ 1u ^ (uint32_t)c.P.normal_neg) & 1u;
	uint32_t pos = (en.w0 >> 16) & 0xFFFu, word;
	do {  // MC:780-784, 1245-1250 (all three ids differ: regular vertices on three different edges)
		word = c.tab.lut[++pos];
		const uint32_t i2 = ids[(int)(word & 15u)], i1 = ids[(int)((word >> 4) & 15u)], i0 = ids[(int)((word >> 8) & 15u)];
		uint32_t *t = c.Tri + 3 * (uint64_t)tpos++;
		t[0] = (n ? i1 : i0) + c.id_delta; t[1] = (n ? i0 : i1) + c.id_delta; t[2] = i2 + c.id_delta;
	} while (word >> 12);
}

}  // namespace mc33
