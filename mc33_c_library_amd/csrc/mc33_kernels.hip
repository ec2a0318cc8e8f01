// mc33_kernels.hip -- HIP kernels (gfx950 / CDNA4, wave64) and the device-level C ABI (include/mc33_hip.h)
// of the MI355X Marching Cubes 33 extractor.  One shared object per grid sample type, like the
// reference's one-type-per-compile model (reference include/marching_cubes_33.h:57-88):
//     default            -> float samples           (libMC33_f32.so)
//     -DMC33_GRD_F64     -> double samples AND double arithmetic / vertices (libMC33_f64.so)
//     -DMC33_GRD_U8 / _U16 / _U32 -> unsigned char / short / int samples (libMC33_u8 / _u16 / _u32.so)
//
// Passes of one extraction, in launch order ("MC:" = reference source/marching_cubes_33.c):
//   k_sweep<S,NI,ZM> - streams the volume once (MC:1832-1868) and does nothing else that costs bandwidth: sign bit per
//                   sample by wave ballot, the bit rows of a 64-row x 256-sample tile slice parked one row per LANE, so
//                   that "is any cell of this slice cut" is a handful of 64-bit logic ops.  For each slice that is cut it
//                   leaves the bit planes (each plane once; 256 bytes in compact form), a header (cut cells, halo bits)
//                   and a partial sum for k_slots.  The halo column comes from the neighbouring wave through an LDS
//                   mailbox.  S: narrow samples are loaded S to a dword; NI: isovalues classified per pass
//                   (mc33hip_sweep_many), each with its own "lane" of output buffers; ZM: how a sample is classified.
//   k_boundary    - the slice between two z-tiles of k_sweep, from the edge planes both left behind.
//   k_slots       - exclusive sums of (cut cells, batches of 64 of them) over the slice slots in storage order.
//   k_cells       - the waves take cut slices off a list k_slots made, 64 cut cells per step: sign index from the bit planes; fast cells (interior,
//                   no test needed, no sample == iso) finished from a 256-entry table, cells whose sign index needs the
//                   face / interior tests tested here (TESTED records), the rest queued for k_slow_plan; one 8-byte work
//                   record per cut cell, contiguous in the reference's visiting order, (#new vertices, #triangles) per
//                   row segment, and one descriptor per batch of 64 records for the vertex pass.
//   k_slow_plan   - the generic plan (MC:683-779, 788-1224: cells on the grid's 0-faces, corners equal to the isovalue).
//   k_slow_count  - triangles of cells with a corner equal to the isovalue, by vertex identity (MC:1235); enqueued when the
//                   context's last extraction had such cells.
//   k_seg_fix     - offsets of the row segments a slow cell changed (and the identity counts k_slow_count was not enqueued for).
//   k_scan_*      - exclusive prefix sums over the row segments in the reference's sweep order: this IS the
//                   reference's vertex / triangle numbering.
//   k_emit_vertices<MODE> - one WAVE per batch of 64 records of one slice slot: the sample rows the batch needs staged in
//                   LDS by cooperative 16-byte loads, one lane per vertex (MC:810-1230, 485-585).
//   k_emit_fast_triangles - one thread per record: vertex ids of the nine shared edges through the owners' records,
//                   triangles (MC:1235-1250); leaves the counters of the extraction in the host's pinned copy.  First of the
//                   emit passes while the record set fits the last-level cache, behind the vertex pass otherwise.
//   k_emit_slow_slots / k_emit_slow - the generic emit of the slow records: 16 lanes per record (a lane per pattern slot) in
//                   sequence behind the fast passes when they are few; a thread per record, on large grids on a second
//                   stream beside the fast passes, when they are many.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <limits>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/mc33_hip.h"
#ifdef MC33_GRD_F64
#define MC33_REAL_DOUBLE  // MC33_real is double in the double build (reference marching_cubes_33.h:80-82)
#endif
#if !defined(MC33_GRD_U8) && !defined(MC33_GRD_U16) && !defined(MC33_GRD_U32)
#define MC33_NAN_SAMPLES 1  // float / double grids may hold NaN samples
#else
#define MC33_INT_SAMPLES 1  // unsigned integer samples: no NaN, no signed zero (k_sweep's ZM)
#endif
#include "mc33_cell.h"
#include "mc33_lut_data.h"
#include "mc33_rules_data.h"

using namespace mc33;

#if defined(MC33_GRD_U16)
typedef uint16_t sample_t;
#define MC33_SAMPLE_BYTES 2
#elif defined(MC33_GRD_U8)
typedef uint8_t sample_t;
#define MC33_SAMPLE_BYTES 1
#elif defined(MC33_GRD_U32)
typedef uint32_t sample_t;
#define MC33_SAMPLE_BYTES 4
#elif defined(MC33_GRD_F64)
typedef double sample_t;
#define MC33_SAMPLE_BYTES 8
#else
typedef float sample_t;
#define MC33_SAMPLE_BYTES 4
#endif

// ---------------------------------------------------------------------------------------------------
// device-side bookkeeping
// ---------------------------------------------------------------------------------------------------
struct Counters {
	uint32_t entry_cursor;  // work records requested (may exceed the capacity)
	uint32_t slow_cursor;   // records left to k_slow_plan
	uint32_t dirty_cursor;  // row segments whose offsets k_seg_fix has to rebuild
	uint32_t batch_cursor;  // batches of <= 64 records of one slice slot (BatchDesc) the emit passes walk
	uint32_t emit_skipped;  // set by the emit kernels when they refused to run (capacity / overflow)
	uint32_t count_pending; // set by k_slow_plan when a record waits for k_slow_count
	uint32_t live_cursor;   // slice slots with cut cells listed by k_slots for k_cells (k_scan_apply, the last kernel of a tail, clears it)
	uint32_t slow_barrier;  // blocks of k_slow_all that have finished a phase (k_slots zeroes it)
	uint64_t totV, totT;    // totals over all classified slices (ghost included)
	uint64_t ghostV, ghostT;
	uint32_t debug[8];      // (-DMC33_DEV: what a guarded kernel found wrong)
};

// One record per (wave tile, cell slice) of the sweep: the sign-bit rows of the two planes of the slice,
// exactly as the wave held them (word k of sample row r in lane r).  k_sweep fills the records of slices
// that hold at least one cut cell; k_cells turns them into work records.
struct SliceHeader {
	uint32_t flags;       // bit 0: record valid (the slice holds cut cells); bit 1: a sample of the two tile
	                      // planes equals the isovalue
	uint32_t prevh_lo, prevh_hi, curh_lo, curh_hi;  // halo-column bits of the 64 sample rows (ballots)
	uint32_t cells;       // cut cells of the slice: what k_slots turns into record ranges
	uint32_t zr_lo, zr_hi;  // bit r: a sample of sample ROW r (halo column included; to the sweep's batch of rows) of one of the
	                      // two planes equals the isovalue - only cells of the rows r - 1 and r can have such a corner, not
	                      // the whole slice (an integer grid with an integer isovalue has such samples all along the
	                      // surface: the CT / MRI case)
	uint32_t zc_lo, zc_hi;  // ... and bit L: a sample that LANE L of the sweep loaded does (lane <-> columns: lane_of_column)
	uint32_t pad_[2];
};
// lane of the sweep wave that loaded column c (0..255) of a row segment; S = samples per lane and load
__host__ __device__ inline uint32_t lane_of_column(uint32_t c, uint32_t S) { return S == 1 ? (c & 63u) : S == 2 ? ((c & 127u) >> 1) : (c >> 2); }
constexpr uint32_t SLOT_CHUNK = 512;   // slice slots per partial sum (one k_slots block)
constexpr uint32_t SLICE_VALID = 1u, SLICE_HAS_ISO = 2u;
// The upper 30 bits of `flags` carry the number of the extraction that wrote the record (epoch >= 1): records
// of earlier calls are simply not valid any more, and the 2 MB of headers need no clearing between calls.
__host__ __device__ inline bool slice_valid(uint32_t flags, uint32_t epoch) { return (flags & ~SLICE_HAS_ISO) == (epoch << 2 | SLICE_VALID); }

// One block of k_sweep: the 4 row segments of group xg, the 63 cell rows of y tile yt, cell slices [z_lo, z_hi).
// The host cuts every (xg, yt) column into chunks of equal WORK (rows x planes), as many in total as the
// GPU holds blocks at once: waves of one SIMD are served oldest first, so a CU that got one block more
// than the others ends that much later, and short tiles (the last y tile) get deeper chunks.
struct SweepTile { uint32_t seg, yt, z_lo, z_hi; };  // the piece of the volume ONE WAVE of k_sweep streams: row segment, y tile, planes

// slice record of (cell slice z, y tile, row segment): groups of 4 consecutive slices of one tile column are
// adjacent (one k_cells block).  The order of the groups is the order the work records are stored in and the emit
// passes walk them in - it has nothing to do with the numbering of vertices and triangles, which comes from the scan
// over the row segments.  z group outermost (order 0).  Measured against y tile / z group / segment (1) and y tile /
// segment / z group (2), which keep the groups of a tile column - three of a slice's four sample planes are the next
// slice's too - close together in an XCD's share of the walk (round 3, profiles/r03_slot_order.txt): the vertex pass
// fetches 5 % (float 1024^3) to 10 % (ushort 2048 x 2048 x 1024) less with (1) and is 2 - 6 us faster, k_cells is
// 6 - 33 us slower (its row-segment counts and directory lines, stored [z][segment][y], are then written far apart by
// blocks that run together); (2) loses everywhere.  The L2 fetches 128-byte lines: what the vertex pass moves is
// within 1.5 x (float) / 2.2 x (ushort) of the distinct lines its stencils touch under ANY order.
#ifndef MC33_SLOT_ORDER
#define MC33_SLOT_ORDER 0
#endif
struct SlotDims { uint32_t nZG, nYT, nseg; };  // z groups (planes: one more than slices), y tiles, row segments
__host__ __device__ inline uint64_t slice_slot(uint32_t dz, uint32_t yt, uint32_t seg, const SlotDims &d) {
#if MC33_SLOT_ORDER == 0
	return ((((uint64_t)(dz >> 2) * d.nYT + yt) * d.nseg + seg) << 2) | (dz & 3u);
#elif MC33_SLOT_ORDER == 1
	return ((((uint64_t)yt * d.nZG + (dz >> 2)) * d.nseg + seg) << 2) | (dz & 3u);
#else
	return ((((uint64_t)yt * d.nseg + seg) * d.nZG + (dz >> 2)) << 2) | (dz & 3u);
#endif
}
// the inverse for a group of four slots (slot >> 2) -> (z group, y tile, row segment)
__device__ inline void slot_group_coords(uint32_t b, const SlotDims &d, uint32_t &zq, uint32_t &yt, uint32_t &seg) {
#if MC33_SLOT_ORDER == 0
	seg = b % d.nseg; const uint32_t t = b / d.nseg; yt = t % d.nYT; zq = t / d.nYT;
#elif MC33_SLOT_ORDER == 1
	seg = b % d.nseg; const uint32_t t = b / d.nseg; zq = t % d.nZG; yt = t / d.nZG;
#else
	zq = b % d.nZG; const uint32_t t = b / d.nZG; seg = t % d.nseg; yt = t / d.nseg;
#endif
}

// What one sweep leaves behind for ONE isovalue.  k_sweep can classify the samples it streams against several isovalues
// at once (NI lanes): an iso sweep over the resident grid (calculate_isosurfaces, BASELINE.json configs[4]) then reads
// the volume once per NI isovalues instead of once per isovalue.
constexpr int SWEEP_MAXNI = 4;
// The kernels of a tail (k_boundary ... k_scan_apply) work for up to SWEEP_MAXNI isovalues in ONE launch: the argument set
// of isovalue q is A.a[q], a block's isovalue is blockIdx.y (wave-uniform: the set is read through scalar loads from the
// kernel argument segment).  An iso sweep (mc33hip_sweep_many) then needs a launch of each kernel per PASS over the grid
// instead of one per isovalue; a single extraction launches with gridDim.y = 1.
template <typename T>
struct PerLane { T a[SWEEP_MAXNI]; };
struct SweepLane {
	SliceHeader *slice_hdr;  // [slice_slot]
	uint4 *slice_bits;       // [slice_slot of the PLANE][half][lane]: {word 2*half lo, hi, word 2*half+1 lo, hi} of the plane's bit
	                         // rows.  A plane is written once, by the first slice with cut cells that touches it: writes are
	                         // what the sweep pays for (100 MB of them cost as much as 600 MB of reads), and consecutive slices
	                         // share a plane
	uint32_t *slice_compact; // [slice_slot of the PLANE][64]: the plane's record in compact form, 256 bytes - an array of its own (round 4), so that
	                         // the records of the four slices of a group, which one wave writes and one wave of k_cells reads, are ONE KiB of
	                         // memory rather than four pieces 2 KiB apart
	uint8_t *plane_fmt;      // [slice_slot of the PLANE]: PLANE_COMPACT / PLANE_RAW - in which of the two arrays the plane's record is (store_plane)
	unsigned long long *slot_part;  // [slot / SLOT_CHUNK]: record batches << 32 | cells of the slices of that chunk
	uint4 *edge_bits;        // [tile * 2 + (0 bottom | 1 top)][2][lane]: bit rows of the tile's first / last plane
	uint4 *edge_hdr;         // ... their halo-column bits and "a sample equals the isovalue" flag
	uint32_t epoch;          // number of this extraction (stamped into the slice headers)
	real_t iso;
	// the isovalue as integers, for packed narrow samples classified without a conversion (k_sweep, ZM 1 / 2; sweep_iso_words):
	// wave-uniform, so they belong in SGPRs - computed on the host, they arrive there with the kernel arguments (computed in
	// the kernel they sat in 8 VGPRs of a form that has none to spare, and spilled: round 3)
	int32_t iso_gt;          // F > iso  <=>  (int)F > iso_gt: floor(iso) held to [-1, largest sample]; nothing is greater than a NaN
	uint32_t iso_eq;         // the isovalue when it is a sample value, else a word no sample equals
};

struct SweepArgs {
	GridView<sample_t> G;
	Params P;                // (P.iso is not used by the sweep: every lane has its own)
	const SweepTile *tiles;  // [wave]: the waves of a block are independent, a block is any four consecutive tiles
	uint32_t ntiles;
	SlotDims sd;             // slice_slot
	unsigned long long *trace;  // MC33_HIP_TRACE_FILE: per wave {start, end} (s_memrealtime, 100 MHz), {start, end} (s_memtime, shader clock)
	uint32_t z_end;          // end of the classified range: tiles that reach it have no tile above
	uint32_t debug;          // MC33_HIP_DEBUG, developer builds (-DMC33_DEV) only - timing experiments, results are wrong: 2 = stream
	                         // only, 16 = stream + the cut-cell test of every slice but no slice is handed on, 64 = no halo-column load
	SweepLane lane[SWEEP_MAXNI];
};
#ifdef MC33_DEV
#define MC33_DEBUG_BITS(a) ((a).debug)
#else
#define MC33_DEBUG_BITS(a) 0u  // the shipped library has no switch that changes results
#endif


// fast[i] of mc33_cell.h unpacked into the record words written for a FAST cell:
// x = w0 without the cell's x, y = w2, z = w3, w = new vertices | triangles << 8   (x == FAST_NONE: not fast)
static void fast_record_table(const uint32_t *fast, uint4 *out) {
	for (uint32_t i = 0; i < 256; i++) {
		const uint32_t f = fast[i];
		if (f == FAST_NONE) { out[i] = uint4{FAST_NONE, mc33_lut[(i & 0x80) ? (i ^ 0xFF) : i], 0, 0}; continue; }  // (y: its table word, for corner_look)
		const Entry e = make_fast_entry(0, i, f, 0, 0);
		out[i] = uint4{e.w0, e.w2, e.w3, ((f >> 16) & 15u) | ((f >> 12) & 15u) << 8};
	}
}

__device__ __forceinline__ float real_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double real_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float real_abs(float a) { return fabsf(a); }
__device__ __forceinline__ double real_abs(double a) { return fabs(a); }
__device__ __forceinline__ uint64_t u64(uint32_t lo, uint32_t hi) { return (uint64_t)hi << 32 | lo; }
__device__ __forceinline__ uint32_t row_above(uint32_t v) {  // lane r <- lane r+1 (lane 63 <- 0): DPP wave_shl:1
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t row_above(uint64_t v) { return u64(row_above((uint32_t)v), row_above((uint32_t)(v >> 32))); }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
	return u64(__builtin_amdgcn_readlane((uint32_t)v, l), __builtin_amdgcn_readlane((uint32_t)(v >> 32), l));
}

// Inclusive prefix sum / running maximum over the 64 lanes of a wave by DPP (no LDS round trips): Hillis-Steele inside
// the rows of 16 lanes (row_shr 1, 2, 4, 8: lanes without a source add 0), then lane 15 of rows 0 and 2 into rows 1 and 3
// (row_bcast:15), then lane 31 into rows 2 and 3 (row_bcast:31).
#define MC33_DPP(x, ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
	x += MC33_DPP(x, 0x111, 0xf); x += MC33_DPP(x, 0x112, 0xf); x += MC33_DPP(x, 0x114, 0xf); x += MC33_DPP(x, 0x118, 0xf);
	x += MC33_DPP(x, 0x142, 0xa); x += MC33_DPP(x, 0x143, 0xc);
	return x;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {
	x = max(x, MC33_DPP(x, 0x111, 0xf)); x = max(x, MC33_DPP(x, 0x112, 0xf)); x = max(x, MC33_DPP(x, 0x114, 0xf)); x = max(x, MC33_DPP(x, 0x118, 0xf));
	x = max(x, MC33_DPP(x, 0x142, 0xa)); x = max(x, MC33_DPP(x, 0x143, 0xc));
	return x;
}
#undef MC33_DPP

// Bit-row layouts.  A wave keeps the sign bits of a sample row of its tile in four 64-bit words (in lane r for row r).
// With one sample per lane and load (S = 1) bit j of word k is sample x = 64 k + j: the STANDARD layout, the one every
// other pass and every record in HBM uses.  Narrow samples are loaded several to a dword (S = 2 unsigned short, S = 4
// unsigned char: a wave request is 256 bytes whatever the type), and the ballot over the lanes then collects every
// S-th sample:
//     S = 2: word 2 k' + q, bit j  <->  x = 128 k' + 2 j + q            S = 4: word q, bit j  <->  x = 4 j + q
// The sweep tests for cut cells in that layout (the neighbour x + 1 of a bit is the same bit of the next word, or the
// next bit of the first word of the group) and converts the rows to the standard layout only when a slice is handed
// on - a third of the slices of a smooth field, one conversion per lane = per row.
template <int S>
__device__ __forceinline__ void succ_words(const uint64_t (&A)[4], uint64_t halo, uint64_t (&N)[4]) {  // N: the bits of the samples x + 1
	if (S == 1) {
#pragma unroll
		for (int k = 0; k < 4; k++) N[k] = (A[k] >> 1) | ((k < 3 ? (A[k < 3 ? k + 1 : 3] & 1ull) : halo) << 63);
	} else if (S == 2) {
		N[0] = A[1]; N[1] = (A[0] >> 1) | ((A[2] & 1ull) << 63);
		N[2] = A[3]; N[3] = (A[2] >> 1) | (halo << 63);
	} else {
		N[0] = A[1]; N[1] = A[2]; N[2] = A[3]; N[3] = (A[0] >> 1) | (halo << 63);
	}
}

// cells of a tile slice cut by the surface: NOT (all 8 sign bits one) and NOT (all zero)  (MC:1860).
// prev/cur: bit rows of planes z / z+1 (lane = sample row), *_h: halo-column bits; everything in layout S.
template <int S = 1>
__device__ __forceinline__ void active_cells(const uint64_t (&prev)[4], const uint64_t (&cur)[4], uint32_t prev_h, uint32_t cur_h,
                                             const uint64_t (&valid)[4], bool rowvalid, uint64_t (&act)[4]) {
	const uint32_t prev_hn = row_above(prev_h), cur_hn = row_above(cur_h);
	uint64_t A[4], O[4], As[4], Os[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint64_t pn = row_above(prev[k]), cn = row_above(cur[k]);
		A[k] = prev[k] & pn & cur[k] & cn;
		O[k] = prev[k] | pn | cur[k] | cn;
	}
	const uint64_t hA = prev_h & prev_hn & cur_h & cur_hn, hO = prev_h | prev_hn | cur_h | cur_hn;
	succ_words<S>(A, hA, As);
	succ_words<S>(O, hO, Os);
#pragma unroll
	for (int k = 0; k < 4; k++) act[k] = rowvalid ? (~((A[k] & As[k]) | ~(O[k] | Os[k])) & valid[k]) : 0ull;
}

__device__ __forceinline__ uint64_t spread2(uint32_t v) {  // bit i -> bit 2 i
	uint64_t x = v;
	x = (x | x << 16) & 0x0000FFFF0000FFFFull; x = (x | x << 8) & 0x00FF00FF00FF00FFull; x = (x | x << 4) & 0x0F0F0F0F0F0F0F0Full;
	x = (x | x << 2) & 0x3333333333333333ull; x = (x | x << 1) & 0x5555555555555555ull;
	return x;
}
__device__ __forceinline__ uint64_t spread4(uint32_t v) {  // bit i (< 16) -> bit 4 i
	uint64_t x = v & 0xFFFFu;
	x = (x | x << 24) & 0x000000FF000000FFull; x = (x | x << 12) & 0x000F000F000F000Full; x = (x | x << 6) & 0x0303030303030303ull;
	x = (x | x << 3) & 0x1111111111111111ull;
	return x;
}
__device__ __forceinline__ uint32_t gather2(uint64_t x) {  // bit 2 i -> bit i
	x &= 0x5555555555555555ull;
	x = (x | x >> 1) & 0x3333333333333333ull; x = (x | x >> 2) & 0x0F0F0F0F0F0F0F0Full; x = (x | x >> 4) & 0x00FF00FF00FF00FFull;
	x = (x | x >> 8) & 0x0000FFFF0000FFFFull; x = (x | x >> 16) & 0x00000000FFFFFFFFull;
	return (uint32_t)x;
}
__device__ __forceinline__ uint32_t gather4(uint64_t x) {  // bit 4 i -> bit i (16 bits)
	x &= 0x1111111111111111ull;
	x = (x | x >> 3) & 0x0303030303030303ull; x = (x | x >> 6) & 0x000F000F000F000Full; x = (x | x >> 12) & 0x000000FF000000FFull;
	x = (x | x >> 24) & 0xFFFFull;
	return (uint32_t)x;
}
template <int S>
__device__ __forceinline__ void to_standard(const uint64_t (&w)[4], uint64_t (&o)[4]) {
	if (S == 1) {
#pragma unroll
		for (int k = 0; k < 4; k++) o[k] = w[k];
	} else if (S == 2) {
#pragma unroll
		for (int g = 0; g < 2; g++) {
			o[2 * g] = spread2((uint32_t)w[2 * g]) | spread2((uint32_t)w[2 * g + 1]) << 1;
			o[2 * g + 1] = spread2((uint32_t)(w[2 * g] >> 32)) | spread2((uint32_t)(w[2 * g + 1] >> 32)) << 1;
		}
	} else {
#pragma unroll
		for (int q = 0; q < 4; q++)
			o[q] = spread4((uint32_t)(w[0] >> (16 * q))) | spread4((uint32_t)(w[1] >> (16 * q))) << 1 | spread4((uint32_t)(w[2] >> (16 * q))) << 2 |
			       spread4((uint32_t)(w[3] >> (16 * q))) << 3;
	}
}
template <int S>
__device__ __forceinline__ void from_standard(const uint64_t (&w)[4], uint64_t (&o)[4]) {
	if (S == 1) {
#pragma unroll
		for (int k = 0; k < 4; k++) o[k] = w[k];
	} else if (S == 2) {
#pragma unroll
		for (int g = 0; g < 2; g++) {
			o[2 * g] = (uint64_t)gather2(w[2 * g]) | (uint64_t)gather2(w[2 * g + 1]) << 32;
			o[2 * g + 1] = (uint64_t)gather2(w[2 * g] >> 1) | (uint64_t)gather2(w[2 * g + 1] >> 1) << 32;
		}
	} else {
#pragma unroll
		for (int q = 0; q < 4; q++)
			o[q] = (uint64_t)gather4(w[0] >> q) | (uint64_t)gather4(w[1] >> q) << 16 | (uint64_t)gather4(w[2] >> q) << 32 | (uint64_t)gather4(w[3] >> q) << 48;
	}
}

// cells of the segment piece [xbase + 64k, +64) that exist.  Signed arithmetic on purpose: the unsigned form
// "first >= nx ? 0 : min(64, nx - first)" is miscompiled by this toolchain (the guarded subtraction is
// hoisted with its no-wrap flag and ConstraintElimination then takes nx >= xbase + 192 for a fact).
__device__ __forceinline__ void valid_masks(uint32_t xbase, uint32_t nx, uint64_t (&valid)[4]) {
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int64_t rem = (int64_t)nx - (int64_t)(xbase + 64u * k);
		valid[k] = rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
	}
}

// A slice with cut cells is handed to k_cells: its bit rows (4 KiB), the halo-column bits, flags and counts; the
// counts also go into the partial sum of the slot's chunk (k_slots).  prev / cur: bit rows of planes z / z+1, lane =
// sample row; bp / bc: ballots of the halo-column bits.  Wave-uniform call.
// The record of a plane in slice_bits.  A bit row of a smooth field changes its value once or twice along its 256
// samples: a row with at most two changes is ONE dword - bit 0: the first sample's bit, bits 1-2: number of changes,
// bytes 1-2: their positions p (bits p and p + 1 differ) - and when all 64 rows of a plane are such rows the record is
// 256 bytes (dword r = row r) instead of 2 KiB (PLANE_COMPACT; k_cells rebuilds the words).  The hand-over is what the sweep
// pays for beyond its reads, by the byte: 68 MB at C3, 1.08 GB per 4-isovalue pass at C5 before this.
constexpr uint32_t PLANE_RAW = 0u, PLANE_COMPACT = 1u;
constexpr uint32_t PLANE_UNIFORM0 = 2u, PLANE_UNIFORM1 = 3u;  // (edge records only: every bit of the plane is 0 / 1 - nothing but the header is written)
__device__ __forceinline__ void decode_row(uint32_t desc, uint64_t (&w)[4]) {
	const uint64_t base = (desc & 1u) ? ~0ull : 0ull;
	const uint32_t n = (desc >> 1) & 3u;
#pragma unroll
	for (int k = 0; k < 4; k++) w[k] = base;
#pragma unroll
	for (int j = 0; j < 2; j++) {
		const int p = (int)((desc >> (8 + 8 * j)) & 0xFFu);
#pragma unroll
		for (int k = 0; k < 4; k++) {  // every bit after position p changes sides
			const int first = p + 1 - 64 * k;
			const uint64_t m = first <= 0 ? ~0ull : first >= 64 ? 0ull : ~0ull << first;
			w[k] ^= (uint32_t)j < n ? m : 0ull;
		}
	}
}
// (w: the plane's bit rows in layout S, lane = sample row; wave-uniform call.  The changes of a row are found in the layout
// the sweep works in - bit j of word m is sample x = 64 m + j | 128 (m >> 1) + 2 j + (m & 1) | 4 j + m for S = 1 | 2 | 4 - so a
// compact plane is never converted to the standard layout at all)
template <int S>
__device__ __forceinline__ bool encode_plane(const uint64_t (&w)[4], uint32_t &desc) {  // true (wave-uniform): every row of the plane fits its dword
	uint64_t nx[4], t[4];
	succ_words<S>(w, w[3] >> 63, nx);  // (the sample after the last one: itself - no change there)
#pragma unroll
	for (int k = 0; k < 4; k++) t[k] = w[k] ^ nx[k];
	const uint32_t n = (uint32_t)(__popcll(t[0]) + __popcll(t[1]) + __popcll(t[2]) + __popcll(t[3]));
	desc = 0;
	if (__ballot(n > 2u) != 0ull) return false;
	// at most two changes: the lowest bit of the first word that has one and the highest bit of the last such word
	const uint32_t m1 = t[0] ? 0u : t[1] ? 1u : t[2] ? 2u : 3u, m2 = t[3] ? 3u : t[2] ? 2u : t[1] ? 1u : 0u;
	const uint64_t t1 = t[0] ? t[0] : t[1] ? t[1] : t[2] ? t[2] : t[3], t2 = t[3] ? t[3] : t[2] ? t[2] : t[1] ? t[1] : t[0];
	const uint32_t b1 = t1 ? (uint32_t)__builtin_ctzll(t1) : 0u, b2 = t2 ? 63u - (uint32_t)__builtin_clzll(t2) : 0u;
	const uint32_t p1 = S == 1 ? 64u * m1 + b1 : S == 2 ? 128u * (m1 >> 1) + 2u * b1 + (m1 & 1u) : 4u * b1 + m1;
	const uint32_t p2 = S == 1 ? 64u * m2 + b2 : S == 2 ? 128u * (m2 >> 1) + 2u * b2 + (m2 & 1u) : 4u * b2 + m2;
	desc = (uint32_t)(w[0] & 1ull) | n << 1 | p1 << 8 | p2 << 16;
	return true;
}
// Stores of the sweep's hand-over go through buffer descriptors: the record's address is wave-uniform (SGPRs), the lanes
// differ by 4 or 16 bytes - one 32-bit offset register for every store of the kernel.  As plain global stores each of them
// had a 64-bit per-lane address, the loop-invariant ones (the edge records of the tile, per isovalue and form) were hoisted
// out of the plane loop, and the 4-isovalue forms - which sit at the register limit of 3 waves per SIMD - spilled them:
// 176 - 192 bytes of scratch per lane in k_sweep<2,4,*> (round 3's VERDICT; tests/test_code_objects.py now checks).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
// A loop over 0 .. N-1 whose index is a compile-time constant in the body: `#pragma unroll` is a request the compiler turns
// down when the body is large (k_sweep<4,4,*>: the plane work of four isovalues over packed uchar samples - its per-isovalue
// arrays were then indexed at run time and lived in 320 - 736 bytes of scratch memory per lane, rounds 2 - 3).
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
	if constexpr (N > 0) {
		static_for<N - 1>(f);
		f(std::integral_constant<int, N - 1>{});
	}
}
// ... only where FORCE says so; otherwise the ordinary unrolled loop (which the optimizer sees rolled first: the forms that
// fitted their registers that way keep it - the double-precision sweep over four isovalues spilled 16 registers when forced)
template <int N, bool FORCE, typename F>
__device__ __forceinline__ void unrolled_for(F &&f) {
	if constexpr (FORCE) static_for<N>(f);
	else {
#pragma unroll
		for (int i = 0; i < N; i++) f(i);
	}
}
// The lane's number, computed where it is asked for.  Everything derived from `threadIdx.x & 63` is loop-invariant, and the
// compiler keeps every such value (lane * 4, lane * 16, LDS addresses) in a register of its own across the sweep's whole loop
// for the one use per plane; the 4-isovalue form has no registers for that.  (All lanes enabled where this is called.)
__device__ __forceinline__ uint32_t fresh_lane() {
	uint32_t l;
	asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
	return l;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t record_rsrc(const void *base, uint32_t bytes) {
	return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
// writes the record at `rec` (2 KiB reserved) in the form that fits; returns the form
// (compact: where the 256-byte form goes - the head of the 2 KiB record for the edge records, slice_compact for the planes of cut slices)
template <int S>
__device__ __forceinline__ uint32_t store_plane_record(uint4 *rec, uint32_t *compact, const uint64_t (&w)[4], uint32_t lane) {
	uint32_t desc;
	if (encode_plane<S>(w, desc)) {
		__builtin_amdgcn_raw_buffer_store_b32(desc, record_rsrc(compact, 256u), lane * 4u, 0u, 0);
		return PLANE_COMPACT;
	}
	const __amdgpu_buffer_rsrc_t rs = record_rsrc(rec, 2048u);
	uint64_t o[4];
	to_standard<S>(w, o);
	__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)}, rs, lane * 16u, 0u, 0);
	__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[2], (uint32_t)(o[2] >> 32), (uint32_t)o[3], (uint32_t)(o[3] >> 32)}, rs, lane * 16u, 1024u, 0);
	return PLANE_RAW;
}
template <int S>
__device__ __forceinline__ void store_plane(const SweepLane &a, uint64_t plane_slot, const uint64_t (&w)[4], uint32_t lane) {
	const uint32_t fmt = store_plane_record<S>(a.slice_bits + plane_slot * 128u, a.slice_compact + plane_slot * 64u, w, lane);
	if (lane == 0) a.plane_fmt[plane_slot] = (uint8_t)fmt;
}

// the raw form of a plane record (2 KiB), w in layout S
template <int S>
__device__ __forceinline__ void store_plane_raw(uint4 *rec, const uint64_t (&w)[4], uint32_t lane) {
	const __amdgpu_buffer_rsrc_t rs = record_rsrc(rec, 2048u);
	uint64_t o[4];
	to_standard<S>(w, o);
	__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)}, rs, lane * 16u, 0u, 0);
	__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[2], (uint32_t)(o[2] >> 32), (uint32_t)o[3], (uint32_t)(o[3] >> 32)}, rs, lane * 16u, 1024u, 0);
}

// What a wave of the single-isovalue sweep hands on is kept in LDS and written behind the tile's LAST load (round 4).  Stores
// issued inside the read stream cost the stream far more than their bytes (15 MB of them a tenth of the kernel,
// whatever their form or place in the loop); the same stores issued when the wave has nothing left to read - measured with
// dummy data first: 0.754 -> 0.684 ms at 1024^3, against 0.636 with no stores at all.  Kept: the compact plane records (a dword
// per row: every plane of a smooth field), the slice headers with their partial sums, the first plane's edge record in compact
// form.  A plane that needs the raw form (noise) is stored at once as before; a log that is full (very deep tiles) is written
// out and started again.
constexpr uint32_t LOG_PLANES = 20, LOG_SLICES = 20, LOG_NONE = 0xFFFFFFFFu;
struct SweepLog {  // per wave
	uint32_t plane[LOG_PLANES][64];  // compact records: dword r = row r
	uint64_t plane_slot[LOG_PLANES];
	uint32_t hdr[LOG_SLICES][12];    // the ten words of a SliceHeader, word 10: batches of 64 records
	uint64_t hdr_slot[LOG_SLICES];
	uint32_t edge[64];               // the tile's first plane for k_boundary (compact)
	uint32_t edge_hdr[8];
};

// (slot: of the slice; slot_up: of the slice above = the slot of the upper plane; write_prev / write_cur: the
// plane has not been written by this wave yet)
template <int S>
__device__ __forceinline__ void hand_over_slice(const SweepLane &a, uint64_t slot, uint64_t slot_up, const uint64_t (&prev)[4],
                                                const uint64_t (&cur)[4], bool write_prev, bool write_cur, uint64_t bp, uint64_t bc,
                                                uint64_t zrows, uint64_t zcols, const uint64_t (&act)[4], uint32_t lane, uint32_t dev = 0,
                                                uint32_t *pend_chunk = nullptr, unsigned long long *pend_sum = nullptr) {
#ifdef MC33_DEV  // MC33_HIP_DEBUG 32: no bit-plane stores, no header; 128: the bit-plane stores alone (the later passes see nothing)
	if (dev & 32u) { write_prev = write_cur = false; }
#endif
	if (write_prev) store_plane<S>(a, slot, prev, lane);  // (prev, cur: layout S)
	if (write_cur) store_plane<S>(a, slot_up, cur, lane);
#ifdef MC33_DEV
	if (dev & (32u | 128u)) return;
#endif
	// cut cells and non-empty rows of the slice: the record ranges are prefix sums of these (k_slots)
	uint32_t ncell = __popcll(act[0]) + __popcll(act[1]) + __popcll(act[2]) + __popcll(act[3]);
#pragma unroll
	for (int dlt = 32; dlt; dlt >>= 1) ncell += __shfl_xor(ncell, dlt);
	const uint32_t nbatch = (ncell + 63u) >> 6;  // the records of a slice are handed to the emit passes 64 at a time (BatchDesc)
	if (lane == 0) {
		// (word by word, the padding left alone: as a struct copy the two zero words of the padding were a 64-bit zero that the
		// 4-isovalue sweep kept in a register pair across its whole loop - and spilled)
		uint32_t *h = (uint32_t *)(a.slice_hdr + slot);
		static_assert(offsetof(SliceHeader, zc_hi) == 36, "SliceHeader words");
		*(uint4 *)h = uint4{a.epoch << 2 | SLICE_VALID | (zrows ? SLICE_HAS_ISO : 0u), (uint32_t)bp, (uint32_t)(bp >> 32), (uint32_t)bc};
		*(uint4 *)(h + 4) = uint4{(uint32_t)(bc >> 32), ncell, (uint32_t)zrows, (uint32_t)(zrows >> 32)};
		*(uint2 *)(h + 8) = uint2{(uint32_t)zcols, (uint32_t)(zcols >> 32)};
		if (!pend_chunk) atomicAdd(a.slot_part + slot / SLOT_CHUNK, (unsigned long long)nbatch << 32 | ncell);
	}
	if (pend_chunk) {
		// (the passes over several isovalues: the partial sums of the wave's slices are added up per chunk of slots - consecutive
		// slices of a tile mostly fall into the same one - and reach memory as one atomic per chunk: the pass over 4 isovalues of
		// the 2048 x 2048 x 1024 grid issued 380 000 of them onto 2 100 addresses, a tenth of a millisecond with the headers)
		const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(slot / SLOT_CHUNK));
		const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncell);
		if (*pend_chunk != chunk) {
			if (*pend_sum && lane == 0) atomicAdd(a.slot_part + *pend_chunk, *pend_sum);
			*pend_chunk = chunk; *pend_sum = 0ull;
		}
		*pend_sum += (unsigned long long)((nc + 63u) >> 6) << 32 | nc;
	}
}

// ---------------------------------------------------------------------------------------------------
// k_sweep: one wave per tile (256 samples in x, 64 sample rows, a run of planes).  The waves are independent; the
// plan puts the tiles of neighbouring row segments next to each other, so that the 4 waves of a block normally read
// whole 1024-sample (4 KiB) row pieces, and fills blocks with whatever tiles come next where a grid is not a multiple
// of 1024 samples wide (plan_sweep).
//
// Every lane loads 4 samples of a row (x = xbase + 64k + lane: fully coalesced 256-byte requests),
// v = iso - F, the sign bits of the 64 lanes are collected by ballot into one 64-bit word per k, and the
// 4 words of sample row r are parked in lane r.  After a plane is in, lane r holds the bits of row r for
// planes z and z+1 and gets row r+1 from its neighbour lane: the "all 8 corners on the same side" test
// (MC:1860) of the 63 x 256 cells of the tile slice is ~100 logic ops per wave.  Slices with cut cells
// have their bit rows written out (4 KiB) for k_cells; nothing else is stored, nothing is allocated:
// the wave only streams.  (Doing the per-cell work here made the kernel end on a long tail of a few
// waves whose tiles hold most of the surface.)
//
// Tiles of one column do not overlap: a tile reads the planes z_lo+1 .. z_hi (the lowest tile of the column also
// z_lo) and handles the slices between them; the slice between its first plane and the last plane of the tile
// below is put together by k_boundary from the bit rows both tiles leave behind (2 KiB each) - re-reading that
// plane instead cost 1/depth of the traffic (6 % at depth 16).
// ---------------------------------------------------------------------------------------------------
// How a ballot (the sign bits of 64 samples of one row, an SGPR pair) is parked in the lane of its row: 0 = two v_writelane_b32
// (rounds 1 - 3), 1 = ONE v_mov_b64 with EXEC narrowed to that lane (gfx940+ moves 64 bits in one instruction, and an SGPR
// pair is a legal source): 4 instead of 8 vector instructions per row and isovalue - the passes over four isovalues of narrow
// samples are bound by exactly these (round 4)
#ifndef MC33_PARK
#define MC33_PARK 2
#endif
#ifndef MC33_SWEEP_BUFS
#define MC33_SWEEP_BUFS 2  // register buffers of loaded batches in k_sweep's single-isovalue forms (3: developer A/B, round 4)
#endif
#ifndef MC33_EDGE_UNIFORM
#define MC33_EDGE_UNIFORM 1  // (0: developer A/B - an edge record for every plane, as until round 4)
#endif
#ifndef MC33_SWEEP_DEFER
#define MC33_SWEEP_DEFER 1  // (0: developer A/B - every store of the sweep where its data is made, as until round 4)
#endif
#ifndef MC33_EDGE_LAST_COMPACT
#define MC33_EDGE_LAST_COMPACT 1
#endif
#ifndef MC33_LOG_EARLY_PLANES
#define MC33_LOG_EARLY_PLANES 0u
#endif
#ifndef MC33_EDGE_COMPACT
#define MC33_EDGE_COMPACT 0  // (developer A/B: the edge records of the single-isovalue pass in compact form too - leave_edge)
#endif
#if defined(MC33_GRD_U16)
constexpr int SWEEP_PACK = 2;  // samples per dword
#elif defined(MC33_GRD_U8)
constexpr int SWEEP_PACK = 4;
#else
constexpr int SWEEP_PACK = 1;
#endif

// S: samples per lane and load.  S = 1: every lane loads single samples (all types); S = SWEEP_PACK > 1: dwords of 2
// unsigned shorts / 4 unsigned chars - needs rows that start on a dword boundary (the host checks), and makes a batch
// 8 / 16 sample rows instead of 4, so that a wave keeps the same 16 x 256 bytes in flight.
// ZM: how a sample is classified.  0: d = iso - F, its sign bit and d == 0, exactly as the reference writes it (float and
// double samples - NaN samples, signed zeros - and the isovalue -0.0).  Integer samples (MC33_INT_SAMPLES) otherwise: the
// sign bit of iso - F is F > iso and iso - F == 0 is F == iso (both converted to MC33_real as the reference does), so the
// subtraction, the |d| and the running minimum go: 1: one compare for the sign, one for "equals the isovalue"; 2: no
// isovalue of the pass is an integer of the sample type's range - nothing can equal it, one compare per sample and isovalue
// (the 4-isovalue pass over ushort samples is bound by its instructions: 94 -> 58 per sample row).
template <int S, int NI, int ZM = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_sweep(const SweepArgs a) {  // (3 waves per SIMD: at most 168 VGPRs - the 4-lane form sits right at that edge)
	constexpr int LPR = 4 / S;    // loads per sample row
	constexpr int RB = 16 / LPR;  // sample rows per batch
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wtile = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, in an SGPR
	if (wtile >= a.ntiles) return;
	const SweepTile tile = a.tiles[wtile];
	const uint32_t yt = tile.yt, seg = tile.seg;
	const Params &P = a.P;
	// The halo column (the first sample of the next row segment, one per row) used to be a load of its own: RB lanes of a
	// batch each touching a different line for ONE sample - 8 % of the sweep's fabric reads at 1024^3 float, 15 % on the
	// ushort grid of configs[4], two thirds of them missing L2 (the main loads are non-temporal), 6 % of the float sweep's
	// time (tools/halo_cost.sh, profiles/r03_halo_cost_before.txt / r03_halo_cost_after.txt).  When the four waves of a block are the four segments of one
	// 1024-sample group over the same rows and planes (the plan makes them so wherever the grid allows), wave k gets the bit
	// from wave k + 1, which has just classified that very sample: every wave posts the column-0 bits of the plane it has
	// completed (one ballot) in an LDS mailbox, one block barrier per PLANE (the four waves run in step anyway: a plane is
	// ~270 loads), and only the last segment of the group still loads its halo.  All four waves complete the same number
	// of planes (same rows, same z range), so every wave reaches every barrier; blocks of unrelated tiles keep the load.
#ifdef MC33_NO_MAILBOX  // (developer A/B: every wave loads its halo column itself)
	bool grouped = false;
#else
	bool grouped = true;
#endif
	{
		const uint32_t t0 = blockIdx.x * 4u;
		if (t0 + 3u >= a.ntiles) grouped = false;
		else {
			const SweepTile first = a.tiles[t0];
#pragma unroll
			for (uint32_t k = 1; k < 4; k++) {
				const SweepTile o = a.tiles[t0 + k];
				grouped = grouped && o.seg == first.seg + k && o.yt == first.yt && o.z_lo == first.z_lo && o.z_hi == first.z_hi;
			}
		}
	}
	__shared__ uint64_t s_mail[2][4][NI][2];  // [plane parity][wave][isovalue]{column-0 bits of the rows, rows whose column-0 sample may equal the isovalue}
	const bool from_right = grouped && (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) < 3u;  // this wave's halo bits come from the wave to its right
	const unsigned long long t_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	const unsigned long long c_start = a.trace ? __builtin_amdgcn_s_memtime() : 0ull;
	const uint32_t xbase = seg * SEG_CELLS, y0 = yt * 63u;
	const uint32_t nrows = min(64u, P.ny + 1 - y0);  // sample rows of this tile
	const uint32_t z_lo = tile.z_lo, z_hi = tile.z_hi;
	const uint32_t pl0 = z_lo == P.zs ? z_lo : z_lo + 1u;  // first plane this tile reads
	const bool has_above = z_hi < a.z_end;

	// per-lane byte offsets of its loads inside a row (clamped into the row: bits of samples beyond the grid belong
	// to cells that the valid masks remove)
	const uint32_t rowbytes = a.G.pitch * (uint32_t)sizeof(sample_t);
	uint32_t xo[LPR];
#pragma unroll
	for (int k = 0; k < LPR; k++)
		xo[k] = S == 1 ? min(xbase + 64u * k + lane, P.nx) * (uint32_t)sizeof(sample_t)
		               : min(xbase + (256u / LPR) * k + (uint32_t)S * lane, P.nx & ~(uint32_t)(S - 1)) * (uint32_t)sizeof(sample_t);
	uint64_t valid[4];
	{
		uint64_t vstd[4];
		valid_masks(xbase, P.nx, vstd);
		from_standard<S>(vstd, valid);
	}
	const bool rowvalid = lane < 63u && y0 + lane < P.ny;
	// halo column: lane r needs the first sample of the next segment in row r; it is fetched by the batch
	// that holds row r (RB lanes per batch; the other lanes aim outside the descriptor: no memory access)
	const uint32_t xh = min(lane, nrows - 1) * rowbytes + min(xbase + SEG_CELLS, P.nx) * (uint32_t)sizeof(sample_t);

	// sign bits of the tile: word k of sample row r lives in lane r (layout S).  *_h: the halo sample's bit;
	// *_z (wave-uniform): "some sample of this plane of the tile (halo included) equals the isovalue"
	// (one set per isovalue lane; with 4 lanes the bit rows of the plane below wait in LDS - they are touched once per
	// plane, and in registers they cost the kernel a third of its waves)
	constexpr bool PREV_LDS = NI >= 4;
	__shared__ uint64_t s_prev[PREV_LDS ? NI : 1][4][PREV_LDS ? 256 : 1];
	__shared__ uint64_t s_prevz[PREV_LDS ? NI : 1][2][4];  // ... and its 'sample equals the isovalue' row / lane masks, per wave
	__shared__ uint32_t s_prevh[PREV_LDS ? NI : 1][PREV_LDS ? 256 : 1];  // ... and its halo-column bits
	constexpr bool DEFER = NI == 1 && MC33_SWEEP_DEFER;   // the hand-over goes through the wave's log in LDS (SweepLog)
	__shared__ typename std::conditional<DEFER, SweepLog, uint32_t>::type s_log[DEFER ? 4 : 1];
	uint32_t log_np = 0, log_ns = 0, log_edge = LOG_NONE;  // planes / slices in the log; the format of the pending first edge record (wave-uniform)
	uint32_t pend_chunk[NI];            // (NI >= 2) partial sums not yet added to memory: their chunk of slots ...
	unsigned long long pend_sum[NI];    // ... batches << 32 | cells (wave-uniform)
#pragma unroll
	for (int q = 0; q < NI; q++) { pend_chunk[q] = 0u; pend_sum[q] = 0ull; }
	const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	uint64_t cur[NI][4], prev[PREV_LDS ? 1 : NI][4];
	uint32_t c_lo[MC33_PARK ? 1 : NI][4], c_hi[MC33_PARK ? 1 : NI][4];  // the rows being assembled: as halves (MC33_PARK 0) ...
	uint64_t c64[MC33_PARK ? NI : 1][4];                                 // ... or as words (MC33_PARK 1)
	uint32_t cur_h[NI], prev_h[PREV_LDS ? 1 : NI];
	uint64_t cur_zc[NI], prev_zc[PREV_LDS ? 1 : NI], zcacc[NI];  // ... and the lanes that loaded one
	uint64_t cur_z[NI], prev_z[PREV_LDS ? 1 : NI], zacc[NI];  // sample rows of the plane that hold a sample equal to the isovalue (wave-uniform),
	                                           // to the batch of RB rows: one compare per batch, not per row
	constexpr bool ZMIN_REG = !(sizeof(real_t) == 8 && NI >= 4);
	real_t zmin[NI];  // min |iso - F| over the lane's samples of the batch being processed (ZM = 0, ZMIN_REG)
	uint64_t zeq[NI]; // lanes that loaded a sample equal to the isovalue in the batch being processed (ZM = 1, or ZM = 0 without ZMIN_REG; wave-uniform)
	bool cur_written[NI], prev_written[NI];  // the plane's bit rows are already in slice_bits
	real_t iso[NI];
#pragma unroll
	for (int q = 0; q < NI; q++) {
#pragma unroll
		for (int k = 0; k < 4; k++) {
			if constexpr (PREV_LDS) s_prev[q][k][threadIdx.x] = 0; else prev[q][k] = 0;
			if constexpr (MC33_PARK) c64[q][k] = 0; else c_lo[q][k] = c_hi[q][k] = 0;
		}
		cur_h[q] = 0; cur_z[q] = zacc[q] = 0; cur_zc[q] = zcacc[q] = 0;
		if constexpr (PREV_LDS) { s_prevz[q][0][wv] = 0; s_prevz[q][1][wv] = 0; s_prevh[q][threadIdx.x] = 0; } else { prev_z[q] = 0; prev_zc[q] = 0; prev_h[q] = 0; }
		cur_written[q] = prev_written[q] = false; zmin[q] = 1; zeq[q] = 0;
		iso[q] = a.lane[q].iso;
	}

	// The tile is consumed as a linear stream of batches of RB sample rows (16 coalesced 256-byte loads per
	// wave), plane after plane.  Two register buffers: the loads of batch t+1 are in flight while batch t
	// is turned into bit rows.  Loads go through a buffer descriptor per plane (scalar base + 32-bit
	// offsets, hardware range check).
	const uint32_t NB = (nrows + (uint32_t)RB - 1u) / (uint32_t)RB;
	const uint32_t T = (z_hi - pl0 + 1u) * NB;
	const uint32_t tile_bytes = nrows * rowbytes;
// Cache policy of the sweep's loads (every sample is read once).  `nt` (aux bit 1) for 4- and 8-byte samples: 0.866 - 0.875 ->
// 0.78 - 0.85 ms at 1024^3 float over four processes each way; the narrow types, whose sweep is bound by instructions rather
// than by the stream, lose with it (ushort 4-isovalue pass + 1.5 %, uchar + 10 %) and keep the default.
#ifndef MC33_SWEEP_AUX
#if defined(MC33_GRD_U16) || defined(MC33_GRD_U8)
#define MC33_SWEEP_AUX 0
#else
#define MC33_SWEEP_AUX 2
#endif
#endif
#if defined(MC33_GRD_U16)
#define MC33_LOAD(rs, vo, so) ((float)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, vo, so, MC33_SWEEP_AUX))
#elif defined(MC33_GRD_U8)
#define MC33_LOAD(rs, vo, so) ((float)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(rs, vo, so, MC33_SWEEP_AUX))
#elif defined(MC33_GRD_U32)
#define MC33_LOAD(rs, vo, so) ((float)(uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, MC33_SWEEP_AUX))
#elif defined(MC33_GRD_F64)
#define MC33_LOAD(rs, vo, so) (__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, vo, so, MC33_SWEEP_AUX)))
#else
#define MC33_LOAD(rs, vo, so) (__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, MC33_SWEEP_AUX)))
#endif
	typedef typename std::conditional<S == 1, real_t, uint32_t>::type raw_t;  // what a load leaves in a register
	// every batch is exactly 17 loads, whatever the position in the tile (the wait counts the compiler
	// derives are then exact and the prefetched batch really stays in flight)
	auto issue = [&](raw_t (&d)[16], real_t &hv, uint32_t p, uint32_t bi) __attribute__((always_inline)) {
		const sample_t *base = a.G.p + (uint64_t)(p - a.G.z0) * a.G.slice + (uint64_t)y0 * a.G.pitch;
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, tile_bytes, 0x00020000);
		const uint32_t r = bi * (uint32_t)RB;
#pragma unroll
		for (int rr = 0; rr < RB; rr++) {
			const uint32_t so = min(r + rr, nrows - 1) * rowbytes;
#pragma unroll
			for (int k = 0; k < LPR; k++) {
				if constexpr (S == 1) d[rr * LPR + k] = MC33_LOAD(rs, xo[k], so);
				else d[rr * LPR + k] = __builtin_amdgcn_raw_buffer_load_b32(rs, xo[k], so, MC33_SWEEP_AUX);
			}
		}
		// (MC33_HIP_DEBUG 64, developer builds: no halo sample is ever fetched - what the one-sample-per-row load costs the stream)
		hv = MC33_LOAD(rs, ((lane / (uint32_t)RB) == bi && !from_right && !(MC33_DEBUG_BITS(a) & 64u)) ? xh : 0xFFFFFFF0u, 0u);
	};
	// the four samples of row rr of a batch in word order of layout S
	auto sample = [&](const raw_t (&dd)[16], int rr, int k) -> real_t {
		if constexpr (S == 1) return dd[rr * 4 + k];
		else if constexpr (S == 2) return (real_t)((dd[rr * 2 + (k >> 1)] >> (16 * (k & 1))) & 0xFFFFu);
		else return (real_t)((dd[rr] >> (8 * k)) & 0xFFu);
	};

	// Packed narrow samples against an isovalue without converting them (ZM 1, 2): F > iso is F > floor(iso) in integers -
	// one compare on the halfword / byte where it sits in the loaded dword instead of a conversion and a compare (the
	// 4-isovalue pass over ushort samples is bound by its instructions).
	// (the integer words come with the kernel arguments: SweepLane::iso_gt / iso_eq)
	auto raw_sample = [&](const raw_t (&dd)[16], int rr, int k) -> uint32_t {  // (S >= 2) the sample as it was loaded
		if constexpr (S == 2) return ((uint32_t)dd[rr * 2 + (k >> 1)] >> (16 * (k & 1))) & 0xFFFFu;
		else if constexpr (S == 4) return ((uint32_t)dd[rr] >> (8 * k)) & 0xFFu;
		else return 0u;
	};
	// ---- the wave's log of what it hands on (DEFER; see SweepLog) ----
	auto log_flush = [&]() __attribute__((always_inline)) {
#ifdef MC33_LOG_NO_FLUSH  // (developer timing experiment: the log is kept and dropped - results wrong)
		log_np = 0; log_ns = 0; log_edge = LOG_NONE;
		return;
#endif
		if constexpr (DEFER) {
			SweepLog &G = s_log[wv];
			const SweepLane &L0 = a.lane[0];
			const uint32_t ln = fresh_lane();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			for (uint32_t k = 0; k < log_np; k++) {  // wave-uniform
				const uint64_t slot = readlane64(G.plane_slot[k], 0);
				__builtin_amdgcn_raw_buffer_store_b32(G.plane[k][ln], record_rsrc(L0.slice_compact + slot * 64u, 256u), ln * 4u, 0u, 0);
				if (ln == 0) L0.plane_fmt[slot] = (uint8_t)PLANE_COMPACT;
			}
			{  // a lane per slice: its header; the partial sums (k_slots) added up per chunk of slots first - the slices of a tile come in
				// rising slot order, so the slices of one chunk are neighbouring lanes, and the first of them adds for all (a tile's 6
				// cut slices lie in 1 - 2 chunks: a third of the atomics, all of which arrive in the kernel's last microseconds)
				const bool mine = ln < log_ns;
				const uint32_t e = mine ? ln : 0u;
				const uint32_t *w = G.hdr[e];
				const uint64_t slot = G.hdr_slot[e];
				if (mine) {
					uint32_t *h = (uint32_t *)(L0.slice_hdr + slot);
					*(uint4 *)h = uint4{w[0], w[1], w[2], w[3]};
					*(uint4 *)(h + 4) = uint4{w[4], w[5], w[6], w[7]};
					*(uint2 *)(h + 8) = uint2{w[8], w[9]};
				}
				const uint32_t chunk = mine ? (uint32_t)(slot / SLOT_CHUNK) : 0xFFFFFFFFu;
				const uint32_t cells = mine ? w[5] : 0u, batches = mine ? w[10] : 0u;
				uint32_t sum_c = cells, sum_b = batches;
#pragma unroll 1  // (rolled: unrolled, its 57 cross-lane reads were all asked for at once and cost the kernel a wave per SIMD)
				for (uint32_t dlt = 1; dlt < log_ns; dlt++) {
					const uint32_t c2 = __shfl_down(chunk, dlt), v2 = __shfl_down(cells, dlt), b2 = __shfl_down(batches, dlt);
					const bool same = c2 == chunk && ln + dlt < 64u;
					sum_c += same ? v2 : 0u; sum_b += same ? b2 : 0u;
				}
				const uint32_t before = __shfl_up(chunk, 1);
				if (mine && (ln == 0u || before != chunk)) atomicAdd(L0.slot_part + chunk, (unsigned long long)sum_b << 32 | sum_c);
			}
			if (log_edge != LOG_NONE) {  // the first plane's edge record: compact, or wholly on one side (header only)
				if (log_edge == PLANE_COMPACT) __builtin_amdgcn_raw_buffer_store_b32(G.edge[ln], record_rsrc(L0.edge_bits + (uint64_t)wtile * 2u * 128u, 2048u), ln * 4u, 0u, 0);
				if (ln == 0) {
					const uint32_t *e = G.edge_hdr;
					L0.edge_hdr[(uint64_t)wtile * 2u * 2u] = uint4{e[0], e[1], e[2], e[3]};
					L0.edge_hdr[(uint64_t)wtile * 2u * 2u + 1u] = uint4{e[4], e[5], e[6], e[7]};
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the log is written again)
			log_np = 0; log_ns = 0; log_edge = LOG_NONE;
		}
	};
	// a plane of a cut slice: into the log when it has the compact form, to memory at once when it needs the raw one
	auto log_plane = [&](uint64_t slot, const uint64_t (&w)[4], uint32_t ln) __attribute__((always_inline)) {
		if constexpr (DEFER) {
			uint32_t desc;
			if (encode_plane<S>(w, desc)) {  // (wave-uniform)
				if (log_np == LOG_PLANES) log_flush();
				SweepLog &G = s_log[wv];
				G.plane[log_np][ln] = desc;
				if (ln == 0) G.plane_slot[log_np] = slot;
				log_np++;
			} else {
				store_plane_raw<S>(a.lane[0].slice_bits + slot * 128u, w, ln);
				if (ln == 0) a.lane[0].plane_fmt[slot] = (uint8_t)PLANE_RAW;
			}
		}
	};
	auto log_header = [&](uint64_t slot, uint64_t bp, uint64_t bc, uint64_t zrows, uint64_t zcols, const uint64_t (&act)[4], uint32_t ln) __attribute__((always_inline)) {
		if constexpr (DEFER) {
			uint32_t ncell = __popcll(act[0]) + __popcll(act[1]) + __popcll(act[2]) + __popcll(act[3]);
#pragma unroll
			for (int dlt = 32; dlt; dlt >>= 1) ncell += __shfl_xor(ncell, dlt);
			if (log_ns == LOG_SLICES) log_flush();
			if (ln == 0) {
				SweepLog &G = s_log[wv];
				uint32_t *w = G.hdr[log_ns];
				w[0] = a.lane[0].epoch << 2 | SLICE_VALID | (zrows ? SLICE_HAS_ISO : 0u);
				w[1] = (uint32_t)bp; w[2] = (uint32_t)(bp >> 32); w[3] = (uint32_t)bc; w[4] = (uint32_t)(bc >> 32); w[5] = ncell;
				w[6] = (uint32_t)zrows; w[7] = (uint32_t)(zrows >> 32); w[8] = (uint32_t)zcols; w[9] = (uint32_t)(zcols >> 32);
				w[10] = (ncell + 63u) >> 6;  // the records of a slice are handed to the emit passes 64 at a time (BatchDesc)
				G.hdr_slot[log_ns] = slot;
			}
			log_ns++;
		}
	};
	real_t halo = 0;  // lane r: halo sample of row r of the plane being assembled
	// (always_inline: the body is called twice, and in the largest forms - packed uchar samples, four isovalues, equality tests - the
	// compiler made a real FUNCTION of it, every captured array behind a pointer into 1.5 KiB of scratch memory per lane)
	auto process = [&](const raw_t (&dd)[16], const real_t &hv, uint32_t p, uint32_t bi) __attribute__((always_inline)) {
		// (developer A/B, MC33_LOG_EARLY_PLANES = n: the log goes out n planes before the tile's end instead of behind it.  Every wave of
		// the launch ends at the same moment, and what they all store then is the kernel's tail - 0.03 of 0.68 ms at 1024^3 - but
		// stores beside even the last planes' loads cost more: 0.691 / 0.703 / 0.694 -> 0.711 / 0.708 / 0.704 (n = 1) -> 0.716 / 0.724 / 0.725 (2))
		if (DEFER && MC33_LOG_EARLY_PLANES && bi == 0u && p + MC33_LOG_EARLY_PLANES == z_hi + 1u && z_hi - pl0 >= 2u * MC33_LOG_EARLY_PLANES) log_flush();
		const uint32_t r = bi * (uint32_t)RB;
		halo = (lane / (uint32_t)RB) == bi ? hv : halo;
		unrolled_for<RB, (S >= 4)>([&](auto rc) __attribute__((always_inline)) {
			const int rr = rc;
			real_t f[4];
#pragma unroll
			for (int k = 0; k < 4; k++) f[k] = sample(dd, rr, k);
			uint64_t bwq[NI][4];  // (MC33_PARK 2: the ballots of the row for all isovalues, parked two isovalues per EXEC switch)
			static_for<NI>([&](auto qc) __attribute__((always_inline)) {
				constexpr int q = decltype(qc)::value;
				uint64_t bw[4];
#pragma unroll
				for (int k = 0; k < 4; k++) {
					uint64_t bb;
					if constexpr (ZM == 0) {
						const real_t d = iso[q] - f[k];                       // MC:1852-1855
						bb = __ballot(sign_of(d) != 0);                       // MC:1856-1859 (sign bit)
#ifdef MC33_NAN_SAMPLES
						bb ^= __ballot(d != d);  // NaN sample: the sign the reference sees is the NaN's own (see iso_diff)
#endif
						// "equals the isovalue": a running minimum of |d| per lane, looked at once per batch (one instruction per sample) -
						// except in the double-precision pass over four isovalues, which has no four register pairs for it: a compare
						// per sample there, gathered in SGPRs
						if constexpr (ZMIN_REG) zmin[q] = real_min(zmin[q], real_abs(d));
						else zeq[q] |= __ballot(d == 0);
					} else if constexpr (S >= 2) {
						const uint32_t ri = raw_sample(dd, rr, k);
						bb = __ballot((int32_t)ri > a.lane[q].iso_gt);
						if constexpr (ZM == 1) zeq[q] |= __ballot(ri == a.lane[q].iso_eq);
					} else {
						bb = __ballot(f[k] > iso[q]);                         // = the sign bit of iso - F for an integer sample
						if constexpr (ZM == 1) zeq[q] |= __ballot(f[k] == iso[q]);
					}
					bw[k] = bb;
				}
				// park the bit row of sample row r+rr in lane r+rr
				// (references and the row number named here: operands of an asm statement do not capture by themselves inside a generic lambda;
				// the row number through readfirstlane - uniform anyway, but short of SGPRs the compiler moved the batch counter into
				// a vector register and handed THAT to the "s" operand)
				const uint32_t rowsel = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r + (uint32_t)rr));
				if constexpr (MC33_PARK == 2 && NI >= 2) {
#pragma unroll
					for (int k = 0; k < 4; k++) bwq[q][k] = bw[k];
				} else if constexpr (MC33_PARK) {
					// EXEC = that one lane, four 64-bit moves from the SGPR pairs, EXEC back (it is all ones here: the wave's control flow is
					// uniform; saved and restored all the same).  SALU writes of EXEC need no wait states before a VALU instruction.
					uint64_t &w0 = c64[q][0], &w1 = c64[q][1], &w2 = c64[q][2], &w3 = c64[q][3];
					uint64_t saved;
					asm volatile(
					    "s_mov_b64 %4, exec\n\t"
					    "s_lshl_b64 exec, 1, %9\n\t"
					    "v_mov_b64 %0, %5\n\tv_mov_b64 %1, %6\n\tv_mov_b64 %2, %7\n\tv_mov_b64 %3, %8\n\t"
					    "s_mov_b64 exec, %4"
					    : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "=&s"(saved)
					    : "s"(bw[0]), "s"(bw[1]), "s"(bw[2]), "s"(bw[3]), "s"(rowsel)
					    : "scc");  // (s_lshl_b64 sets SCC: without the clobber the compiler carried a loop condition across the statement in it)
				} else {
					// v_writelane takes its lane select from M0 when the data operand is an SGPR too (one SGPR per VOP3 on gfx9-class encodings)
					uint32_t &l0 = c_lo[q][0], &h0 = c_hi[q][0], &l1 = c_lo[q][1], &h1 = c_hi[q][1], &l2 = c_lo[q][2], &h2 = c_hi[q][2], &l3 = c_lo[q][3], &h3 = c_hi[q][3];
					const uint32_t m0 = (uint32_t)bw[0], m1 = (uint32_t)(bw[0] >> 32), m2 = (uint32_t)bw[1], m3 = (uint32_t)(bw[1] >> 32);
					const uint32_t m4 = (uint32_t)bw[2], m5 = (uint32_t)(bw[2] >> 32), m6 = (uint32_t)bw[3], m7 = (uint32_t)(bw[3] >> 32);
					asm volatile(
					    "s_mov_b32 m0, %16\n\t"
					    "v_writelane_b32 %0, %8, m0\n\tv_writelane_b32 %1, %9, m0\n\t"
					    "v_writelane_b32 %2, %10, m0\n\tv_writelane_b32 %3, %11, m0\n\t"
					    "v_writelane_b32 %4, %12, m0\n\tv_writelane_b32 %5, %13, m0\n\t"
					    "v_writelane_b32 %6, %14, m0\n\tv_writelane_b32 %7, %15, m0"
					    : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2), "+v"(l3), "+v"(h3)
					    : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7), "s"(rowsel)
					    : "m0");
				}
			});
			if constexpr (MC33_PARK == 2 && NI >= 2) {
				static_for<NI / 2>([&](auto hc) __attribute__((always_inline)) {
					constexpr int q0 = 2 * decltype(hc)::value;
					const uint32_t rowsel = (uint32_t)__builtin_amdgcn_readfirstlane((int)(r + (uint32_t)rr));
					uint64_t &w0 = c64[q0][0], &w1 = c64[q0][1], &w2 = c64[q0][2], &w3 = c64[q0][3];
					uint64_t &w4 = c64[q0 + 1][0], &w5 = c64[q0 + 1][1], &w6 = c64[q0 + 1][2], &w7 = c64[q0 + 1][3];
					const uint64_t b0 = bwq[q0][0], b1 = bwq[q0][1], b2 = bwq[q0][2], b3 = bwq[q0][3];
					const uint64_t b4 = bwq[q0 + 1][0], b5 = bwq[q0 + 1][1], b6 = bwq[q0 + 1][2], b7 = bwq[q0 + 1][3];
					uint64_t saved;
					asm volatile(
					    "s_mov_b64 %8, exec\n\t"
					    "s_lshl_b64 exec, 1, %17\n\t"
					    "v_mov_b64 %0, %9\n\tv_mov_b64 %1, %10\n\tv_mov_b64 %2, %11\n\tv_mov_b64 %3, %12\n\t"
					    "v_mov_b64 %4, %13\n\tv_mov_b64 %5, %14\n\tv_mov_b64 %6, %15\n\tv_mov_b64 %7, %16\n\t"
					    "s_mov_b64 exec, %8"
					    : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7), "=&s"(saved)
					    : "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5), "s"(b6), "s"(b7), "s"(rowsel)
					    : "scc");
				});
			}
		});
		if constexpr (ZM != 2)
			static_for<NI>([&](auto qc) __attribute__((always_inline)) {  // a sample of these RB rows equals the isovalue: mark the rows
				constexpr int q = decltype(qc)::value;
				const uint64_t zb = (ZM == 0 && ZMIN_REG) ? __ballot(zmin[q] == 0) : zeq[q];
				if (zb) { zacc[q] |= ((1ull << RB) - 1ull) << r; zcacc[q] |= zb; }
				zmin[q] = 1; zeq[q] = 0;
			});
		if (bi != NB - 1) return;
		// ---- the plane is complete ----
		// (the forms over several isovalues sit at their register limit: the lane's number is computed afresh here - fresh_lane)
		const uint32_t lp = NI >= 2 ? fresh_lane() : lane;
		const uint32_t tid = NI >= 2 ? wv * 64u + lp : threadIdx.x;
		const uint32_t par = (p - pl0) & 1u;
		if (grouped) {  // (block-uniform) column 0 of this plane for the wave to the left; the right neighbour's for this wave
			static_for<NI>([&](auto qc) __attribute__((always_inline)) {
				constexpr int q = decltype(qc)::value;
				const uint64_t hb = __ballot(((MC33_PARK ? (uint32_t)c64[MC33_PARK ? q : 0][0] : c_lo[MC33_PARK ? 0 : q][0]) & 1u) != 0u);  // (word 0 bit 0 is the segment's first sample in every layout S)
				if (lp == 0) {
					s_mail[par][wv][q][0] = hb;
					s_mail[par][wv][q][1] = (ZM != 2 && (zcacc[q] & 1ull)) ? zacc[q] : 0ull;  // (lane 0 loaded column 0; rows to the batch: a superset is fine)
				}
			});
			// (not __syncthreads(): that also waits for the prefetched batch's loads - only the mailbox's LDS writes must have landed)
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		}
		static_for<NI>([&](auto qc) __attribute__((always_inline)) {
			constexpr int q = decltype(qc)::value;
			const SweepLane &L = a.lane[q];
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if constexpr (MC33_PARK) { cur[q][k] = c64[q][k]; c64[q][k] = 0; }
				else { cur[q][k] = u64(c_lo[q][k], c_hi[q][k]); c_lo[q][k] = c_hi[q][k] = 0; }
			}
			if (from_right) {  // (wave-uniform; the mailbox word is read here, per isovalue: held over the loop it cost the 4-isovalue form registers it does not have)
				const uint64_t nb_bits = s_mail[par][wv + 1u][q][0], nb_zero = ZM != 2 ? s_mail[par][wv + 1u][q][1] : 0ull;
				cur_h[q] = (uint32_t)((nb_bits >> lp) & 1ull);
				const uint64_t zh = nb_zero & (nrows >= 64u ? ~0ull : ((1ull << nrows) - 1ull));
				cur_z[q] = zacc[q] | zh;
				cur_zc[q] = zh ? ~0ull : zcacc[q];
				zacc[q] = zcacc[q] = 0;
			} else {
				const real_t dh = iso[q] - halo;
				cur_h[q] = sign_of(dh);
#ifdef MC33_NAN_SAMPLES
				cur_h[q] ^= (uint32_t)(dh != dh);
#endif
				const uint64_t zh = __ballot(lp < nrows && dh == 0);  // (lanes past the tile never loaded a halo sample)
				cur_z[q] = zacc[q] | zh;
				cur_zc[q] = zh ? ~0ull : zcacc[q];  // (a halo sample: any column)
				zacc[q] = zcacc[q] = 0;
			}
			auto leave_edge = [&](uint32_t which) __attribute__((always_inline)) {  // bit rows of this plane for k_boundary (a plane record like those of slice_bits)
				// (in compact form where it fits only in the passes over several isovalues, which are bound by what they write:
				// 2.26 -> 2.19 ms per 4-isovalue pass at C5; the single-isovalue pass lost with it - 0.789 -> 0.818 ms at C3,
				// eight processes each way, and again in round 3: 0.73 -> 0.77 - 0.81 - and keeps the raw form.  These 17 MB (1024^3, two records per tile) cost the float
				// sweep 0.065 of its 0.73 ms - the first plane's 0.045, the last one's 0.02 - and three times what the 25 MB of the
				// slices handed on cost; it is the two 1 KiB stores, not the header; holding the first plane's record back in registers
				// for 1 - 8 planes by tile number, or to the tile's end, or storing it nontemporal, changes nothing - and with the
				// slices handed on, as in every real extraction, the sweep WITHOUT edge records is no faster at all: 0.72 - 0.75 ->
				// 0.75 - 0.78 ms; the costs of the sweep's stores do not add (round 3, profiles/r03_sweep_parts.txt))
				uint32_t fmt = PLANE_RAW;
				uint4 *rec = L.edge_bits + ((uint64_t)wtile * 2u + which) * 128u;
				// A plane of the tile that lies wholly on one side of the surface - two thirds of them on a smooth field - leaves no
				// record, only the header with its side (round 4: the records are two thirds of what the single-isovalue sweep writes)
				const uint64_t w_or = cur[q][0] | cur[q][1] | cur[q][2] | cur[q][3], w_and = cur[q][0] & cur[q][1] & cur[q][2] & cur[q][3];
				const bool all0 = MC33_EDGE_UNIFORM && __ballot(w_or != 0ull) == 0ull, all1 = MC33_EDGE_UNIFORM && __ballot(w_and != ~0ull) == 0ull;
				bool deferred = false;  // (the tile's FIRST plane: its record, when it is small, waits in the log with everything else)
				if (all0 || all1) { fmt = all0 ? PLANE_UNIFORM0 : PLANE_UNIFORM1; deferred = DEFER && which == 0u && !MC33_DEBUG_BITS(a); }
				else if (DEFER && which == 0u && !MC33_DEBUG_BITS(a)) {
					uint32_t desc;
					if (encode_plane<S>(cur[q], desc)) {
						if constexpr (DEFER) s_log[wv].edge[lp] = desc;
						fmt = PLANE_COMPACT; deferred = true;
					} else store_plane_raw<S>(rec, cur[q], lp);
				}
				else if constexpr (NI >= 2 || MC33_EDGE_COMPACT || (DEFER && MC33_EDGE_LAST_COMPACT)) fmt = store_plane_record<S>(rec, (uint32_t *)rec, cur[q], lp);  // (DEFER: the last plane's record, stored at the tile's end with the rest: the compact form where it fits)
				else if (!(MC33_DEBUG_BITS(a) & 8192u)) {  // (developer builds: 8192 no record, 4096 no header)
					uint64_t o[4];
					to_standard<S>(cur[q], o);
					const __amdgpu_buffer_rsrc_t rs = record_rsrc(rec, 2048u);
					__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[0], (uint32_t)(o[0] >> 32), (uint32_t)o[1], (uint32_t)(o[1] >> 32)}, rs, lp * 16u, 0u, 0);
					__builtin_amdgcn_raw_buffer_store_b128(u32x4_t{(uint32_t)o[2], (uint32_t)(o[2] >> 32), (uint32_t)o[3], (uint32_t)(o[3] >> 32)}, rs, lp * 16u, 1024u, 0);
				}
				const uint64_t bh = __ballot(cur_h[q] != 0);
				if (deferred) {
					if constexpr (DEFER) {
						if (lp == 0) {
							uint32_t *e = s_log[wv].edge_hdr;
							e[0] = (uint32_t)bh; e[1] = (uint32_t)(bh >> 32); e[2] = (uint32_t)cur_z[q]; e[3] = (uint32_t)(cur_z[q] >> 32);
							e[4] = (uint32_t)cur_zc[q]; e[5] = (uint32_t)(cur_zc[q] >> 32); e[6] = fmt; e[7] = 0u;
						}
						log_edge = fmt;
					}
				} else
				if (lp == 0 && !(MC33_DEBUG_BITS(a) & 4096u)) {
					L.edge_hdr[((uint64_t)wtile * 2u + which) * 2u] = uint4{(uint32_t)bh, (uint32_t)(bh >> 32), (uint32_t)cur_z[q], (uint32_t)(cur_z[q] >> 32)};
					L.edge_hdr[((uint64_t)wtile * 2u + which) * 2u + 1u] = uint4{(uint32_t)cur_zc[q], (uint32_t)(cur_zc[q] >> 32), fmt, 0u};
				}
			};
			if (MC33_DEBUG_BITS(a) & 2u) {
			} else {
				// (developer builds: 256 no edge records, 1024 / 2048 none for the first / last plane, 512 no cut-cell test)
				if (p == pl0 && pl0 != z_lo && !(MC33_DEBUG_BITS(a) & (256u | 1024u))) leave_edge(0);
				if (p > pl0 && !(MC33_DEBUG_BITS(a) & 512u)) {
					uint64_t act[4], pq[4];
#pragma unroll
					for (int k = 0; k < 4; k++) {
						if constexpr (PREV_LDS) pq[k] = s_prev[q][k][tid]; else pq[k] = prev[q][k];
					}
					uint32_t ph;
					if constexpr (PREV_LDS) ph = s_prevh[q][tid]; else ph = prev_h[q];
					active_cells<S>(pq, cur[q], ph, cur_h[q], valid, rowvalid, act);
					if (__ballot((act[0] | act[1] | act[2] | act[3]) != 0ull) && !(MC33_DEBUG_BITS(a) & 16u)) {  // wave-uniform: hand the slice to k_cells
						uint64_t pz, pzc;
						if constexpr (PREV_LDS) { pz = readlane64(s_prevz[q][0][wv], 0); pzc = readlane64(s_prevz[q][1][wv], 0); }  // (wave-uniform: into SGPRs, not four registers held from an early LDS read to the header's store)
						else { pz = prev_z[q]; pzc = prev_zc[q]; }
						if (DEFER && !MC33_DEBUG_BITS(a)) {  // (developer switches keep the direct stores they were written for)
							const uint64_t slot = slice_slot(p - 1 - P.zs, yt, seg, a.sd), slot_up = slice_slot(p - P.zs, yt, seg, a.sd);
							if (!prev_written[q]) log_plane(slot, pq, lp);
							log_plane(slot_up, cur[q], lp);
							log_header(slot, __ballot(ph != 0), __ballot(cur_h[q] != 0), pz | cur_z[q], pzc | cur_zc[q], act, lp);
						} else
						hand_over_slice<S>(L, slice_slot(p - 1 - P.zs, yt, seg, a.sd), slice_slot(p - P.zs, yt, seg, a.sd), pq, cur[q],
						                   !prev_written[q], true, __ballot(ph != 0), __ballot(cur_h[q] != 0), pz | cur_z[q], pzc | cur_zc[q], act, lp,
						                   MC33_DEBUG_BITS(a), NI >= 2 ? &pend_chunk[q] : nullptr, NI >= 2 ? &pend_sum[q] : nullptr);
						cur_written[q] = true;
					}
				}
				if (p == z_hi && has_above && !(MC33_DEBUG_BITS(a) & (256u | 2048u))) leave_edge(1);
			}
#pragma unroll
			for (int k = 0; k < 4; k++) {
				if constexpr (PREV_LDS) s_prev[q][k][tid] = cur[q][k]; else prev[q][k] = cur[q][k];
			}
			if constexpr (PREV_LDS) s_prevh[q][tid] = cur_h[q]; else prev_h[q] = cur_h[q];
			if constexpr (PREV_LDS) { s_prevz[q][0][wv] = cur_z[q]; s_prevz[q][1][wv] = cur_zc[q]; } else { prev_z[q] = cur_z[q]; prev_zc[q] = cur_zc[q]; }
			prev_written[q] = cur_written[q];
			cur_written[q] = false;
			// (one isovalue's plane work at a time: interleaved by the scheduler, the four of them need more registers than 3 waves per SIMD leave)
			if constexpr (NI >= 2) __builtin_amdgcn_sched_barrier(0);
		});
	};

	raw_t dA[16], dB[16];
	real_t hA = 0, hB = 0;
	uint32_t ip = pl0, ib = 0, pp = pl0, pb = 0;  // (plane, batch) of the next issue / of the next process
	// past the end of the tile the prefetch simply re-reads the last batch (it is never processed)
#define MC33_ADV(p_, b_) do { if (++(b_) == NB) { (b_) = 0; ++(p_); } } while (0)
#define MC33_ADV_ISSUE() do { if (ip != z_hi || ib + 1 != NB) MC33_ADV(ip, ib); } while (0)
	// (Tried in round 3: the loads that refill a buffer issued as soon as its rows are bit rows, BEFORE the work on a complete
	// plane, so that two batches stay in flight during that work and a hand-over's stores are younger than the refill.  The
	// refill's registers are then live across the plane's work: 92 -> 134 VGPRs for one isovalue per pass = 3 waves per SIMD
	// instead of 4, float 1024^3 0.73 -> 0.82 ms; held to 128 (36 bytes of scratch) 0.73 - 0.75 -> 0.74 - 0.75, ushort
	// 1.66 -> 1.72 ms.  No gain.  What the plane's work costs the stream is its stores, wherever they are issued:
	// profiles/r03_sweep_parts.txt.)
	issue(dA, hA, ip, ib); MC33_ADV_ISSUE();
#ifdef MC33_DEV  // MC33_HIP_DEBUG 16384: the loads alone - every batch waited for and dropped, no classification at all
	if (MC33_DEBUG_BITS(a) & 16384u) {
		auto drop = [&](const raw_t (&dd)[16], const real_t &hv) __attribute__((always_inline)) {
#pragma unroll
			for (int k = 0; k < 16; k++) asm volatile("" ::"v"(dd[k]));
			asm volatile("" ::"v"(hv));
		};
		for (uint32_t t = 0; t < T; t += 2) {
			issue(dB, hB, ip, ib); MC33_ADV_ISSUE();
			drop(dA, hA);
			issue(dA, hA, ip, ib); MC33_ADV_ISSUE();
			drop(dB, hB);
		}
		return;
	}
#endif
	if constexpr (MC33_SWEEP_BUFS == 3 && NI == 1) {
		// Three buffers (round 4, single-isovalue passes): TWO batches stay in flight while one is turned into bit rows - also across
		// the work on a complete plane (cut-cell test, hand-over stores), which with two buffers ran with one batch in flight.
		raw_t dC[16];
		real_t hC = 0;
		issue(dB, hB, ip, ib); MC33_ADV_ISSUE();
		for (uint32_t t = 0; t < T; t += 3) {
			issue(dC, hC, ip, ib); MC33_ADV_ISSUE();
			process(dA, hA, pp, pb); MC33_ADV(pp, pb);
			issue(dA, hA, ip, ib); MC33_ADV_ISSUE();
			if (t + 1 < T) { process(dB, hB, pp, pb); MC33_ADV(pp, pb); }
			issue(dB, hB, ip, ib); MC33_ADV_ISSUE();
			if (t + 2 < T) { process(dC, hC, pp, pb); MC33_ADV(pp, pb); }
		}
	} else
	for (uint32_t t = 0; t < T; t += 2) {
		issue(dB, hB, ip, ib); MC33_ADV_ISSUE();
		process(dA, hA, pp, pb); MC33_ADV(pp, pb);
		issue(dA, hA, ip, ib); MC33_ADV_ISSUE();
		if (t + 1 < T) { process(dB, hB, pp, pb); MC33_ADV(pp, pb); }
	}
	log_flush();  // (DEFER: everything the tile hands on, behind its last load)
	if constexpr (NI >= 2) {
#pragma unroll
		for (int q = 0; q < NI; q++)
			if (pend_sum[q] && lane == 0) atomicAdd(a.lane[q].slot_part + pend_chunk[q], pend_sum[q]);
	}
	if (a.trace && lane == 0) {
		unsigned long long *tr = a.trace + 4ull * wtile;
		tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = c_start; tr[3] = __builtin_amdgcn_s_memtime();
	}
#undef MC33_ADV_ISSUE
#undef MC33_ADV
#undef MC33_LOAD
}

// ---------------------------------------------------------------------------------------------------
// k_boundary: the slice between the last plane of a tile and the first plane of the tile above it, from the bit
// rows the two left behind.  One wave per pair of tiles.
// ---------------------------------------------------------------------------------------------------
struct TileBoundary { uint32_t below, above, z, yt, seg, pad_[3]; };  // tile (wave) indices of k_sweep; slice z

__global__ __launch_bounds__(256) void k_boundary(const SweepArgs a, const TileBoundary *bounds, uint32_t nbounds) {
	const SweepLane &L = a.lane[blockIdx.y];  // (the isovalue lanes of the sweep that left the edge records)
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t bi = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	if (bi >= nbounds) return;
	const TileBoundary b = bounds[bi];
	const Params &P = a.P;
	const uint32_t seg = b.seg;
	const uint64_t rp = (uint64_t)b.below * 2u + 1u, rc = (uint64_t)b.above * 2u;  // top of below, bottom of above
	const uint4 p0 = L.edge_bits[rp * 128u + lane], p1 = L.edge_bits[rp * 128u + 64u + lane];
	const uint4 c0 = L.edge_bits[rc * 128u + lane], c1 = L.edge_bits[rc * 128u + 64u + lane];
	const uint4 hp = L.edge_hdr[rp * 2u], hc = L.edge_hdr[rc * 2u], zp = L.edge_hdr[rp * 2u + 1u], zc = L.edge_hdr[rc * 2u + 1u];
	// (both forms of both records are asked for at once - which one a plane has stands in its header, zp.z / zc.z)
	const uint32_t dp = ((const uint32_t *)(L.edge_bits + rp * 128u))[lane], dc = ((const uint32_t *)(L.edge_bits + rc * 128u))[lane];
	uint64_t prev[4] = {u64(p0.x, p0.y), u64(p0.z, p0.w), u64(p1.x, p1.y), u64(p1.z, p1.w)};
	uint64_t cur[4] = {u64(c0.x, c0.y), u64(c0.z, c0.w), u64(c1.x, c1.y), u64(c1.z, c1.w)};
	if (zp.z == PLANE_COMPACT) decode_row(dp, prev);
	if (zc.z == PLANE_COMPACT) decode_row(dc, cur);
	if (zp.z >= PLANE_UNIFORM0) { prev[0] = prev[1] = prev[2] = prev[3] = zp.z == PLANE_UNIFORM1 ? ~0ull : 0ull; }  // (no record was written: what was loaded is an older extraction's)
	if (zc.z >= PLANE_UNIFORM0) { cur[0] = cur[1] = cur[2] = cur[3] = zc.z == PLANE_UNIFORM1 ? ~0ull : 0ull; }
	const uint64_t bp = u64(hp.x, hp.y), bc = u64(hc.x, hc.y);
	uint64_t valid[4], act[4];
	valid_masks(seg * SEG_CELLS, P.nx, valid);
	const bool rowvalid = lane < 63u && b.yt * 63u + lane < P.ny;
	active_cells(prev, cur, (uint32_t)((bp >> lane) & 1ull), (uint32_t)((bc >> lane) & 1ull), valid, rowvalid, act);
	if (__ballot((act[0] | act[1] | act[2] | act[3]) != 0ull))
		// (the two tiles may have written these planes for slices of their own: same bytes again)
		hand_over_slice<1>(L, slice_slot(b.z - P.zs, b.yt, seg, a.sd), slice_slot(b.z + 1u - P.zs, b.yt, seg, a.sd), prev, cur,
		                true, true, bp, bc, u64(hp.z, hp.w) | u64(hc.z, hc.w), u64(zp.x, zp.y) | u64(zc.x, zc.y), act, lane);
}

// Lists of the slow cells and of the row segments they make "dirty".  k_cells appends to them; ONE cursor for the whole
// grid meant one atomic address that every wave with a slow cell queues at (an integer grid with an integer isovalue
// has such cells all along the surface: the atomics alone made k_cells 1.0 ms instead of 0.2).  So the slice slots are
// cut into at most LIST_CHUNKS (1024) groups of 2^shift consecutive slots; group g appends - with its own cursor - into the part
// of the list that starts at the index of the group's first work record: a group cannot hold more slow cells (or dirty
// rows) than records, so the parts cannot collide, and nothing has to be sized.  The consumers (k_slow_plan, k_seg_fix,
// k_emit_slow) turn a flat index into (group, position) with a prefix sum of the group counts, rebuilt by every block
// in LDS.
constexpr uint32_t LIST_CHUNKS = 1024;  // (measured at 1024^3 with 240 000 slow cells: 256 groups still queue, 1024 and 4096 do not)
struct ListChunks {
	uint32_t *slow_cnt, *dirty_cnt;  // [n]
	uint32_t n, shift;               // groups, log2 of slots per group
};

struct ChunkMap {  // per block, in LDS: exclusive prefix sums of the group counts
	uint32_t *pre;   // [LIST_CHUNKS + 1]
	uint32_t n, total;
	// all threads of the block (256): thread t takes groups 4 t .. 4 t + 3 (one 16-byte load); red: 256 words of scratch
	__device__ void build(uint32_t *lds, uint32_t *red, const uint32_t *cnt, uint32_t n_) {
		pre = lds; n = n_;
		const uint32_t t = threadIdx.x;
		uint4 v = 4u * t < n ? ((const uint4 *)cnt)[t] : uint4{0u, 0u, 0u, 0u};  // (the cursors beyond n are zero: k_slots clears them all)
		const uint32_t sum = v.x + v.y + v.z + v.w;
		red[t] = sum;
		__syncthreads();
		for (uint32_t d = 1; d < 256; d <<= 1) {
			const uint32_t x = t >= d ? red[t - d] : 0u;
			__syncthreads();
			red[t] += x;
			__syncthreads();
		}
		const uint32_t run = red[t] - sum;
		pre[4u * t] = run; pre[4u * t + 1u] = run + v.x; pre[4u * t + 2u] = run + v.x + v.y; pre[4u * t + 3u] = run + v.x + v.y + v.z;
		if (t == 255) pre[LIST_CHUNKS] = red[t];
		__syncthreads();
		total = pre[LIST_CHUNKS];
	}
	// flat index -> group g with pre[g] <= i < pre[g + 1]
	__device__ uint32_t group_of(uint32_t i) const {
		uint32_t lo = 0, hi = LIST_CHUNKS;  // invariant: pre[lo] <= i < pre[hi]
#pragma unroll
		for (int s = 0; s < 10; s++) {
			const uint32_t mid = (lo + hi) >> 1;
			const bool right = pre[mid] <= i;
			lo = right ? mid : lo;
			hi = right ? hi : mid;
		}
		return lo;
	}
};

// ---------------------------------------------------------------------------------------------------
// k_slots: exclusive prefix sums of (cut cells, batches of 64 of them) over the slice slots in slot order = the
// work-record range and the range of batch descriptors of every slice.  The sweep has already added every slice
// into the partial sum of its chunk of SLOT_CHUNK slots; block c sums the partials below c and scans its
// own chunk.  Record order is therefore a function of the grid alone (no allocation atomics).  The slots with cut cells
// are also listed, for k_cells.
// ---------------------------------------------------------------------------------------------------
struct SlotsArgs {
	const SliceHeader *hdr;
	const unsigned long long *part;
	unsigned long long *part_next;
	uint32_t part_cap, epoch;
	uint2 *slot_base;
	Counters *ctr;
	ListChunks lc;
	unsigned long long *scan_state;
	uint32_t scan_words;
	uint32_t *live_list;
	uint32_t live_cap;
};
__global__ __launch_bounds__(256) void k_slots(const PerLane<SlotsArgs> A, uint64_t nslots) {
	const SlotsArgs &sa = A.a[blockIdx.y];
	const SliceHeader *hdr = sa.hdr;
	const unsigned long long *part = sa.part;
	unsigned long long *part_next = sa.part_next;
	const uint32_t part_cap = sa.part_cap, epoch = sa.epoch, scan_words = sa.scan_words, live_cap = sa.live_cap;
	uint2 *slot_base = sa.slot_base;
	Counters *ctr = sa.ctr;
	const ListChunks lc = sa.lc;
	unsigned long long *scan_state = sa.scan_state;
	uint32_t *live_list = sa.live_list;
	__shared__ unsigned long long s_red[256];
	__shared__ uint32_t s_live[256], s_live_base;
	const uint32_t c = blockIdx.x, t = threadIdx.x;
	for (uint32_t q = c * 256u + t; q < scan_words; q += gridDim.x * 256u) scan_state[q] = 0;  // the group sums of this extraction's scan (k_scan_reduce adds to them)
	if (t == 0) part_next[c] = 0;  // the partial sums of the NEXT extraction live in the other half: cleared here
	if (c == 0) for (uint32_t q = gridDim.x + t; q < part_cap; q += 256u) part_next[q] = 0;  // (a later range may be longer)
	if (c == 0) for (uint32_t q = t; q < LIST_CHUNKS; q += 256u) { lc.slow_cnt[q] = 0; lc.dirty_cnt[q] = 0; }  // the list cursors of this extraction
	if (c == 0 && t == 0) {        // ... and so are the counters the later passes of this one add to
		ctr->slow_cursor = 0; ctr->dirty_cursor = 0; ctr->emit_skipped = 0; ctr->count_pending = 0; ctr->slow_barrier = 0;
		ctr->totV = ctr->totT = ctr->ghostV = ctr->ghostT = 0;
		for (int q = 0; q < 8; q++) ctr->debug[q] = 0;
	}
	unsigned long long below = 0;
	for (uint32_t q = t; q < c; q += 256u) below += part[q];
	s_red[t] = below;
	__syncthreads();
	for (uint32_t d = 128; d; d >>= 1) {
		if (t < d) s_red[t] += s_red[t + d];
		__syncthreads();
	}
	const unsigned long long base = s_red[0];
	__syncthreads();
	constexpr uint32_t PER = SLOT_CHUNK / 256;
	const uint64_t s0 = (uint64_t)c * SLOT_CHUNK + (uint64_t)t * PER;
	uint32_t cells[PER], rows[PER];
	unsigned long long mine = 0;
#pragma unroll
	for (uint32_t k = 0; k < PER; k++) {
		const bool in = s0 + k < nslots;
		const uint4 h = in ? *(const uint4 *)((const uint32_t *)(hdr + s0 + k) + 4) : uint4{0, 0, 0, 0};  // {curh_hi, cells, rows, pad}
		const bool valid = in && slice_valid(hdr[s0 + k].flags, epoch);
		cells[k] = valid ? h.y : 0u; rows[k] = (cells[k] + 63u) >> 6;  // (second sum: batches of 64 records, as the sweep added them)
		mine += (unsigned long long)rows[k] << 32 | cells[k];
	}
	uint32_t nlive = 0;  // slots of this thread with cut cells
#pragma unroll
	for (uint32_t k = 0; k < PER; k++) nlive += cells[k] ? 1u : 0u;
	s_red[t] = mine; s_live[t] = nlive;
	__syncthreads();
	for (uint32_t d = 1; d < 256; d <<= 1) {  // inclusive scan over the threads
		const unsigned long long v = t >= d ? s_red[t - d] : 0ull;
		const uint32_t w = t >= d ? s_live[t - d] : 0u;
		__syncthreads();
		s_red[t] += v; s_live[t] += w;
		__syncthreads();
	}
	// The slots with cut cells, listed for k_cells (a third of the slots of a smooth field: a wave per SLOT spent 88 us at
	// 1024^3 mostly being launched - 69 632 waves at the 870 per microsecond this GPU starts them at, two thirds of them to
	// find their slice empty; round 3).  The list's order is whatever order the blocks of this kernel arrive in: it decides
	// which wave of k_cells takes which slice and nothing else - where a slice's records go is slot_base.
	if (t == 255) s_live_base = s_live[255] ? atomicAdd(&ctr->live_cursor, s_live[255]) : 0u;
	__syncthreads();
	{
		uint32_t at = s_live_base + s_live[t] - nlive;
#pragma unroll
		for (uint32_t k = 0; k < PER; k++)
			if (cells[k]) { if (at < live_cap) live_list[at] = (uint32_t)(s0 + k); at++; }  // (bounded: the cursor is only as clean as the tail before left it)
	}
	unsigned long long run = base + s_red[t] - mine;
#pragma unroll
	for (uint32_t k = 0; k < PER; k++) {
		if (s0 + k < nslots) slot_base[s0 + k] = uint2{(uint32_t)run, (uint32_t)(run >> 32)};
		run += (unsigned long long)rows[k] << 32 | cells[k];
	}
	if (c == gridDim.x - 1 && t == 255) {  // totals; 32-bit fields (a carry out of the cells would also exceed every capacity)
		const unsigned long long tot = base + s_red[255];
		ctr->entry_cursor = (tot & 0xFFFFFFFFull) > 0xFFFFFF00ull ? 0xFFFFFFFFu : (uint32_t)tot;
		ctr->batch_cursor = (uint32_t)(tot >> 32);
	}
}

// ---------------------------------------------------------------------------------------------------
// k_cells: turns the slice records of the sweep into work records; a wave per slice at a time (the waves take the slices
// with cut cells off k_slots' list), waves independent.
// Lane = row for the bookkeeping (activity masks, per-row counts, directory); for the cells themselves the
// wave takes 64 cells at a time in record order (row, then x): lane g finds its row by a search in the
// prefix sums of the row counts, its cell as the n-th set bit of the row's activity mask, reads the 8
// corner bits from the bit rows (LDS), and finishes FAST cells (interior, group-0 table word, no corner
// equal to iso) from the sign index via the LDS table; the in-segment vertex / triangle offsets are a
// segmented scan over the 64 cells.  Records of a slice are written as one contiguous run.
// ---------------------------------------------------------------------------------------------------
// The records of one slice slot, 64 at a time: what a wave of the emit passes works on.  Everything a wave needs to know
// about where its records live is wave-uniform and comes from here (one scalar load): no division per record, and all 64
// records share their three sample planes and their 63 cell rows - which is what lets the vertex pass stage the sample
// rows of a batch in LDS.  Written by k_cells; slot s owns the descriptors [slot_base[s].y, slot_base[s + 1].y).
struct alignas(32) BatchDesc {
	uint32_t first, count;   // work records [first, first + count), count <= 64
	uint32_t sidx0;          // row-segment index (entry_seg) of cell row 0 of the slot's tile: entry_seg - sidx0 = row in the tile
	uint32_t z, y0, xbase;   // cell slice, first cell row of the y tile, first cell of the row segment
	uint32_t pad_[2];
};

// A row segment's counts in seg_cnt: {vertices: 12 bits, triangles: 12 bits, tag: 8 bits}.  The tag names the tail (k_slots ...
// k_scan_apply) that wrote the word, 1 .. 255 in turn; a word with another tag counts as zero.  So nobody writes the counts of
// the row segments that hold nothing - k_cells only looks at slices with cut cells (round 3) - and nobody clears the array
// between extractions (every 255 tails the host does).  A segment of 256 cells has at most 256 x 9 vertices (a row of the
// y = 0, z = 0 edge of the grid, every edge of every cell cut) and 256 x 12 triangles.
constexpr uint32_t SEG_TAGS = 255u;
static_assert(SEG_CELLS * 9u + 4u < 4096u && SEG_CELLS * 12u < 4096u, "a row segment's vertex / triangle counts must fit the 12-bit fields of seg_tagged");
__host__ __device__ inline uint32_t seg_tagged(uint32_t nv, uint32_t nt, uint32_t tag) { return nv | nt << 12 | tag << 24; }
__host__ __device__ inline uint32_t seg_counts(uint32_t word, uint32_t tag) { return (word >> 24) == tag ? (word & 0xFFFFFFu) : 0u; }  // nv | nt << 12
struct CellsArgs {
	uint32_t dev;            // (-DMC33_DEV: MC33_HIP_CELLS_DEV experiments)
	uint32_t pack;           // samples per lane and load of the sweep that made the records (lane_of_column)
	GridView<sample_t> G;    // (only looked at for cells of rows that may hold a sample equal to the isovalue)
	Params P;
	const uint4 *fast;       // per sign index: record words of a FAST cell (fast_record_table)
	uint32_t ze;
	SlotDims sd;
	const uint32_t *live_list;  // slots with cut cells (k_slots)
	uint32_t seg_tag;           // seg_tagged
	uint32_t live_cap;
	const SliceHeader *slice_hdr;
	const uint4 *slice_bits;
	const uint32_t *slice_compact;
	const uint8_t *plane_fmt;
	uint32_t epoch;          // number of this extraction: headers written by earlier ones are not valid
	const uint2 *slot_base;  // [slice_slot]: {first work record, first batch descriptor} (k_slots)
	uint32_t *seg_cnt;
	SegDir *seg_dir;
	EntryA *entries_a;       // work records, half A
	EntryB *entries_b;       // ... half B: written here for TESTED cells, by k_slow_plan for slow ones, never for fast ones
	const uint32_t *pat;     // what an interior cell makes of each pattern of the reference's table (build_pattern_info): for the
	                         // cells whose sign index needs the face / interior tests (their table word rides in `fast`)
	uint32_t *entry_seg;
	uint32_t *slow_list, *dirty_list;
	ListChunks lc;
	uint32_t entry_cap;
	BatchDesc *batches;
	uint32_t batch_cap;
	Counters *ctr;
	unsigned long long *trace;  // MC33_HIP_TRACE_CELLS: per wave {start, bits in, rows done, end}
};

// n-th (0-based) set bit of w; n < popcount(w)
__device__ __forceinline__ uint32_t nth_set_bit(uint64_t w, uint32_t n) {
	// the half first, then five halving steps on 32 bits
	const uint32_t lo = (uint32_t)w, clo = (uint32_t)__popc(lo);
	const bool hi = n >= clo;
	n -= hi ? clo : 0u;
	uint32_t v = hi ? (uint32_t)(w >> 32) : lo, pos = hi ? 32u : 0u;
#pragma unroll
	for (int width = 16; width; width >>= 1) {
		const uint32_t c = (uint32_t)__popc(v & ((1u << width) - 1u));
		const bool up = n >= c;
		n -= up ? c : 0u;
		v = up ? v >> width : v;
		pos += up ? (uint32_t)width : 0u;
	}
	return pos;
}

struct CellsLds {            // per wave
	uint64_t bits[64][9];    // row r: prev[0..3], cur[0..3] (row 63 of the tile is only ever the row above); the
	uint64_t act[64][5];     // odd row pitches keep neighbouring rows on different LDS banks
	uint32_t incl[64], run[64], slowrow[64];
	uint32_t rowof[64];      // per batch of 64 cells: row + 1 at the position of the row's first cell, 0 elsewhere
};

// The 8 samples of one cell: {0xFFFFFFFF, 0} if one of them equals the isovalue; else, for a sign index i that needs the
// face / interior tests (i = 0: none wanted), {offset of the pattern the tests choose, its pattern-info word}.
__device__ __forceinline__ uint2 corner_look(const GridView<sample_t> &G, real_t iso, uint32_t lut_word, const uint32_t *pat,
                                                        uint32_t x, uint32_t y, uint32_t z, uint32_t i) {
	Corner8 v;
	bool zero = false;
#pragma unroll
	for (uint32_t k = 0; k < 8; k++) {
		const uint32_t cc = corner_code(k);
		v.a[k] = iso_diff(iso, (real_t)G.at(x + (cc & 1u), y + ((cc >> 1) & 1u), z + (cc >> 2)));
		zero |= v.a[k] == 0;
	}
	if (zero) return uint2{0xFFFFFFFFu, 0u};
	if (!i) return uint2{0u, 0u};
	uint32_t wm, wn;
	const uint32_t poff = pattern_offset_word(lut_word, i, v, wm, wn);  // (the word came with the LDS table entry: no load of its own)
	return uint2{poff, pat[poff]};
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_cells(const PerLane<CellsArgs> A) {  // (4 waves per SIMD is what its LDS allows: keep the registers of the rare test code from costing one)
	const CellsArgs &a = A.a[blockIdx.y];
	__shared__ uint4 s_fast[256];
	__shared__ CellsLds s_w[4];
	s_fast[threadIdx.x] = a.fast[threadIdx.x];
	const uint32_t lane = threadIdx.x & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	CellsLds &L = s_w[wv];
	const Params &P = a.P;
	// As many waves as the GPU holds (the host sizes the grid), each taking slices off k_slots' list of slots with cut cells:
	// nobody is launched to find a slice empty.  (The list entry of the wave's NEXT slice fetched one slice ahead through the
	// scalar cache, so that a slice starts with one round trip instead of two: no change, 67.6 us either way - with the empty
	// waves gone the kernel is within a quarter of what its 32 M vector instructions take.)
	const uint32_t nlive = min(a.ctr->live_cursor, a.live_cap);
	__syncthreads();  // s_fast
	for (uint32_t item = blockIdx.x * 4u + wv; item < nlive; item += gridDim.x * 4u) {
	const unsigned long long t_start = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
	const uint64_t slot = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.live_list[item]);
	uint32_t seg, yt, zq;
	slot_group_coords((uint32_t)(slot >> 2), a.sd, zq, yt, seg);
	const uint32_t xbase = seg * SEG_CELLS, y0 = yt * 63u;
	const uint32_t z = P.zs + zq * 4u + (uint32_t)(slot & 3u);
	const uint32_t y = y0 + lane;
	const bool in_grid = seg < P.nseg && z < a.ze;
	SliceHeader h;
	h.flags = 0; h.cells = 0;
	uint32_t dl = 0, du = 0, fmt_l = PLANE_COMPACT, fmt_u = PLANE_COMPACT;
	uint64_t slot_up = 0;
	uint2 base = {0u, 0u};
	uint32_t lbase = 0;
	if (in_grid) {  // header, ranges and bit rows are fetched together (one round trip); the rows of a slice
		// without cut cells are whatever an earlier call left there and are not looked at.  (Fetching the bit rows only
		// once the header says the slice is cut - two thirds of the slots of a smooth field are not - saves 180 MB of
		// reads at 1024^3 and no time: measured, round 2)
		h = a.slice_hdr[slot];
		base = a.slot_base[slot];
		lbase = a.slot_base[(slot >> a.lc.shift) << a.lc.shift].x;  // where the list part of the slot's group begins
		// the records of the two planes of the slice (the upper plane's sits in the slot of the slice above): their compact
		// form, one dword per row, and how they are written; a plane in raw form (a row with more than two changes) costs
		// a second round trip below
		slot_up = slice_slot(z + 1u - P.zs, yt, seg, a.sd);
		dl = a.slice_compact[slot * 64u + lane];
		du = a.slice_compact[slot_up * 64u + lane];
		fmt_l = a.plane_fmt[slot]; fmt_u = a.plane_fmt[slot_up];
	}
	const bool live = in_grid && slice_valid(h.flags, a.epoch);  // wave-uniform
	const bool rowvalid = lane < 63u && y < P.ny;
	const uint64_t sidx = ((uint64_t)(z - P.zs) * P.nseg + seg) * P.ny + y;  // storage order [z][segment][y]
	if (!live) continue;  // (cannot be: the slot is on the list.  The counts of row segments nobody writes count as zero: their tag is an older tail's, seg_counts)
	uint64_t prev[4], cur[4], act[4];
	{
		const bool raw_l = __builtin_amdgcn_readfirstlane((int)fmt_l) != (int)PLANE_COMPACT, raw_u = __builtin_amdgcn_readfirstlane((int)fmt_u) != (int)PLANE_COMPACT;
		uint4 q[4] = {};
		if (raw_l) { const uint4 *lower = a.slice_bits + slot * 128u + lane; q[0] = lower[0]; q[1] = lower[64]; }
		if (raw_u) { const uint4 *upper = a.slice_bits + slot_up * 128u + lane; q[2] = upper[0]; q[3] = upper[64]; }
		if (raw_l) { prev[0] = u64(q[0].x, q[0].y); prev[1] = u64(q[0].z, q[0].w); prev[2] = u64(q[1].x, q[1].y); prev[3] = u64(q[1].z, q[1].w); }
		else decode_row(dl, prev);
		if (raw_u) { cur[0] = u64(q[2].x, q[2].y); cur[1] = u64(q[2].z, q[2].w); cur[2] = u64(q[3].x, q[3].y); cur[3] = u64(q[3].z, q[3].w); }
		else decode_row(du, cur);
	}
	const uint64_t bp = u64(h.prevh_lo, h.prevh_hi), bc = u64(h.curh_lo, h.curh_hi);  // halo-column bits of the rows
	{
		uint64_t valid[4];
		valid_masks(xbase, P.nx, valid);
		active_cells(prev, cur, (uint32_t)((bp >> lane) & 1ull), (uint32_t)((bc >> lane) & 1ull), valid, rowvalid, act);
	}
	const uint32_t c0 = __popcll(act[0]), c1 = __popcll(act[1]), c2 = __popcll(act[2]);
	const uint32_t cnt = c0 + c1 + c2 + __popcll(act[3]);
	const uint32_t incl = wave_scan_add(cnt);  // inclusive prefix of the per-row counts over the lanes
	const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
	const unsigned long long t_bits = a.trace ? __builtin_amdgcn_s_memrealtime() + (total & 0u) : 0ull;
	const uint32_t ebase = base.x;
	{  // the slot's records in batches of 64 for the emit passes
		const uint32_t nb = (total + 63u) >> 6;
		for (uint32_t k = lane; k < nb; k += 64u)
			if (base.y + k < a.batch_cap) {
				BatchDesc bd;
				bd.first = ebase + 64u * k; bd.count = min(64u, total - 64u * k);
				bd.sidx0 = (uint32_t)(((uint64_t)(z - P.zs) * P.nseg + seg) * P.ny + y0);
				bd.z = z; bd.y0 = y0; bd.xbase = xbase; bd.pad_[0] = bd.pad_[1] = 0;
				a.batches[base.y + k] = bd;
			}
	}

	// rows whose cells cannot take the fast path: on the y = 0 / z = 0 faces (extra owned edges), or in a
	// tile plane pair that holds a sample equal to the isovalue
	const uint64_t zc = u64(h.zc_lo, h.zc_hi);  // ... and the sweep lanes that loaded one
	const uint64_t zr = u64(h.zr_lo, h.zr_hi);  // sample rows with a sample equal to the isovalue: the cells of rows r - 1 and r
#pragma unroll
	for (int k = 0; k < 4; k++) { L.bits[lane][k] = prev[k]; L.bits[lane][4 + k] = cur[k]; L.act[lane][k] = act[k]; }
	L.bits[lane][8] = (bc >> lane) & 1ull;  // the halo-column bit of the upper plane's row: the bit after its last word
	L.incl[lane] = incl;
	L.run[lane] = 0;
	// bit 0: no cell of the row can take the fast path (grid faces); bit 2: a corner may equal the isovalue - the cell's
	// own 8 samples decide; bit 1 is set when a cell of the row went to the slow list
	L.slowrow[lane] = ((y == 0 || z == 0) ? 1u : 0u) | (((zr >> lane) & 3ull) ? 4u : 0u);
	const uint32_t first = ebase + incl - cnt;
	const unsigned long long t_rows = a.trace ? __builtin_amdgcn_s_memrealtime() + (first & 0u) : 0ull;
	const uint64_t sidx0 = ((uint64_t)(z - P.zs) * P.nseg + seg) * P.ny + y0;

	uint32_t carry_row = 0;
	for (uint32_t g0 = 0; g0 < total; g0 += 64u) {  // wave-uniform
		const uint32_t g = g0 + lane;
		const bool on = g < total;
		// the row of cell g: the rows that begin inside this batch mark the position of their first cell, the cells after
		// it follow by a running maximum over the lanes (rows come in rising order), and the cells before the first mark
		// belong to the row the previous batch ended in.  (Round 1: a binary search in the prefix sums, six dependent LDS reads)
		L.rowof[lane] = 0u;
		if (cnt && incl - cnt >= g0 && incl - cnt < g0 + 64u) L.rowof[incl - cnt - g0] = lane + 1u;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the marks of the other lanes: not a value this thread could know)
		const uint32_t mark = wave_scan_max(L.rowof[lane]);
		const uint32_t r = on ? (mark ? mark - 1u : carry_row) : 62u;  // (row 63 of a tile is never a cell row)
		carry_row = __builtin_amdgcn_readlane(r, 63);
		const uint64_t a0 = L.act[r][0], a1 = L.act[r][1], a2 = L.act[r][2], a3 = L.act[r][3];
		const uint32_t p0 = __popcll(a0), p1 = p0 + __popcll(a1), p2 = p1 + __popcll(a2), rowcnt = p2 + __popcll(a3);
		const uint32_t kin = on ? g - (L.incl[r] - rowcnt) : 0u;  // position of the cell among the cells of its row
		const uint32_t wsel = (kin >= p0) + (kin >= p1) + (kin >= p2);
		const uint64_t aw = wsel == 0 ? a0 : wsel == 1 ? a1 : wsel == 2 ? a2 : a3;
		const uint32_t xl = 64u * wsel + (on ? nth_set_bit(aw, kin - (wsel == 0 ? 0u : wsel == 1 ? p0 : wsel == 2 ? p1 : p2)) : 0u);
		// the 8 corner bits: sample x and x+1 of rows r, r+1 on the two planes
		// (the bit rows as dwords: dword 2 w + h of a plane's row holds the bits 32 h .. 32 h + 31 of word w; the two bits are
		// taken from the 64-bit window that begins at the dword of x - the dword after the prev plane's last one is replaced
		// by the halo bit, the one after the cur plane's last one IS the halo bit: bits[r][8], set above)
		const uint32_t didx = xl >> 5, bit = xl & 31u;
		uint32_t i = 0;
#pragma unroll
		for (int c = 0; c < 4; c++) {  // c: 0 = (row r, prev) 1 = (row r+1, prev) 2 = (row r+1, cur) 3 = (row r, cur): corners 0..3, MC:1846-1859
			const uint32_t rr = r + ((c == 1 || c == 2) ? 1u : 0u), pl = (c >= 2) ? 4u : 0u;
			const uint32_t *rowp = (const uint32_t *)&L.bits[rr][pl];
			const uint32_t w32 = rowp[didx];
			uint32_t n32 = rowp[didx + 1u];
			if (c < 2) n32 = didx == 7u ? (uint32_t)(bp >> rr) & 1u : n32;
			const uint32_t t = __builtin_amdgcn_alignbit(n32, w32, bit);
			i |= (t & 1u) << (7 - c) | ((t >> 1) & 1u) << (3 - c);
		}
		const uint4 f = s_fast[i];
		const uint32_t rowflag = L.slowrow[r];
		bool zero_corner = false;
		// (its row may hold such a sample, and so may one of its two columns: then the cell's own 8 samples decide)
		const bool look = on && (rowflag & 5u) == 4u && f.x != FAST_NONE &&
		                  (xl == 255u || ((zc >> lane_of_column(xl, a.pack)) | (zc >> lane_of_column(min(xl + 1u, 255u), a.pack))) & 1ull);
		// an interior cell whose sign index needs the face / interior tests: the tests are made here on its 8 samples, and
		// unless one of them equals the isovalue the cell is finished like a fast one (TESTED record, mc33_cell.h)
		const bool amb = on && !(rowflag & 1u) && f.x == FAST_NONE && (xbase + xl) != 0;
		uint32_t tpoff = 0, tinfo = 0;
#ifdef MC33_DEV
		if (a.dev & 1u) zero_corner = look;  // experiment: no look at the samples (every candidate goes the slow way)
		if (__ballot(look || amb) && !(a.dev & 1u)) {
#else
		if (__ballot(look || amb)) {  // wave-uniform
#endif
			if (look || amb) {
				const uint2 t = corner_look(a.G, P.iso, f.y, a.pat, xbase + xl, y0 + r, z, amb ? i : 0u);
				zero_corner = t.x == 0xFFFFFFFFu;
				if (!zero_corner) { tpoff = t.x; tinfo = t.y; }
			}
		}
		const bool tested = tinfo != 0;  // (a pattern has at least one triangle)
		const bool fastcell = on && !(rowflag & 1u) && !zero_corner && f.x != FAST_NONE && (xbase + xl) != 0;
		// new vertices | triangles << 16
		const uint32_t val = fastcell ? ((f.w & 0xFFu) | (f.w >> 8) << 16) : tested ? (((tinfo >> 20) & 15u) | ((tinfo >> 16) & 15u) << 16) : 0u;
		// offsets inside the row segment: exclusive scan over the cells of the same row
		const uint32_t sc = wave_scan_add(val);
		// ... minus the scan value before the first cell of my row inside this batch: both halves of the packed sums only
		// grow along the lanes, so that is the running maximum of the values at the row heads (0 when my row began earlier)
		const uint32_t before_head = wave_scan_max(on && kin == 0u ? sc - val : 0u);
		const uint32_t carry = kin > lane ? L.run[r] : 0u;  // the row began in an earlier batch
		const uint32_t off = carry + (sc - val) - before_head;
		{  // the slow cells of the batch go on the list of the slot's group, one atomic per wave
			const bool slowlane = on && !fastcell && !tested && ebase + g < a.entry_cap;
			const uint64_t sm = __ballot(slowlane);
			if (sm) {
				const uint32_t leader = (uint32_t)__builtin_ctzll(sm);
				uint32_t at = 0;
				if (lane == leader) at = atomicAdd(&a.lc.slow_cnt[slot >> a.lc.shift], (uint32_t)__popcll(sm));
				at = lbase + __shfl(at, leader) + (uint32_t)__popcll(sm & ((1ull << lane) - 1ull));
				if (slowlane && at < a.entry_cap) a.slow_list[at] = ebase + g;
			}
		}
		if (on) {
			const uint32_t ri = ebase + g;
			Entry e;
			if (fastcell) { e.w0 = f.x | xl; e.w1 = off; e.w2 = f.y; e.w3 = f.z; }
			else if (tested) e = make_tested_entry(xl, i, tpoff, tinfo, off & 0xFFFFu, off >> 16);
			else {
				e = make_pending_entry(xl, i);
				L.slowrow[r] = rowflag | 2u;
			}
			if (ri < a.entry_cap) {
				a.entries_a[ri] = entry_a(e);  // (half B of a fast record follows from its sign index; k_slow_plan writes the slow ones')
				if (tested) a.entries_b[ri] = entry_b(e);
				a.entry_seg[ri] = (uint32_t)(sidx0 + r);
			}
			if (kin + 1u == rowcnt || lane == 63u) L.run[r] = off + val;  // last cell of the row in this batch
		}
	}
	if (rowvalid && cnt) {
		const bool dirty = (L.slowrow[lane] & 2u) != 0;
		const uint32_t run = L.run[lane];
		if (!dirty) a.seg_cnt[sidx] = seg_tagged(run & 0xFFFFu, run >> 16, a.seg_tag);  // (a row with slow cells: k_seg_fix)
		const uint32_t nf = cnt | (dirty ? SEG_DIRTY : 0u);
		// one 64-byte line per row; only the words that hold cells are written (a lookup reads the word of an ACTIVE
		// cell), and word 0 of a row with slow cells (k_seg_fix takes the record range from it): the few cut cells of
		// a row mostly sit in one word, and these lines were the largest thing k_cells wrote
		uint4 *dq = (uint4 *)a.seg_dir[sidx].q;
		if (act[0] || dirty) dq[0] = uint4{(uint32_t)act[0], (uint32_t)(act[0] >> 32), first, nf};
		if (act[1]) dq[1] = uint4{(uint32_t)act[1], (uint32_t)(act[1] >> 32), first + c0, nf};
		if (act[2]) dq[2] = uint4{(uint32_t)act[2], (uint32_t)(act[2] >> 32), first + c0 + c1, nf};
		if (act[3]) dq[3] = uint4{(uint32_t)act[3], (uint32_t)(act[3] >> 32), first + c0 + c1 + c2, nf};
	}
	{  // rows with slow cells: the same
		const bool dirtylane = rowvalid && cnt && (L.slowrow[lane] & 2u) && first < a.entry_cap;
		const uint64_t dm = __ballot(dirtylane);
		if (dm) {
			const uint32_t leader = (uint32_t)__builtin_ctzll(dm);
			uint32_t at = 0;
			if (lane == leader) at = atomicAdd(&a.lc.dirty_cnt[slot >> a.lc.shift], (uint32_t)__popcll(dm));
			at = lbase + __shfl(at, leader) + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
			if (dirtylane && at < a.entry_cap) a.dirty_list[at] = (uint32_t)sidx;
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the wave's LDS record is written again for its next slice)
	if (a.trace && lane == 0) {
		unsigned long long *tr = a.trace + 4ull * slot;
		tr[0] = t_start; tr[1] = t_bits; tr[2] = t_rows; tr[3] = __builtin_amdgcn_s_memrealtime();
#ifdef MC33_TRACE_XCC  // (developer builds: which XCD and CU ran the wave, in place of the second stamp - HW_REG_XCC_ID, HW_REG_HW_ID)
		tr[1] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32;
#endif
	}
	}  // slices of this wave
}

// ---------------------------------------------------------------------------------------------------
// k_slow_plan: cells the sweep could not finish from the sign index (ambiguous MC33 cases: face and
// interior tests MC:347-462; cells on the x/y/z = 0 faces; corners equal to the isovalue MC:788-1224)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_sum(uint64_t x) {
#pragma unroll
	for (int d = 32; d; d >>= 1) x += __shfl_xor(x, d);
	return x;
}

struct SlowArgs {
	GridView<sample_t> G;
	Params P;
	Tables tab;
	uint32_t z_emit;  // slices below are ghosts of a z-slab
	EntryA *entries_a;
	EntryB *entries_b;
	EntryC *entries_c;       // plans of the slow records (k_slow_plan writes, k_slow_count and k_emit_slow follow them)
	const EntryB *fast_b;
	const uint32_t *entry_seg;
	const uint32_t *slow_list;
	uint32_t *seg_cnt;
	uint32_t seg_tag;
	const SegDir *seg_dir;
	const uint32_t *dirty_list;
	ListChunks lc;
	const uint2 *slot_base;
	uint32_t entry_cap;
	Counters *ctr;
};

// (the bodies of the three kernels as functions: each is a kernel of its own, and all three are the phases of k_slow_all)
__device__ __forceinline__ void slow_plan_body(const SlowArgs &a, real_t (*s_v)[256], uint32_t *s_pre, uint32_t *s_red, uint32_t *s_dirty) {
	ChunkMap cm;
	cm.build(s_pre, s_red, a.lc.slow_cnt, a.lc.n);  // (first: its loads and the one of the cursor below go out together)
	if (a.ctr->entry_cursor > a.entry_cap) return;  // the sweep will be repeated with more room
	const uint32_t n = cm.total;
	if (blockIdx.x == 0) {  // the totals of both lists, for the kernels that follow: their blocks beyond the lists leave at once
		uint32_t d = 0;
		for (uint32_t k = threadIdx.x; k < a.lc.n; k += 256u) d += a.lc.dirty_cnt[k];
		d = (uint32_t)wave_sum((uint64_t)d);
		if ((threadIdx.x & 63u) == 0) s_dirty[threadIdx.x >> 6] = d;
		__syncthreads();
		if (threadIdx.x == 0) { a.ctr->slow_cursor = n; a.ctr->dirty_cursor = s_dirty[0] + s_dirty[1] + s_dirty[2] + s_dirty[3]; }
	}
	if (blockIdx.x * 256u >= n) return;             // (nothing for this block: most blocks of most calls)
	const Tables &tab = a.tab;  // (the tables in LDS instead: tried - flat loads tie the LDS and memory wait counters together; slower)
	const VRef v{&s_v[0][threadIdx.x], 256};
	const Params &P = a.P;
	for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
		const uint32_t gq = cm.group_of(t);
		const uint32_t ei = a.slow_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])];
		const uint32_t s = a.entry_seg[ei];
		const uint32_t xl = a.entries_a[ei].a0 & 0xFFu;
		const SegCoord sc = segment_coord(P, s);
		const uint32_t y = sc.y, z = sc.z, x = sc.xbase + xl;
		const uint32_t i = load_cell(a.G, P.iso, x, y, z, v);
		CellPlan pl;
		plan_cell(pl, tab, P, a.G, x, y, z, i, v);
		// Triangles with two equal vertices are not appended (MC:1235): with a corner equal to the isovalue that is a question
		// of vertex IDENTITY, answered by k_slow_count once the plans of all slow cells are stored.  Ghost cells only lend vertex
		// ids to the slab above; their triangle count cancels out of every offset, so the identity test (which may follow a
		// reference one more plane down) is skipped.
		Entry en = make_entry(xl, i, pl, pl.ntri, 0, 0, true);
		if (cell_is_tested(pl, x, y, z)) en.w3 ^= ENTRY_SLOW | ENTRY_TESTED;  // the fast emit passes can write it
		else if (pl.zmask && z >= a.z_emit) { en.w3 |= ENTRY_COUNT; a.ctr->count_pending = 1u; }
		a.entries_a[ei] = entry_a(en);
		a.entries_b[ei] = entry_b(en);
		if (en.w3 & ENTRY_SLOW) a.entries_c[ei] = entry_c(pl);
	}
}
__global__ __launch_bounds__(256) void k_slow_plan(const PerLane<SlowArgs> A) {
	__shared__ real_t s_v[8][256];
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256], s_dirty[4];
	slow_plan_body(A.a[blockIdx.y], s_v, s_pre, s_red, s_dirty);
}

// the triangles of the slow cells that have a corner equal to the isovalue, counted by vertex identity on the stored plans
__device__ __forceinline__ void slow_count_body(const SlowArgs &a, real_t (*s_w)[256], uint64_t (*s_key)[256], uint32_t *s_pre, uint32_t *s_red) {
	if (!a.ctr->count_pending) return;  // (no sample of a slow cell equals the isovalue: most calls)
	ChunkMap cm;
	cm.build(s_pre, s_red, a.lc.slow_cnt, a.lc.n);
	if (a.ctr->entry_cursor > a.entry_cap) return;
	const uint32_t n = cm.total;
	if (blockIdx.x * 256u >= n) return;
	EmitCtx<sample_t> c;
	c.tab = a.tab; c.P = a.P; c.G = a.G;
	c.seg_base = nullptr; c.seg_dir = a.seg_dir;
	c.entries_a = a.entries_a; c.entries_b = a.entries_b; c.entries_c = a.entries_c; c.fast_b = a.fast_b; c.fast_b_in_lds = false; c.entry_seg = a.entry_seg;
	c.V = nullptr; c.N = nullptr; c.Tri = nullptr;
	c.z_emit = a.z_emit; c.v_skip = c.t_skip = c.id_delta = 0;
	const VRef w{&s_w[0][threadIdx.x], 256};
	for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
		const uint32_t gq = cm.group_of(t);
		const uint32_t ei = a.slow_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])];
		const EntryA ea = a.entries_a[ei];
		if (!(ea.a0 & ENTRYA_COUNT)) continue;
		const Entry en = entry_join(ea, a.entries_b[ei]);
		CellPlan pl;
		plan_restore(pl, a.tab.lut, en, a.entries_c[ei]);
		const SegCoord sc = segment_coord(a.P, a.entry_seg[ei]);
		RootMemo memo{&s_key[0][threadIdx.x], 256, 0u};
		const uint32_t nt = count_triangles_stored(c, pl, sc.xbase + (ea.a0 & 0xFFu), sc.y, sc.z, w, memo, (uint64_t)a.entry_seg[ei], ei);
		a.entries_a[ei].a0 = (ea.a0 & ~(15u << 20) & ~ENTRYA_COUNT) | nt << 20;
	}
}
__global__ __launch_bounds__(256) void k_slow_count(const PerLane<SlowArgs> A) {
	__shared__ real_t s_w[8][256];
	__shared__ uint64_t s_key[12][256];
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256];
	slow_count_body(A.a[blockIdx.y], s_w, s_key, s_pre, s_red);
}

// one thread per row segment that holds slow cells: running offsets of its records, segment totals.  A record that still waits
// for its triangle count (ENTRYA_COUNT) is counted on the way: k_slow_count, a launch of its own for exactly that, costs 5 - 6 us
// even when its blocks read one flag and leave - the case of nearly every extraction - so the host enqueues it only when the last
// extraction of the context had such records (then they are many, and a thread per RECORD is the faster way through them), and
// whatever is left over when it was not enqueued - the isovalue has moved onto the samples since - is caught here, a segment's
// records one after the other.
__device__ __forceinline__ void seg_fix_body(const SlowArgs &a, real_t (*s_w)[256], uint64_t (*s_key)[256], uint32_t *s_pre, uint32_t *s_red) {
	const uint32_t dirty_total = a.ctr->dirty_cursor, records = a.ctr->entry_cursor;  // (asked for together)
	const bool pending = a.ctr->count_pending != 0u;  // (k_slow_plan: some record waits for its count - unless k_slow_count has been through)
	if (blockIdx.x * 256u >= dirty_total) return;  // (k_slow_plan left the total there)
	ChunkMap cm;
	cm.build(s_pre, s_red, a.lc.dirty_cnt, a.lc.n);
	if (records > a.entry_cap) return;
	EmitCtx<sample_t> c;
	c.tab = a.tab; c.P = a.P; c.G = a.G;
	c.seg_base = nullptr; c.seg_dir = a.seg_dir;
	c.entries_a = a.entries_a; c.entries_b = a.entries_b; c.entries_c = a.entries_c; c.fast_b = a.fast_b; c.fast_b_in_lds = false; c.entry_seg = a.entry_seg;
	c.V = nullptr; c.N = nullptr; c.Tri = nullptr;
	c.z_emit = a.z_emit; c.v_skip = c.t_skip = c.id_delta = 0;
	const VRef w{&s_w[0][threadIdx.x], 256};
	const uint32_t n = cm.total;
	for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
		const uint32_t gq = cm.group_of(t);
		const uint32_t s = a.dirty_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])];
		const uint32_t first = a.seg_dir[s].q[0][2], cnt = a.seg_dir[s].q[0][3] & ~SEG_DIRTY;
		uint32_t nv = 0, nt = 0;
		for (uint32_t k = 0; k < cnt; k++) {
			EntryA *e = a.entries_a + first + k;  // (counts and offsets live in half A)
			EntryA ea = *e;
			if (pending && (ea.a0 & ENTRYA_COUNT)) {  // (as k_slow_count)
				const Entry en = entry_join(ea, a.entries_b[first + k]);
				CellPlan pl;
				plan_restore(pl, a.tab.lut, en, a.entries_c[first + k]);
				const SegCoord sc = segment_coord(a.P, s);
				RootMemo memo{&s_key[0][threadIdx.x], 256, 0u};
				const uint32_t ntri = count_triangles_stored(c, pl, sc.xbase + (ea.a0 & 0xFFu), sc.y, sc.z, w, memo, (uint64_t)s, first + k);
				ea.a0 = (ea.a0 & ~(15u << 20) & ~ENTRYA_COUNT) | ntri << 20;
				e->a0 = ea.a0;
			}
			e->a1 = nv | nt << 16;
			nv += entrya_nnew(ea);
			nt += entrya_ntri(ea);
		}
		a.seg_cnt[s] = seg_tagged(nv, nt, a.seg_tag);
	}
}
__global__ __launch_bounds__(256) void k_seg_fix(const PerLane<SlowArgs> A) {
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256];
	__shared__ real_t s_w[8][256];
	__shared__ uint64_t s_key[12][256];
	seg_fix_body(A.a[blockIdx.y], s_w, s_key, s_pre, s_red);
}

// The three as ONE launch (round 5; built, bit-identical, SLOWER - MC33_HIP_SLOW_MERGED=1 runs it, the library does not): for the
// usual case of FEW slow records - cells on the grid's faces, a corner equal to the isovalue here and there: 8 800 at 1024^3 are 35
// blocks' worth of work behind three launches, each a grid-wide dependency (plans must all be stored before identities are
// counted, counts before the offsets of a row segment are rebuilt).  The blocks of this kernel - as few as the last extraction's
// slow records need, 64 at most, all resident at once - pass two barriers instead: every block adds one to a counter in the
// isovalue's Counters (k_slots has zeroed it) when its share of a phase is stored, and waits until all have.  Measured at 1024^3
// (profiles/r05_tail_merge.txt): 39 us against 13 + 10 for k_slow_plan + k_seg_fix as launches of their own, the tail 0.141
// against 0.126 ms.  The blocks sit on eight XCDs, each behind an L2 of its own: a barrier between them is a write-back of that
// L2 (buffer_wbl2 sc1), a device-scope atomic, a polling loop on a line that comes from memory every time, and an invalidate
// (buffer_inv sc1) - ~8 us each, where the end of a kernel does the same for every XCD at once in ~1 us (the sum of the tail's
// kernel durations IS its event time: there is no gap between launches to win back).  The same holds for anything else that
// would fold a grid-wide dependency of the tail into a kernel - the slice between two sweep tiles done by whichever wave ends
// second, k_slots at the head of k_cells: each needs this release / acquire pair per wave or block.  Not pursued.
constexpr uint32_t SLOW_ALL_MAX_BLOCKS = 64;
__device__ __forceinline__ void slow_barrier(uint32_t *counter, uint32_t target) {
	__threadfence();   // (every thread: its stores of the phase, released to the device - the blocks run on different XCDs, each behind an L2 of its own)
	__syncthreads();
	if (threadIdx.x == 0) {
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
		while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
	}
	__syncthreads();
	__threadfence();   // (acquire: nothing read below may come from a line cached before the others' stores)
}
__global__ __launch_bounds__(256) void k_slow_all(const PerLane<SlowArgs> A) {
	const SlowArgs &a = A.a[blockIdx.y];
	__shared__ real_t s_w[8][256];
	__shared__ uint64_t s_key[12][256];
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256], s_dirty[4];
	slow_plan_body(a, s_w, s_pre, s_red, s_dirty);
	slow_barrier(&a.ctr->slow_barrier, gridDim.x);
	slow_count_body(a, s_w, s_key, s_pre, s_red);
	slow_barrier(&a.ctr->slow_barrier, 2u * gridDim.x);
	seg_fix_body(a, s_w, s_key, s_pre, s_red);
}

// ---------------------------------------------------------------------------------------------------
// prefix sums over the row segments (sweep order)
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t SCAN_PER_THREAD = 8, SCAN_CHUNK = 256 * SCAN_PER_THREAD, SCAN_GROUP = 32, SCAN_GROUPED_FROM = 4096;
__host__ __device__ inline uint64_t scan_groups(uint64_t nchunks) { return (nchunks + SCAN_GROUP - 1) / SCAN_GROUP; }


// position in sweep order -> storage index, advanced incrementally (one division per thread, not per element)
struct SweepWalk {
	uint32_t sg, y, nseg, ny;
	uint64_t zbase;  // z * nseg * ny
	__device__ SweepWalk(const Params &P, uint64_t q) : nseg(P.nseg), ny(P.ny) {
		if ((q >> 32) == 0) {  // (32-bit divisions where they do: a 64-bit one is ~80 instructions, and every load of the scan waits for two)
			const uint32_t q32 = (uint32_t)q, zy = q32 / P.nseg;
			sg = q32 - zy * P.nseg;
			const uint32_t z = zy / P.ny;
			y = zy - z * P.ny;
			zbase = (uint64_t)z * P.nseg * P.ny;
			return;
		}
		const uint64_t zy = q / P.nseg;
		sg = (uint32_t)(q % P.nseg);
		const uint64_t z = zy / P.ny;
		y = (uint32_t)(zy % P.ny);
		zbase = z * P.nseg * P.ny;
	}
	__device__ uint64_t store() const { return zbase + (uint64_t)sg * ny + y; }
	__device__ void next() {
		if (++sg == nseg) { sg = 0; if (++y == ny) { y = 0; zbase += (uint64_t)nseg * ny; } }
	}
};

// The records are stored [z][segment][y]; the scan runs over them in sweep order [z][y][segment]: a chunk of
// SCAN_CHUNK consecutive sweep positions is the same set of records whatever the order inside it only
// when it covers whole (y, all segments) groups - so the mapping is applied per element.
struct ScanArgs {  // per isovalue
	const uint32_t *seg_cnt;
	uint32_t tag;
	uint64_t *bsV, *bsT, *grV, *grT;
	SegBase *seg_base;
	Counters *ctr;
};
__global__ __launch_bounds__(256) void k_scan_reduce(const PerLane<ScanArgs> A, uint64_t n, Params P) {
	const ScanArgs &sa = A.a[blockIdx.y];
	const uint32_t *seg_cnt = sa.seg_cnt;
	const uint32_t tag = sa.tag;
	uint64_t *bsV = sa.bsV, *bsT = sa.bsT, *grV = sa.grV, *grT = sa.grT;
	__shared__ uint64_t sv[4], st[4];
	const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK;
	uint64_t v = 0, t = 0;
	// (the sum of a chunk does not depend on the order inside it: each thread takes 8 consecutive positions)
	const uint64_t q0 = base + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
	SweepWalk walk(P, q0);
	for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
		if (q0 + k < n) { const uint32_t c = seg_counts(seg_cnt[walk.store()], tag); v += c & 0xFFFu; t += c >> 12; }
		walk.next();
	}
	v = wave_sum(v); t = wave_sum(t);
	if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; st[threadIdx.x >> 6] = t; }
	__syncthreads();
	if (threadIdx.x == 0) {
		const uint64_t cv = sv[0] + sv[1] + sv[2] + sv[3], ct = st[0] + st[1] + st[2] + st[3];
		bsV[blockIdx.x] = cv; bsT[blockIdx.x] = ct;
		// ... and into the sum of the chunk's group of SCAN_GROUP chunks (cleared by k_slots): k_scan_apply then adds up the groups
		// before its own and the chunks of its own group before it - a few dozen values instead of up to 8 181 (2048 x 2048 x 1024)
		// (grV == nullptr: not worth its atomics - 4 us at 1024^3 - below SCAN_GROUPED_FROM chunks)
		if (grV && cv) atomicAdd((unsigned long long *)grV + blockIdx.x / SCAN_GROUP, (unsigned long long)cv);
		if (grV && ct) atomicAdd((unsigned long long *)grT + blockIdx.x / SCAN_GROUP, (unsigned long long)ct);
	}
}

// Second pass: every block first adds up what lies before its chunk - the sums of the groups of SCAN_GROUP chunks before its
// own group and of the chunks of its group before it (a few hundred values, resident in L2; cheaper than a separate
// one-block scan kernel between the two passes) - then scans its chunk.  The last block also knows the totals.
// (Until round 3 a block added up ALL chunk sums before its own: 2 045 chunks at 1024^3, 8 181 at 2048 x 2048 x 1024 -
// 128 KB per block there.  One pass with a decoupled look-back - chunk states {nothing / own sum / running sum}
// in one 64-bit word per sum, agent-scope atomics, chunks by ticket - was written and is correct and slower: 41 us against
// 10 + 9 at 1024^3, 116 against 76 at 2048 x 2048 x 1024: a state crosses from one XCD's L2 to another's through memory,
// and the chain of running sums is as long as the launch has rounds of blocks.)
__global__ __launch_bounds__(256) void k_scan_apply(const PerLane<ScanArgs> A, uint64_t n, Params P, uint64_t ghost_segs) {
	const ScanArgs &sa = A.a[blockIdx.y];
	const uint32_t *seg_cnt = sa.seg_cnt;
	const uint32_t tag = sa.tag;
	const uint64_t *bsV = sa.bsV, *bsT = sa.bsT, *grV = sa.grV, *grT = sa.grT;
	SegBase *seg_base = sa.seg_base;
	Counters *ctr = sa.ctr;
	__shared__ uint32_t sv[4], st[4];
	__shared__ uint64_t s_bv[4], s_bt[4];
	// (the chunk's own counts are asked for first: the sums before the chunk end in a block barrier, and no load crosses one)
	const uint64_t q0 = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
	uint32_t cv[SCAN_PER_THREAD], ct[SCAN_PER_THREAD], v = 0, t = 0, raw[SCAN_PER_THREAD];
	uint64_t st_idx[SCAN_PER_THREAD];
	SweepWalk walk(P, q0);
#pragma unroll
	for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
		st_idx[k] = (q0 + k < n) ? walk.store() : 0;
		walk.next();
		raw[k] = seg_cnt[st_idx[k]];  // (position 0 for the lanes beyond the end: a valid address, the value is dropped)
	}
	uint64_t bv = 0, bt = 0;  // vertices / triangles of all chunks before this one
	const uint32_t grp = grV ? blockIdx.x / SCAN_GROUP : 0u;
	for (uint32_t k = threadIdx.x; k < grp; k += 256u) { bv += grV[k]; bt += grT[k]; }
	for (uint32_t k = grp * SCAN_GROUP + threadIdx.x; k < blockIdx.x; k += 256u) { bv += bsV[k]; bt += bsT[k]; }
	bv = wave_sum(bv); bt = wave_sum(bt);
	if ((threadIdx.x & 63u) == 0) { s_bv[threadIdx.x >> 6] = bv; s_bt[threadIdx.x >> 6] = bt; }
	__syncthreads();
	bv = s_bv[0] + s_bv[1] + s_bv[2] + s_bv[3]; bt = s_bt[0] + s_bt[1] + s_bt[2] + s_bt[3];
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { ctr->totV = bv + bsV[blockIdx.x]; ctr->totT = bt + bsT[blockIdx.x]; }
#pragma unroll
	for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
		const uint32_t c = (q0 + k < n) ? seg_counts(raw[k], tag) : 0u;
		cv[k] = c & 0xFFFu; ct[k] = c >> 12;
		v += cv[k]; t += ct[k];
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) ctr->live_cursor = 0u;  // the last kernel of a tail leaves the cursor of k_slots' list zero for the next
	uint32_t iv = v, it = t;
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t a = __shfl_up(iv, d), b = __shfl_up(it, d);
		if ((int)lane >= d) { iv += a; it += b; }
	}
	if (lane == 63) { sv[wv] = iv; st[wv] = it; }
	__syncthreads();
	uint32_t ev = (uint32_t)bv + iv - v, et = (uint32_t)bt + it - t;
	for (uint32_t k = 0; k < wv; k++) { ev += sv[k]; et += st[k]; }
#pragma unroll
	for (uint32_t k = 0; k < SCAN_PER_THREAD; k++) {
		if (q0 + k < n) {
			if (cv[k] | ct[k]) seg_base[st_idx[k]] = SegBase{ev, et};  // (nobody asks for the base of a row segment that holds nothing)
			if (q0 + k == ghost_segs) { ctr->ghostV = ev; ctr->ghostT = et; }  // first segment of the emitted range
		}
		ev += cv[k]; et += ct[k];
	}
}

// ---------------------------------------------------------------------------------------------------
// emit: one thread per work record.  k_emit_fast_vertices / k_emit_fast_triangles handle the records the
// sweep finished itself, k_emit_slow the ones k_slow_plan planned (generic path: aliases, cells on the grid faces, ...)
// ---------------------------------------------------------------------------------------------------
struct EmitArgs {
	EmitCtx<sample_t> c;
	Counters *ctr;
	const uint32_t *slow_list;
	ListChunks lc;
	const uint2 *slot_base;
	uint32_t entry_cap;
	uint64_t capV, capT;
	uint64_t ghost_segs;  // row segments of the ghost slice (0 without ghost)
	uint32_t id_base;
	const unsigned long long *dev_base;  // (or nullptr) {id_base, first output vertex row, first output triangle row} in DEVICE memory: a z-slab whose
	                                     // place among the ranks came out of a collective and has not been to the host (mc33hip_emit_at_device_bases)
	const BatchDesc *batches;  // the records in batches of one slice slot (k_cells)
	uint32_t batch_cap;
	uint32_t stage_rows;  // every sample row of the grid starts on a 16-byte boundary: k_emit_vertices may stage rows in LDS
	Counters *host_ctr;   // pinned host copy of the counters: the triangle pass (the last kernel of an extraction) leaves them there
#ifdef MC33_DEV
	uint32_t *below_idx;  // [record][3] (developer experiment MC33_HIP_TRI_BELOW): positions of x in the three neighbouring row segments
#endif
};

// The fast emit passes take the records in storage order, which k_slots made (4 slices of a tile column, next
// row segment, next y tile, ...): neighbouring threads work on neighbouring cells, and consecutive slices of a
// column - which share two of their three sample planes - are handled close in time.  (Tried: one block per
// (slice, y tile) piece over all row segments, so that whole output cache lines come from one XCD - the write
// traffic fell from 2.7x to 1.4x of the algorithmic bytes, the time did not.)

// Work split of the fast emit passes: blocks are dealt to the 8 XCDs round robin (block b runs on XCD b % 8), each
// XCD has its own L2, and neighbouring records read neighbouring samples.  So every XCD gets ONE contiguous
// eighth of the record array, and the blocks that are resident on it together walk it side by side.
struct XcdWalk {
	uint32_t first, end, stride;
	__device__ XcdWalk(uint32_t n) {
		const uint32_t chunks = (n + 255u) / 256u, per_xcd = (chunks + 7u) / 8u;
		const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, blocks_per_xcd = (gridDim.x + 7u) >> 3;
		const uint32_t c0 = xcd * per_xcd;
		first = (c0 + slot) * 256u + threadIdx.x;
		end = min((c0 + per_xcd) * 256u, n);
		stride = blocks_per_xcd * 256u;
	}
};

// capacity / overflow check shared by both emit kernels; fills the slab offsets of the context
template <bool TOGETHER = false>
__device__ __forceinline__ bool emit_prepare(const EmitArgs &a, EmitCtx<sample_t> &c, const Counters &ctr) {
	const uint64_t gV = a.ghost_segs ? ctr.ghostV : 0, gT = a.ghost_segs ? ctr.ghostT : 0;
	// where this slab's part begins: from the launch arguments, or - a z-slab whose counts were exchanged on the device - from
	// three words a one-thread kernel made of the gathered table (wave-uniform: scalar loads, asked for with the counters)
	uint64_t idb = a.id_base, vo = 0, to = 0;
	if (a.dev_base) { idb = a.dev_base[0]; vo = a.dev_base[1]; to = a.dev_base[2]; }
	// TOGETHER: `|`, not `||` - every counter is asked for before the first is looked at; with short circuits the compiler fetches
	// them one comparison at a time, a scalar round trip each (the triangle pass, whose waves live for one record per lane)
	const bool over = TOGETHER ? (bool)((ctr.entry_cursor > a.entry_cap) | (vo + (ctr.totV - gV) > a.capV) | (to + (ctr.totT - gT) > a.capT) | (ctr.totV > 0xFFFFFFFFull) |
	                                    (ctr.totT > 0xFFFFFFFFull) | (idb + (ctr.totV - gV) > 0xFFFFFFFFull))
	                           : (ctr.entry_cursor > a.entry_cap || vo + (ctr.totV - gV) > a.capV || to + (ctr.totT - gT) > a.capT || ctr.totV > 0xFFFFFFFFull ||
	                              ctr.totT > 0xFFFFFFFFull || idb + (ctr.totV - gV) > 0xFFFFFFFFull);
	if (over) {
		if (blockIdx.x == 0 && threadIdx.x == 0) a.ctr->emit_skipped = 1;
		return false;
	}
	c.v_skip = (uint32_t)gV;
	c.t_skip = (uint32_t)gT;
	c.id_delta = (uint32_t)idb - (uint32_t)gV;
	if (a.dev_base) { c.V += 3ull * vo; c.N += 3ull * vo; c.Tri += 3ull * to; }
	return true;
}

#ifdef MC33_DEV
// vertices of the fast records (positions + normals), one thread per record with 12 loads of its own: the round-2 pass,
// kept in developer builds for A/B timing against k_emit_vertices (MC33_HIP_OLD_VERTEX_PASS=1)
__global__ __launch_bounds__(256) void k_emit_fast_vertices(const EmitArgs a) {
	__shared__ EntryB s_fast_b[256];
	s_fast_b[threadIdx.x] = a.c.fast_b[threadIdx.x];
	__syncthreads();
	const Counters ctr = *a.ctr;
	EmitCtx<sample_t> c = a.c;
	c.fast_b = s_fast_b; c.fast_b_in_lds = true;
	if (!emit_prepare(a, c, ctr)) return;
	const XcdWalk w(ctr.entry_cursor);
	for (uint32_t e = w.first; e < w.end; e += w.stride) {
		const uint32_t seg = c.entry_seg[e];  // (asked for together with the record, not after its flags are known)
		const EntryA ea = c.entries_a[e];
		asm volatile("" ::"v"(seg), "v"(ea.a0));  // (both wanted here: the compiler would move the segment's load behind the flag test)
		const Entry en = entry_join(ea, ctx_half_b(c, ea, e));
		if (!(en.w3 & ENTRY_SLOW)) emit_fast_vertices(c, en, seg);
	}
}
#endif

// ---------------------------------------------------------------------------------------------------
// k_emit_vertices: the vertices of the fast and tested records, one WAVE per batch of <= 64 records of one slice slot.
//
// The round-2 pass (k_emit_fast_vertices; developer builds keep it for A/B) was one thread per record with 12 short sample
// loads each; its 64 lanes sit in 64 different sample rows, so every load instruction is 64 cache-line look-ups in the
// CU's L1 - 65 cycles whatever its width, against 17 when four lanes share a row (tools/tcp_probe.hip,
// profiles/r03_tcp_probe.txt) - ~800 cycles per 64 records, although the 64 records of a batch share their rows: record
// (x, y, z) reads rows y, y+1, y+2 of planes z, z+1 and rows y, y+1 of plane z+2, and its neighbour one row up reads two
// of those three again.
//
// Per batch (all 64 records in one tile of one slice: same three planes, 63 cell rows):
//   1. lane = record: first / last record of every cell row of the batch (the records of a slot are sorted by row, then x);
//   2. lane = sample row r of the tile: the x interval the batch needs of row r - the records of cell rows r-2 .. r - as a
//      window of EV_W 16-byte chunks; a row that needs more is not staged;
//   3. lane = chunk: the chunks of all windows, the three planes of a chunk by the same lane, consecutive lanes on
//      consecutive chunks of a row - every load of the batch issued before the first is waited for - into an LDS image
//      [plane][row][chunk];
//   4. lane = VERTEX: a record makes one vertex on average; one lane per record with a branch per owned edge ran each
//      branch for a third of the lanes.  The vertices of the batch are listed by kind (edge 5, 6, 10), a lane takes one,
//      fetches its 10 samples from the image and runs vertex_on_edge's arithmetic in its order.  MC:990-1000, 1029-1039,
//      1175-1185 (the stencil), 485-585 (the stores);
//   5. records of rows that are not staged (long runs along x - where consecutive lanes read consecutive samples anyway -
//      or noise) and tested records with a centre vertex load for themselves as before (fast_samples_direct).
// The records and row bases of the next batch and the descriptor of the one after are in flight while a batch is worked on.
// What bounds it (profiles/r03_v1_c3_pmc.txt): 447 MB of HBM traffic for 94 MB of vertices, in 64-byte sectors scattered
// over three planes, at the 3.5 - 3.8 TB/s such traffic reaches on this part (tcp_probe: 64 lanes x 64 B per 720 cycles
// and CU); fewer instructions (1100 -> 830 per batch) changed nothing.  In launch order the same batches move 383 MB -
// see XcdBatchWalk.
// ---------------------------------------------------------------------------------------------------
// Window of a sample row in the LDS image: EV_W 16-byte chunks from the chunk that holds the first sample the batch needs of
// that row.  A row that needs more (a long run of records along x, or records far apart: noise) is not staged; the
// records that read it load for themselves - which is cheap exactly then, many lanes of a load sharing a row.
// (Measured per width, round 3: float 1024^3 2 / 3 / 4 chunks 176 / 119 / 143 us; ushort 2048 x 2048 x 1024 2 / 3 / 4 chunks
// 459 / 428 / 510 us per isovalue - three it is for both; a narrower window sends rows to the lanes' own loads, a wider one
// costs a block per CU.  And an image of fewer ROWS - 48 / 40 / 32 from the row of the batch's first record on, rows beyond
// it to the lanes' own loads; 4 blocks per CU instead of 3, three load groups instead of four: float 120 -> 117 / 119 / 124 us,
// ushort 422 -> 451 / 461 / 521 us per isovalue: more resident waves do not pay for the rows that fall out.)
#ifndef MC33_EV_W
#define MC33_EV_W (sizeof(sample_t) == 8 ? 4u : sizeof(sample_t) >= 2 ? 3u : 2u)
#endif
constexpr uint32_t EV_W = MC33_EV_W;
constexpr uint32_t EV_ROWS = 65;         // sample rows 0..64 of a tile (63 cell rows, y + 2 above the last)
constexpr uint32_t EV_EMPTY = 0xFFFFFFFFu;
// The image as a RING of planes (round 5; built, measured, NOT the form that runs: MC33_EV_RING=1 in developer builds).  Plane z
// lives in slot z % 3 and stays there while the wave goes on to the next batch - a wave takes a CONTIGUOUS piece of the batches
// (MC33_EV_PIECE; or runs of MC33_EV_RUN batches), i.e. the batches of a slice one after the other and then the next slice of the
// same tile column (slice_slot: the four slices of a group are adjacent), whose stencils (MC:990-1000, 1029-1039, 1175-1185) read
// two of the same three planes.  Every (slot, row) carries a tag - which chunks of the row it holds - and a batch loads only the
// windows its records need that are not there yet.  Bit-identical (148 GPU tests), and slower (profiles/r05_vertex_ring.txt):
// float 1024^3 123 -> 150 us, 322 -> 518 MB read; ushort 2048 x 2048 x 1024 485 -> 750 us per isovalue, 1.42 -> 1.71 GB.  What the
// strided walk shares between neighbouring waves at the same moment is whole 128-byte LINES in L2 (32 floats of a row: any shift
// of the surface from slice to slice stays inside), what the ring keeps is the 48-byte window one batch needed - the next slice's
// records, a cell or two further along x, miss it half of the time; and a wave that walks consecutive batches alone has no
// neighbour to share lines with (the reads grow with the run: 326 / 438 / 497 / 495 MB for runs of 1 / 4 / 8 / 16).  The tags and
// selects cost 14 % more vector instructions on top (46.2 M against 40.5 M per launch; 138 us with the ring's code on the strided
// walk, where it can reuse nothing).
#ifndef MC33_EV_RING
#define MC33_EV_RING 0
#endif
struct EmitVLds {                        // per wave
	uint32_t rowA[64], rowB[64];         // cell rows 0..62 of the tile: first / last record of the batch in that row, lane << 8 | x in the segment
	uint32_t rowvb[64];                  // cell rows: id of the first vertex of the row segment (seg_base)
	uint32_t rowinfo[EV_ROWS + 1];       // sample rows: staged << 31 | chunks - 1 << 16 | first chunk - chunk of the segment's first sample
	uint32_t vlist[256];                 // vertices of the batch: record (lane) | kind << 8, by kind
	uint32_t tag[3][EV_ROWS + 1];        // what slot s holds of sample row r: valid << 31 | chunks - 1 << 16 | first chunk (as rowinfo)
	uint4 data[3 * EV_ROWS * EV_W];      // [slot][row][chunk]
};

// Which wave takes which batch: each XCD gets one contiguous eighth of the batches (its own L2: see XcdWalk), and the waves
// resident on it walk that eighth side by side - wave k of the XCD takes batches k, k + W, k + 2 W, ... (W waves per XCD) -
// so that consecutive batches (consecutive slices of a tile column share two of their three sample planes; neighbouring
// row segments share the lines their vertices are written to) are in flight together.
// Measured alternatives (round 3, C3, HBM traffic of the pass / time): this walk 451 MB / 125 us - the waves drift apart
// over their ~20 batches; blocks of 4 batches in launch order, nothing prefetched: 383 MB (the vertex writes then cost
// exactly their bytes) but 157 us, every block paying its start-up chain; the same with 2-4 rounds per block 139 us;
// batches handed out in order by per-XCD atomic counters: 402 MB but 171 us (the compiler waits for every atomic on the
// spot).  MC33_EV_BLOCKED: every block a contiguous piece of its XCD's eighth instead.
#ifndef MC33_EV_RUN
#define MC33_EV_RUN 1  // batches a wave takes back to back before it strides on (developer A/B, round 4: consecutive batches are consecutive slices of a tile column)
#endif
struct XcdBatchWalk {
	uint32_t first, end, stride;
	// the t-th batch of this wave: runs of MC33_EV_RUN consecutive batches, the runs dealt to the waves of the XCD in turn
	__device__ uint32_t at(uint32_t t) const {
		constexpr uint32_t R = MC33_EV_RUN;
		if (R == 1) return first + t * stride;
		return base + ((t / R) * waves + wave) * R + (t % R);
	}
	uint32_t base, waves, wave;
	__device__ XcdBatchWalk(uint32_t n) {
		const uint32_t per_xcd = (n + 7u) / 8u;
		const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, blocks_per_xcd = (gridDim.x + 7u) >> 3;
		const uint32_t xend = min((xcd + 1u) * per_xcd, n);
#ifdef MC33_EV_BLOCKED
		const uint32_t piece = ((per_xcd + blocks_per_xcd - 1u) / blocks_per_xcd + 3u) & ~3u;  // batches per block
		first = xcd * per_xcd + slot * piece + (threadIdx.x >> 6);
		end = min(xcd * per_xcd + (slot + 1u) * piece, xend);
		stride = 4u;
#else
		first = xcd * per_xcd + slot * 4u + (threadIdx.x >> 6);
		end = xend;
		stride = blocks_per_xcd * 4u;
#endif
		base = xcd * per_xcd; waves = blocks_per_xcd * 4u; wave = slot * 4u + (threadIdx.x >> 6);
		if (MC33_EV_RUN > 1) first = at(0u);
#ifndef MC33_EV_PIECE
#define MC33_EV_PIECE MC33_EV_RING
#endif
#if MC33_EV_PIECE
		// every wave ONE contiguous piece of its XCD's eighth (the image in LDS is carried from batch to batch)
		const uint32_t piece = (per_xcd + waves - 1u) / waves;
		first = min(base + wave * piece, xend);
		end = min(first + piece, xend);
		stride = 1u;
#endif
	}
};

__device__ __forceinline__ uint32_t lane_below(uint32_t v) {  // lane r <- lane r-1 (lane 0 <- 63): DPP wave_shr:1
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {  // set bits of m in the lanes below this one
	return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// what a wave knows about its batch (wave-uniform: SGPRs)
struct BatchInfo { uint32_t first, count, sidx0, z, y0, xbase; };
__device__ __forceinline__ BatchInfo load_batch(const BatchDesc *batches, uint32_t j) {
	// read through the constant address space: the descriptors are not written while the emit passes run, and a uniform
	// address there is a SCALAR load (one request per wave, not 64 lanes asking for the same line)
	typedef const __attribute__((address_space(4))) uint32_t *cptr;
	cptr p = (cptr)(uintptr_t)(batches + __builtin_amdgcn_readfirstlane((int)j));
	return BatchInfo{p[0], p[1], p[2], p[3], p[4], p[5]};
}

// waves per SIMD the LDS image allows (4 blocks of 35 KiB with 2-chunk windows, 3 of 47 KiB with 3, 2 of 60 KiB with 4): the
// kernel may use the registers that leaves it, and no more
#ifndef MC33_EV_WAVES
#define MC33_EV_WAVES (MC33_SAMPLE_BYTES == 1 ? 4 : MC33_SAMPLE_BYTES == 8 ? 2 : 3)
#endif
template <int MODE>  // the vertex store (Params::store_mode): one kernel per store, see store_vertex
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MC33_EV_WAVES, MC33_EV_WAVES))) void k_emit_vertices(const EmitArgs a) {
	constexpr uint32_t SZ = (uint32_t)sizeof(sample_t);
	__shared__ EntryB s_fast_b[256];
	__shared__ EmitVLds s_w[4];
	s_fast_b[threadIdx.x] = a.c.fast_b[threadIdx.x];
	__syncthreads();
	const Counters ctr = *a.ctr;
	EmitCtx<sample_t> c = a.c;
	c.fast_b = s_fast_b; c.fast_b_in_lds = true;
	if (!emit_prepare(a, c, ctr)) return;
	const uint32_t lane = threadIdx.x & 63u;
	EmitVLds &L = s_w[threadIdx.x >> 6];
	const Params &P = c.P;
	const GridView<sample_t> &G = c.G;
	const uint64_t sliceB = G.slice * SZ;
	const uint32_t nbatch = min(ctr.batch_cursor, a.batch_cap);
	const XcdBatchWalk w(nbatch);
	if (w.first >= w.end) return;
#ifdef MC33_DEV  // a descriptor that cannot be right is reported (Counters::debug; the first one) and replaced by an empty one instead of followed
	auto load_batch = [&](const BatchDesc *b, uint32_t jj) -> BatchInfo {
		BatchInfo d = ::load_batch(b, jj);
		if (jj >= nbatch || d.count == 0u || d.count > 64u || (uint64_t)d.first + d.count > ctr.entry_cursor || d.z >= P.nz || d.y0 >= P.ny || d.xbase >= P.nx) {
			if (lane == 0 && atomicCAS(&a.ctr->debug[0], 0u, 1u) == 0u) {
				a.ctr->debug[1] = d.first; a.ctr->debug[2] = d.count; a.ctr->debug[3] = d.z; a.ctr->debug[4] = jj; a.ctr->debug[5] = w.first; a.ctr->debug[6] = w.end; a.ctr->debug[7] = nbatch;
			}
			d = BatchInfo{0u, 0u, 0u, c.z_emit, 0u, 0u};
		}
		return d;
	};
#endif
	// A wave walks its batches with the NEXT batch's records already asked for (and the descriptor after that): what is left
	// between two batches is the one round trip of the staging loads.  Loads and stores complete in issue order: whatever is
	// waited for after a batch's vertex stores have been issued waits for those stores as well, so everything the next batch
	// needs from memory is asked for AND waited for before the stores (finish_rec).
	struct Rec { EntryA a; EntryB b; uint32_t seg, rowvb; };
	auto load_rec = [&](const BatchInfo &d) -> Rec {
		Rec r;
		const uint32_t e = d.first + (lane < d.count ? lane : 0u);
		r.seg = c.entry_seg[e];
		r.a = c.entries_a[e];
		r.b = EntryB{0u, 0u};
		// first vertex of row segment (cell row `lane` of the tile): by row, not by record - it does not depend on the records
		const uint32_t rows = min(63u, P.ny - d.y0);
		r.rowvb = c.seg_base[d.sidx0 + min(lane, rows - 1u)].vbase;
		return r;
	};
	auto finish_rec = [&](Rec &r, const BatchInfo &d) {  // half B: from the sign index; a tested record's (rare: noisy fields) is a load of its own
		r.b = fast_half_b(c, (r.a.a0 >> 8) & 0xFFu);
		const bool stored = lane < d.count && (r.a.a0 & (ENTRYA_SLOW | ENTRYA_TESTED)) == ENTRYA_TESTED;
		if (__ballot(stored)) {  // wave-uniform
			if (stored) r.b = c.entries_b[d.first + lane];
			asm volatile("" ::"v"(r.b.b0), "v"(r.b.b1));  // (here, not at its first use behind the stores)
		}
		asm volatile("" ::"v"(r.seg), "v"(r.rowvb));
	};
	// ring of planes: the image holds the planes ring_z .. ring_z + 2 of tile column (ring_y0, ring_xbase) - those of the batch staged last (wave-uniform)
	uint32_t ring_z = 0u, ring_y0 = 0xFFFFFFFFu, ring_xbase = 0xFFFFFFFFu;
	uint32_t t = 0, j = w.at(0u);
	BatchInfo d0 = load_batch(a.batches, j), d1 = d0;
	if (w.at(1u) < w.end) d1 = load_batch(a.batches, w.at(1u));
	Rec rec0 = load_rec(d0), rec1 = rec0;
	finish_rec(rec0, d0);
	for (;;) {  // wave-uniform
		const bool more = w.at(t + 1u) < w.end, more2 = w.at(t + 2u) < w.end;
		BatchInfo d2 = d1;
		if (more2) d2 = load_batch(a.batches, w.at(t + 2u));
		auto next_batch = [&]() {  // (before the vertex stores: see above)
			if (more) finish_rec(rec1, d1);
		};
		const uint32_t z = d0.z, y0 = d0.y0, xbase = d0.xbase, count = d0.count, sidx0 = d0.sidx0;
		const bool on = lane < count;
		const EntryA ea = rec0.a;
		const uint32_t seg = rec0.seg;
		const Entry en = entry_join(ea, rec0.b);
		if (more) rec1 = load_rec(d1);
		const uint32_t r5 = (en.w2 >> 20) & 15u, r6 = (en.w2 >> 24) & 15u, r10 = (en.w3 >> 8) & 15u, r12 = entry_rank_centre(en);
		// (a ghost slice of a z-slab - z < z_emit - has its vertices written by the rank below)
		const bool creates = on && z >= c.z_emit && !(en.w3 & ENTRY_SLOW) && (r5 & r6 & r10 & r12) != 15u;
		if (__ballot(creates)) {
			const uint32_t rho = seg - sidx0, xl = en.w0 & 0xFFu;  // cell row in the tile (0..62), x in the segment
			const uint32_t x = xbase + xl, y = y0 + rho;
			const bool xin = x + 1 < P.nx, yin = y + 1 < P.ny, zin = z + 1 < P.nz;  // (zin: wave-uniform)
			const uint32_t cbase = (xbase * SZ) >> 4;
			L.rowvb[lane] = rec0.rowvb;
			bool staged_lane = false;
			uint32_t i0 = 0, i1 = 0, i2 = 0;
			if (a.stage_rows) {
				// ---- which records begin and end each cell row of the batch
				L.rowA[lane] = EV_EMPTY;
				L.rowB[lane] = EV_EMPTY;
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				{
					const uint32_t below = lane_below(rho), above = row_above(rho);
					if (on && (lane == 0u || below != rho)) L.rowA[rho] = lane << 8 | xl;
					if (on && (lane + 1u == count || above != rho)) L.rowB[rho] = lane << 8 | xl;
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				// ---- lane = sample row r of the tile: the chunks of it the batch needs = those of the records of cell rows r-2 .. r.
				// Lane r reads the first / last record of cell row r and gets rows r-1, r-2 from the lanes below (DPP); sample row 64
				// (above cell row 62 only) is lane 62's own interval, computed by every lane alike.
				const uint32_t fa = L.rowA[lane], fb = L.rowB[lane];
				const bool have = lane <= 62u && fa != EV_EMPTY;
				const uint32_t mn0 = have ? fa & 0xFFu : 255u, mx0 = have ? fb & 0xFFu : 0u;
				// (the shifts with every lane enabled, pinned by an empty asm: written as `lane >= 1 ? lane_below(..) : ..` the compiler
				// turned the select into a branch and ran the DPP move under it - lane 0 disabled, so lane 1 read nothing)
				uint32_t mn1 = lane_below(mn0), mx1 = lane_below(mx0);
				asm volatile("" : "+v"(mn1), "+v"(mx1));
				uint32_t mn2 = lane_below(mn1), mx2 = lane_below(mx1);
				asm volatile("" : "+v"(mn2), "+v"(mx2));
				mn1 = lane >= 1u ? mn1 : 255u; mx1 = lane >= 1u ? mx1 : 0u;
				mn2 = lane >= 2u ? mn2 : 255u; mx2 = lane >= 2u ? mx2 : 0u;
				auto window = [&](uint32_t r, uint32_t xmin, uint32_t xmax) -> uint32_t {  // (xmin > xmax: no record needs the row)
					const uint32_t lo = xbase + xmin, hi = min(xbase + xmax + 2u, P.nx);
					const uint32_t clo = (lo * SZ) >> 4, nm1 = ((hi * SZ + SZ - 1u) >> 4) - clo;  // first chunk, chunks - 1
					const bool staged = xmin <= xmax && y0 + r <= P.ny && nm1 < EV_W;
					return (staged ? 1u << 31 : 0u) | (nm1 & 3u) << 16 | ((clo - cbase) & 0xFFFFu);
				};
				L.rowinfo[lane] = window(lane, min(mn0, min(mn1, mn2)), max(mx0, max(mx1, mx2)));
#ifdef MC33_DEV  // the same interval straight from the table
				{
					uint32_t xmin = 255u, xmax = 0u;
					for (int k = 0; k < 3; k++) {
						const int ri = (int)lane - k;
						if (ri >= 0 && ri <= 62 && L.rowA[ri] != EV_EMPTY) { xmin = min(xmin, L.rowA[ri] & 0xFFu); xmax = max(xmax, L.rowB[ri] & 0xFFu); }
					}
					if (xmin != min(mn0, min(mn1, mn2)) || xmax != max(mx0, max(mx1, mx2)))
						if (atomicCAS(&a.ctr->debug[0], 0u, 2u) == 0u) { a.ctr->debug[1] = lane; a.ctr->debug[2] = xmin; a.ctr->debug[3] = xmax; a.ctr->debug[4] = min(mn0, min(mn1, mn2)); a.ctr->debug[5] = max(mx0, max(mx1, mx2)); a.ctr->debug[6] = mn1; a.ctr->debug[7] = mn2; }
				}
#endif
				{
					const uint32_t info64 = window(64u, (uint32_t)__builtin_amdgcn_readlane((int)mn0, 62), (uint32_t)__builtin_amdgcn_readlane((int)mx0, 62));
					if (lane == 0) L.rowinfo[64] = info64;
				}
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				i0 = L.rowinfo[rho]; i1 = L.rowinfo[rho + 1u]; i2 = L.rowinfo[yin ? rho + 2u : rho];
				staged_lane = creates && ((i0 & i1 & i2) >> 31) != 0u;
				// (a batch of long runs along x - many records per row, every window too narrow - stages nothing: its records
				// load for themselves, consecutive lanes reading consecutive samples of a row)
				if (__ballot(staged_lane)) {
				// ---- lane = chunk: item i of a plane is chunk i % EV_W of row i / EV_W; the three planes of a chunk by the same
				// lane, all loads of the batch issued before the first one is waited for (named registers: an array indexed by
				// the group ended up in scratch memory)
				const char *plane0 = (const char *)G.p + ((uint64_t)(z - G.z0) * G.slice + (uint64_t)y0 * G.pitch) * SZ + (uint64_t)cbase * 16u;
				const uint32_t pitchB = G.pitch * SZ;
				constexpr uint32_t NITEM = EV_ROWS * EV_W, NGRP = (NITEM + 63u) / 64u;
				static_assert(NGRP <= 5, "groups of the staging loads");
				// ring: plane z + k of this slice lives in slot (z + k) % 3.  A slot that holds another plane, or a plane of another
				// tile column, is empty: its tags are cleared (wave-uniform decisions) before anybody looks at them.
				const uint32_t s0 = MC33_EV_RING ? z % 3u : 0u, s1 = s0 == 2u ? 0u : s0 + 1u, s2 = s1 == 2u ? 0u : s1 + 1u;
				if (MC33_EV_RING) {
					const bool other = ring_y0 != y0 || ring_xbase != xbase;
					const uint32_t sk[3] = {s0, s1, s2};
#pragma unroll
					for (uint32_t k = 0; k < 3u; k++)
						if (other || z + k - ring_z > 2u) {  // (unsigned: also a plane below the ones held)
							L.tag[sk[k]][lane] = 0u;
							if (lane < EV_ROWS + 1u - 64u) L.tag[sk[k]][64u + lane] = 0u;
						}
					ring_z = z; ring_y0 = y0; ring_xbase = xbase;
					__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				}
				// (the row words of all groups first: one LDS wait, not one per group)
				uint32_t info[NGRP];
				uint32_t have0[NGRP], have1[NGRP], have2[NGRP];
#pragma unroll
				for (uint32_t g = 0; g < NGRP; g++) {
					const uint32_t rr = min((g * 64u + lane) / EV_W, EV_ROWS - 1u);
					info[g] = L.rowinfo[rr];
					if (MC33_EV_RING) { have0[g] = L.tag[s0][rr]; have1[g] = L.tag[s1][rr]; have2[g] = L.tag[s2][rr]; }
				}
				// does the window a slot holds of a row (tag) cover the one the batch needs (rowinfo)?
				auto covers = [](uint32_t have, uint32_t want) -> bool {
					const uint32_t hl = have & 0xFFFFu, wl = want & 0xFFFFu;
					return (have >> 31) && hl <= wl && wl + ((want >> 16) & 3u) <= hl + ((have >> 16) & 3u);
				};
				const uint4 zero4 = {0u, 0u, 0u, 0u};
				uint4 qa0 = zero4, qa1 = zero4, qa2 = zero4, qb0 = zero4, qb1 = zero4, qb2 = zero4, qc0 = zero4, qc1 = zero4, qc2 = zero4;
				uint4 qd0 = zero4, qd1 = zero4, qd2 = zero4, qe0 = zero4, qe1 = zero4, qe2 = zero4;
				// (returns bit k: plane z + k of the item was loaded)
				auto fetch = [&](uint32_t g, uint4 &q0, uint4 &q1, uint4 &q2) -> uint32_t {
					const uint32_t it = g * 64u + lane;
					const uint32_t r = it / EV_W, ck = it - r * EV_W;
					const bool need = it < NITEM && (info[g] >> 31) && ck <= ((info[g] >> 16) & 3u);
					// (unconditional loads from a safe address for the other lanes were tried: 4 % slower on ushort grids; so were loads
					// through a buffer descriptor with those lanes aimed past its end - the hardware answers zeros, no branch, no
					// registers to clear, 54 vector instructions less in the kernel - 112 -> 115 us at C3, 501 -> 525 at C5: a lane that is
					// switched off costs the memory pipeline nothing, a lane that is refused does)
					uint32_t got = 0u;
					if (need) {
						const char *addr = plane0 + (uint64_t)r * pitchB + (uint64_t)((info[g] & 0xFFFFu) + ck) * 16u;
						if (!MC33_EV_RING || !covers(have0[g], info[g])) { q0 = *(const uint4 *)addr; got |= 1u; }
						if (!MC33_EV_RING || !covers(have1[g], info[g])) { q1 = *(const uint4 *)(addr + sliceB); got |= 2u; }
						if (zin && (!MC33_EV_RING || !covers(have2[g], info[g]))) { q2 = *(const uint4 *)(addr + 2u * sliceB); got |= 4u; }
					}
					return got;
				};
				// (the chunk goes to its slot, and the lane of a row's first chunk notes what the slot now holds of the row)
				auto put = [&](uint32_t g, uint32_t got, const uint4 &q0, const uint4 &q1, const uint4 &q2) {
					const uint32_t it = g * 64u + lane;
					const uint32_t r = it / EV_W, ck = it - r * EV_W;
					if (got & 1u) { L.data[s0 * NITEM + it] = q0; if (MC33_EV_RING && ck == 0u) L.tag[s0][r] = info[g]; }
					if (got & 2u) { L.data[s1 * NITEM + it] = q1; if (MC33_EV_RING && ck == 0u) L.tag[s1][r] = info[g]; }
					if (got & 4u) { L.data[s2 * NITEM + it] = q2; if (MC33_EV_RING && ck == 0u) L.tag[s2][r] = info[g]; }
				};
				const uint32_t na = fetch(0u, qa0, qa1, qa2), nb = fetch(1u, qb0, qb1, qb2), nc = fetch(2u, qc0, qc1, qc2);
				const uint32_t nd = NGRP > 3 ? fetch(3u, qd0, qd1, qd2) : 0u, ne = NGRP > 4 ? fetch(4u, qe0, qe1, qe2) : 0u;
				put(0u, na, qa0, qa1, qa2); put(1u, nb, qb0, qb1, qb2); put(2u, nc, qc0, qc1, qc2);
				if (NGRP > 3) put(3u, nd, qd0, qd1, qd2);
				if (NGRP > 4) put(4u, ne, qe0, qe1, qe2);
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				}
			} else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			const uint32_t vbase = L.rowvb[rho] + (en.w1 & 0xFFFFu);
			next_batch();  // (before the stores below)
			// ---- the vertices.  Staged records: one LANE PER VERTEX (a record makes one on average; one branch per owned edge
			// would run three times for a third of the lanes each).  The vertex of the edge from corner A to corner 6 = B
			// along axis k: t = vA / (vA - vB); along the edge the gradient is vB - vA, across it (axes u1, u2) central
			// differences at both ends blended by t, or one-sided ones on the far faces of the grid - vertex_on_edge's
			// arithmetic in its order (MC:990-1000 edge 5, 1029-1039 edge 6, 1175-1185 edge 10), whatever the axis.
			const bool direct = creates && (!staged_lane || r12 != 15u);  // (a centre vertex needs all 8 corners: rare, the record's own loads)
			const bool viaimg = creates && !direct;
			const uint64_t m5 = __ballot(viaimg && r5 != 15u), m6 = __ballot(viaimg && r6 != 15u), m10 = __ballot(viaimg && r10 != 15u);
			const uint32_t n5 = (uint32_t)__popcll(m5), n6 = (uint32_t)__popcll(m6), nv = n5 + n6 + (uint32_t)__popcll(m10);
			if (nv) {
				if (viaimg && r5 != 15u) L.vlist[lanes_below(m5)] = lane;
				if (viaimg && r6 != 15u) L.vlist[n5 + lanes_below(m6)] = lane | 1u << 8;
				if (viaimg && r10 != 15u) L.vlist[n5 + n6 + lanes_below(m10)] = lane | 2u << 8;
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				// what a vertex lane needs of its record
				const uint32_t rw0 = xl | rho << 8 | r5 << 16 | r6 << 20 | r10 << 24 | (xin ? 1u << 28 : 0u) | (yin ? 1u << 29 : 0u);
				const uint32_t rw1 = (i0 & 0xFFu) | (i1 & 0xFFu) << 8 | (i2 & 0xFFu) << 16;  // first chunk of rows rho, rho + 1, rho + 2 | rho
				const char *img = (const char *)L.data;
#ifdef MC33_EV_ONE_ITER  // (developer timing experiment, results wrong: what would the pass take if no batch had more than 64 vertices?)
				for (uint32_t v0 = 0; v0 < min(nv, 64u); v0 += 64u) {
#else
				for (uint32_t v0 = 0; v0 < nv; v0 += 64u) {  // wave-uniform
#endif
					const bool act = v0 + lane < nv;
					const uint32_t ent = L.vlist[act ? v0 + lane : 0u];
					const uint32_t src = (ent & 63u) << 2, kind = ent >> 8;
					const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)rw0), w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)rw1);
					const uint32_t vb = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)vbase);
					if (act) {
						const uint32_t vxl = w0 & 0xFFu, vrho = (w0 >> 8) & 0xFFu;
						const bool vxin = (w0 >> 28) & 1u, vyin = (w0 >> 29) & 1u;
						const uint32_t rank = kind == 0u ? (w0 >> 16) & 15u : kind == 1u ? (w0 >> 20) & 15u : (w0 >> 24) & 15u;
						// sample (dx, dy, dz) of the cell: byte in the image
						const uint32_t xb0 = vxl * SZ + ((xbase * SZ) & 15u);
						// (the byte of sample x in the three rows of the cell, once; a plane is EV_ROWS * EV_W chunks on)
#if MC33_EV_RING
						// ring: plane z + dz sits in slot (z + dz) % 3, and what a slot holds of a row begins at the chunk its TAG names
						// (the window may have been staged for an earlier batch with other records)
						const uint32_t zs0 = z % 3u, zs1 = zs0 == 2u ? 0u : zs0 + 1u, zs2 = zs1 == 2u ? 0u : zs1 + 1u;
						const uint32_t vr2 = vyin ? vrho + 2u : vrho;
						auto rowbyte = [&](uint32_t sl, uint32_t rw) -> uint32_t {  // byte of the cell's sample x in slot sl, row rw
							return sl * (EV_ROWS * EV_W * 16u) + (rw * EV_W - (L.tag[sl][rw] & 0xFFFFu)) * 16u + xb0;
						};
						// (nine named values and selects: an array indexed by dy / dz - some are per-lane values - lived in scratch memory)
						const uint32_t b00 = rowbyte(zs0, vrho), b01 = rowbyte(zs0, vrho + 1u), b02 = rowbyte(zs0, vr2);
						const uint32_t b10 = rowbyte(zs1, vrho), b11 = rowbyte(zs1, vrho + 1u), b12 = rowbyte(zs1, vr2);
						const uint32_t b20 = rowbyte(zs2, vrho), b21 = rowbyte(zs2, vrho + 1u), b22 = rowbyte(zs2, vr2);
						(void)w1;
						auto smp = [&](uint32_t dx, uint32_t dy, uint32_t dz) -> sample_t {
							const uint32_t p0 = dy == 0u ? b00 : dy == 1u ? b01 : b02, p1 = dy == 0u ? b10 : dy == 1u ? b11 : b12, p2 = dy == 0u ? b20 : dy == 1u ? b21 : b22;
							return *(const sample_t *)(img + (dz == 0u ? p0 : dz == 1u ? p1 : p2) + dx * SZ);
						};
#else
						const uint32_t rb0 = (vrho * EV_W - (w1 & 0xFFu)) * 16u + xb0, rb1 = ((vrho + 1u) * EV_W - ((w1 >> 8) & 0xFFu)) * 16u + xb0,
						               rb2 = ((vrho + 2u) * EV_W - ((w1 >> 16) & 0xFFu)) * 16u + xb0;
						auto smp = [&](uint32_t dx, uint32_t dy, uint32_t dz) -> sample_t {
							const uint32_t rb = dy == 0u ? rb0 : dy == 1u ? rb1 : rb2;
							return *(const sample_t *)(img + rb + dz * (EV_ROWS * EV_W * 16u) + dx * SZ);
						};
#endif
						// corner A of the edge: (1,1,0) edge 5 | (1,0,1) edge 6 | (0,1,1) edge 10; B = (1,1,1)
						const uint32_t ax = kind != 2u, ay = kind != 1u, az = kind != 0u;
						const real_t iso = P.iso;
						const sample_t fA = smp(ax, ay, az), fB = smp(1u, 1u, 1u);
						// across the edge: u1 = x (edges 5, 6) or y (edge 10); u2 = y (edge 5) or z (edges 6, 10).  Where the outer
						// neighbour does not exist (far faces) the inner one is read twice and the one-sided form is taken.
						const bool in1 = kind != 2u ? vxin : vyin, in2 = kind == 0u ? vyin : zin;
						const uint32_t x2 = vxin ? 2u : 1u, y2 = vyin ? 2u : 0u, z2 = zin ? 2u : 0u;
						sample_t a1m, a1p, b1m, b1p, a2m, a2p, b2m, b2p;
						if (kind != 2u) { a1m = smp(0u, ay, az); a1p = smp(x2, ay, az); b1m = smp(0u, 1u, 1u); b1p = smp(x2, 1u, 1u); }
						else { a1m = smp(0u, 0u, 1u); a1p = smp(0u, y2, 1u); b1m = smp(1u, 0u, 1u); b1p = smp(1u, y2, 1u); }
						if (kind == 0u) { a2m = smp(1u, 0u, 0u); a2p = smp(1u, y2, 0u); b2m = smp(1u, 0u, 1u); b2p = smp(1u, y2, 1u); }
						else { a2m = smp(ax, ay, 0u); a2p = smp(ax, ay, z2); b2m = smp(1u, 1u, 0u); b2p = smp(1u, 1u, z2); }
						const real_t va = iso - (real_t)fA, vbv = iso - (real_t)fB;
						const real_t t = va / (va - vbv);
						const real_t g0 = vbv - va;
						const real_t g1 = in1 ? 0.5f * (sample_diff(a1m, a1p) * (1 - t) + sample_diff(b1m, b1p) * t)
						                      : (va - (iso - (real_t)a1m)) * (1 - t) + (vbv - (iso - (real_t)b1m)) * t;
						const real_t g2 = in2 ? 0.5f * (sample_diff(a2m, a2p) * (1 - t) + sample_diff(b2m, b2p) * t)
						                      : (va - (iso - (real_t)a2m)) * (1 - t) + (vbv - (iso - (real_t)b2m)) * t;
						const uint32_t vx = xbase + vxl, vy = y0 + vrho;
						real_t r[6];
						r[0] = kind == 2u ? (real_t)vx + t : (real_t)(vx + 1u);
						r[1] = kind == 1u ? (real_t)vy + t : (real_t)(vy + 1u);
						r[2] = kind == 0u ? (real_t)z + t : (real_t)(z + 1u);
						r[3] = kind == 2u ? g0 : g1;
						r[4] = kind == 0u ? g2 : kind == 1u ? g0 : g1;
						r[5] = kind == 0u ? g0 : g2;
						store_vertex<MODE>(P, r, c.V, c.N, vb + rank - c.v_skip);
					}
				}
			}
			if (__ballot(direct)) {  // (wave-uniform: records of rows that are not staged, tested records with a centre vertex)
				if (direct) {
					FastSamples<sample_t> S;
					fast_samples_direct(G, x, y, z, xin, yin, zin, S);
					fast_vertices_compute<sample_t, MODE>(c, x, y, z, vbase, r5, r6, r10, r12, S);
				}
			}
		} else next_batch();
		if (!more) break;
		t++;
		j = w.at(t);
		d0 = d1; d1 = d2; rec0 = rec1;
	}
}

// triangles of the fast records (ids of shared edges through the owners' records).
// (Round 3 tried the wave-per-batch form of k_emit_vertices here too: row bases by row, the batch's owner records - two short
// runs of the record array, bounded by a wave minimum / maximum - staged in LDS by coalesced loads, 10 instead of 22 load
// instructions per 64 records.  Bit-identical, and slower: 568 against 395 us per isovalue at C5.  A batch is a chain of
// dependent steps - directory words, run bounds, staging, LDS, ids - and 16 waves per CU do not hide it; one thread per
// record at 32 waves per CU does.  Dropped.)
#ifdef MC33_DEV
template <int BELOW = 0>  // developer experiment: 1 = keep the owner positions found through the directory, 2 = take them from that array instead
#else
[[maybe_unused]] constexpr int BELOW = 0;
#endif
__global__ __launch_bounds__(256) void k_emit_fast_triangles(const EmitArgs a) {
	__shared__ uint32_t s_id[13][256];
	__shared__ EntryB s_fast_b[256];
	// A wave of this kernel lives for one record per lane and ~8 dependent round trips; what it does before the first of the
	// record's own counts in full.  So: the table's words, the counters and - as soon as the counters say which record - the
	// record itself are all asked for before anything is waited for, and the table goes into LDS behind that.  (Before round 4:
	// table load, wait, LDS, barrier, counters, wait, more counters, wait, record: three round trips ahead of the first.)
	const EntryB fb = a.c.fast_b[threadIdx.x];
	const Counters ctr = *a.ctr;
	EmitCtx<sample_t> c = a.c;
	c.fast_b = s_fast_b; c.fast_b_in_lds = true;
	const bool ok = emit_prepare<true>(a, c, ctr);
	const XcdWalk w(ctr.entry_cursor);
	uint32_t e = w.first;
	const uint32_t e0 = ok && e < w.end ? e : 0u;  // (records 0 and 1 exist in every allocation)
	uint32_t seg = c.entry_seg[e0];
	EntryA2 pair = entry_pair(c.entries_a + (e0 ? e0 - 1u : 0u));  // the record and the one before it (the owner of two of its edges, mostly)
	uint32_t kb[3] = {0u, 0u, 0u};
#ifdef MC33_DEV
	if (BELOW == 2) { kb[0] = a.below_idx[3ull * e0]; kb[1] = a.below_idx[3ull * e0 + 1u]; kb[2] = a.below_idx[3ull * e0 + 2u]; }  // (with the record)
#endif
	asm volatile("" ::"v"(seg), "v"(pair.lo.a0), "v"(pair.hi.a0), "v"(kb[0]), "v"(kb[1]), "v"(kb[2]));
	s_fast_b[threadIdx.x] = fb;
	__syncthreads();
	// The counters of the extraction for the host, straight into its pinned copy (everything before this kernel on the stream
	// has finished: they are final, and every emit kernel decides `emit_skipped` alike): the call's one synchronisation then
	// finds them there, without a device-to-host copy command of 100 bytes behind the last kernel.
	if (a.host_ctr && blockIdx.x == 0 && threadIdx.x < sizeof(Counters) / 4) {
		// (a lane per word from memory to memory: a private copy of the struct put scratch memory into the kernel - every wave's
		// launch pays for that - as soon as the struct grew by two words: 90 -> 122 us at 1024^3, round 3; one lane copying word
		// after word was a chain of 24 load / store round trips to host memory in the first wave of the grid)
		static_assert(sizeof(Counters) % 4 == 0 && sizeof(Counters) / 4 <= 64, "Counters in words, a lane each");
		static_assert(offsetof(Counters, emit_skipped) % 4 == 0, "emit_skipped is a word");
		const volatile uint32_t *src = (const volatile uint32_t *)a.ctr;
		volatile uint32_t *dst = (volatile uint32_t *)a.host_ctr;
		const uint32_t k = threadIdx.x;
		dst[k] = k == offsetof(Counters, emit_skipped) / 4 ? (ok ? 0u : 1u) : src[k];
	}
	if (!ok) return;
	const URef ids{&s_id[0][threadIdx.x], 256};
	while (e < w.end) {
		const EntryA ea = e ? pair.hi : pair.lo;
		const Entry en = entry_join(ea, ctx_half_b(c, ea, e));
#ifdef MC33_DEV
		if (BELOW == 1) { uint32_t kept[3] = {e, e, e}; if (!(en.w3 & ENTRY_SLOW)) emit_fast_triangles(c, en, pair.lo, seg, e, ids, nullptr, kept); for (int g = 0; g < 3; g++) a.below_idx[3ull * e + g] = kept[g]; }
		else if (BELOW == 2) { if (!(en.w3 & ENTRY_SLOW)) emit_fast_triangles(c, en, pair.lo, seg, e, ids, kb); }
		else
#endif
		if (!(en.w3 & ENTRY_SLOW)) emit_fast_triangles(c, en, pair.lo, seg, e, ids);
		e += w.stride;
		if (e >= w.end) break;
		seg = c.entry_seg[e];
		pair = entry_pair(c.entries_a + e - 1u);
#ifdef MC33_DEV
		if (BELOW == 2) { kb[0] = a.below_idx[3ull * e]; kb[1] = a.below_idx[3ull * e + 1u]; kb[2] = a.below_idx[3ull * e + 2u]; }
#endif
		asm volatile("" ::"v"(seg), "v"(pair.lo.a0), "v"(pair.hi.a0));
	}
}

// k_emit_slow: the records the generic per-cell code writes (cells on the grid's 0-faces, corners equal to the isovalue, aliases), one
// thread per record walking its up to 13 pattern slots one after the other (emit_cell).  The form for MANY slow records (noise,
// integer isovalues on integer grids: 2 M of them at 1024^3 in 0.75 ms); with few the call waits for the length of one thread's chain
// of 20 - 30 dependent round trips - see k_emit_slow_slots.
__global__ __launch_bounds__(256) void k_emit_slow(const EmitArgs a) {
	__shared__ real_t s_v[8][256];
	__shared__ real_t s_w[8][256];
	__shared__ uint32_t s_id[13][256];
	const Counters ctr = *a.ctr;
	EmitCtx<sample_t> c = a.c;
	if (!emit_prepare(a, c, ctr)) return;
	const VRef v{&s_v[0][threadIdx.x], 256}, w{&s_w[0][threadIdx.x], 256};
	const URef ids{&s_id[0][threadIdx.x], 256};
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256];
	if (blockIdx.x * 256u >= ctr.slow_cursor) return;  // (k_slow_plan left the total there: most blocks of most calls)
	ChunkMap cm;
	cm.build(s_pre, s_red, a.lc.slow_cnt, a.lc.n);
	const uint32_t n = cm.total;
	for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
		const uint32_t gq = cm.group_of(t);
		emit_cell(c, a.slow_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])], v, w, ids);
	}
}

// ---------------------------------------------------------------------------------------------------
// k_emit_slow_slots: the same records with SIXTEEN LANES PER RECORD, a lane per pattern slot (edges 0..11 and the centre; lanes
// 13..15 only take triangles) - the form for FEW slow records (round 4).
//
// One thread per record (k_emit_slow) goes, for each of up to 13 slots, either through the vertex it creates (its gradient samples: a
// round trip) or through the chase to the record that did (directory word, record, its stored plan, next hop: two or three round
// trips per hop), one slot after the other - a chain of 20 - 30 dependent round trips, 31 us for the 8 800 such cells of the 1024^3
// cos field with the GPU to itself.  It used to be hidden beside the fast passes on a second stream, at the price of an event at the
// fork, a cross-queue wait at the join (6 - 7 us each inside an emit stage of 200) and of slow blocks still resident when the vertex
// pass placed its own (see enqueue_emit).  With a lane per slot the chain is as long as ONE slot's: record, plan, the slot's vertex or
// chase, ids through LDS, a lane per triangle - 14 us for those 8 800 records, 32 us for 34 000 (2048 x 2048 x 1024 ushort), in
// sequence behind the fast passes.  Every lane repeats the record's set-up, though: 2 M records take 2.3 ms this way against 0.75 ms
// with a thread each - the host picks by the last count (enqueue_emit).  Same functions, same stores per vertex and per triangle:
// emit_cell (mc33_cell.h, what the host emulator runs) is the statement of what this computes.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit_slow_slots(const EmitArgs a) {
	__shared__ real_t s_v[8][16];      // corner values of the block's 16 records (the 16 lanes of a record write the same eight)
	__shared__ real_t s_w[8][256];     // per lane: the corners of an owner cell on a chase (iso = -0.0 only)
	__shared__ uint32_t s_id[16][16];  // [record][slot]: vertex ids
	__shared__ uint32_t s_pre[LIST_CHUNKS + 1], s_red[256];
	const Counters ctr = *a.ctr;
	EmitCtx<sample_t> c = a.c;
	if (!emit_prepare(a, c, ctr)) return;
	if (blockIdx.x * 16u >= ctr.slow_cursor) return;  // (k_slow_plan left the total there)
	const uint32_t sub = threadIdx.x & 15u, cell = threadIdx.x >> 4, lane = threadIdx.x & 63u, gsh = lane & 48u;
	const VRef v{&s_v[0][cell], 16}, w{&s_w[0][threadIdx.x], 256};
	uint32_t *ids = s_id[cell];
	ChunkMap cm;
	cm.build(s_pre, s_red, a.lc.slow_cnt, a.lc.n);
	const uint32_t n = cm.total;
	if (n > gridDim.x * 32u) {
		// Far more records than the launch was sized for: the host went by the count of an EARLIER extraction (mc33hip_extract_into does
		// not stop to read this one's), and the isovalue has moved onto the samples since.  A thread per record then, as k_emit_slow.
		__shared__ real_t s_v1[8][256];
		__shared__ uint32_t s_id1[13][256];
		const VRef v1{&s_v1[0][threadIdx.x], 256};
		const URef ids1{&s_id1[0][threadIdx.x], 256};
		for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n; t += gridDim.x * 256u) {
			const uint32_t gq = cm.group_of(t);
			emit_cell(c, a.slow_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])], v1, w, ids1);
		}
		return;
	}
	for (uint32_t t0 = blockIdx.x * 16u; t0 < n; t0 += gridDim.x * 16u) {  // (block-uniform)
		const uint32_t t = t0 + cell;
		bool live = t < n;
		uint32_t entry_index = 0, x = 0, y = 0, z = 0, vbase = 0, tpos = 0;
		Entry en{};
		if (live) {
			const uint32_t gq = cm.group_of(t);
			entry_index = a.slow_list[a.slot_base[(uint64_t)gq << a.lc.shift].x + (t - cm.pre[gq])];
			const uint32_t s = c.entry_seg[entry_index];
			en = ctx_entry(c, entry_index);
			const SegCoord sc = segment_coord(c.P, s);
			y = sc.y; z = sc.z; x = sc.xbase + (en.w0 & 0xFFu);
			// (a tested cell that k_slow_plan found on the slow list: the fast emit passes write it; a ghost slice: the rank below does)
			live = (en.w3 & ENTRY_SLOW) && z >= c.z_emit;
			if (live) {
				const SegBase sb = c.seg_base[s];
				vbase = sb.vbase + (en.w1 & 0xFFFFu);
				tpos = sb.tbase + (en.w1 >> 16) - c.t_skip;
			}
		}
		CellPlan p{};
		uint32_t id = NO_ID;
		if (live) {
			load_cell(c.G, c.P.iso, x, y, z, v);
			plan_restore(p, c.tab.lut, en, c.entries_c[entry_index]);
			plan_restore_points(p, v);
			const uint32_t e = sub;
			if (e < 13u && (p.visited & (1u << e))) {  // the slot's id; a NEW vertex is written on the way (emit_cell's loop body)
				const uint32_t r = plan_rank(p, e);
				if (r != 15u) {
					id = vbase + r;
					if (p.created & (1u << e)) {
						real_t g[6];
						if (e == 12u) vertex_centre(x, y, z, v, g);
						else if (p.onpoint & (1u << e)) {
							const uint32_t cc = corner_code((p.onb & (1u << e)) ? edge_b(e) : edge_a(e));
							vertex_on_point(c.P, c.G, x + (cc & 1), y + ((cc >> 1) & 1), z + (cc >> 2), g);
						} else
							vertex_on_edge(c.P, c.G, x, y, z, e, v, g);
						store_vertex(c.P, g, c.V, c.N, vbase + r - c.v_skip);
					}
				} else {
					const RootRef root = chase_root(c, tgt_edge(plan_tgt(p, e), x, y, z), w);
					id = root.rec == NO_ID ? NO_ID : c.seg_base[root.seg].vbase + root.voff + root.rank;
				}
			}
		}
		ids[sub] = id;
		// the triangles: lane k of the record takes the k-th of its pattern (at most 12; the last one has no continuation bits)
		uint32_t word = 0xF000u;
		if (live) word = c.tab.lut[min((uint32_t)p.poff + 1u + sub, (uint32_t)MC33_LUT_COUNT - 1u)];
		const uint32_t ends = (uint32_t)(__ballot(live && !(word >> 12)) >> gsh) & 0xFFFFu;  // (every lane of the wave gets here)
		const bool mine = live && ends && sub <= (uint32_t)__builtin_ctz(ends | 0x10000u);
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the ids of the 16 lanes of a record: one wave)
		uint32_t ti[3] = {0u, 0u, 0u};
		bool keep = false;
		if (mine) {  // MC:780-784, 1235-1250
			const uint32_t e2 = word & 15u, e1 = (word >> 4) & 15u, e0 = (word >> 8) & 15u;
			ti[2] = ids[e2]; ti[1] = ids[e1]; ti[0] = ids[e0];
			// MC:1235 on the ids - except for iso = -0.0, where ids may be "no vertex" (see chase_root): the triangle slots were counted by
			// vertex identity (count_triangles_stored), and the same test decides here
			RootMemo memo{nullptr, 0, 0u};
			keep = c.P.negzero_iso ? (slots_differ_stored(c, p, x, y, z, e2, e1, w, memo) && slots_differ_stored(c, p, x, y, z, e2, e0, w, memo) &&
			                          slots_differ_stored(c, p, x, y, z, e1, e0, w, memo))
			                       : (ti[0] != ti[1] && ti[0] != ti[2] && ti[1] != ti[2]);
		}
		const uint32_t kept = (uint32_t)(__ballot(keep) >> gsh) & 0xFFFFu;
		if (keep) {
			uint32_t *tr = c.Tri + 3 * (uint64_t)(tpos + (uint32_t)__popc(kept & ((1u << sub) - 1u)));
			const bool swap = (p.n != 0) != (c.P.normal_neg != 0);  // MC:1246-1250
			tr[0] = (swap ? ti[1] : ti[0]) + c.id_delta; tr[1] = (swap ? ti[0] : ti[1]) + c.id_delta; tr[2] = ti[2] + c.id_delta;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (before the next round's ids and corner values)
	}
}

// ===================================================================================================
// Host side: context, uploads, launches (C ABI of include/mc33_hip.h)
// ===================================================================================================
// ---------------------------------------------------------------------------------------------------
// Counts and bases of a z-slab that never leave the device (mc33hip_count_async ... mc33hip_emit_at_device_bases):
// the slab's {vertices, triangles} for a collective to gather, and what the emit passes need from the gathered table.
// ---------------------------------------------------------------------------------------------------
__global__ void k_publish_counts(const Counters *ctr, uint64_t ghost_segs, long long *dst) {
	const uint64_t gV = ghost_segs ? ctr->ghostV : 0, gT = ghost_segs ? ctr->ghostT : 0;
	dst[0] = (long long)(ctr->totV - gV);
	dst[1] = (long long)(ctr->totT - gT);
}
// table[r * stride] / [r * stride + 1]: vertices / triangles of rank r.  out: {id of this rank's first vertex = vertices of the ranks
// below; the rows at which it writes into the output arrays: the same when the arrays are the concatenated ones, 0 when they are its own}
__global__ void k_slab_bases(const long long *table, int stride, int rank, int concatenated, unsigned long long *out) {
	unsigned long long v = 0, t = 0;
	for (int r = 0; r < rank; r++) { v += (unsigned long long)table[(size_t)r * stride]; t += (unsigned long long)table[(size_t)r * stride + 1]; }
	out[0] = v; out[1] = concatenated ? v : 0ull; out[2] = concatenated ? t : 0ull;
}

static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof g_err, fmt, ap);
	va_end(ap);
	if (getenv("MC33_HIP_VERBOSE")) fprintf(stderr, "[mc33hip] %s\n", g_err);
}
#define HIP_TRY(expr)                                                                         \
	do {                                                                                      \
		hipError_t e_ = (expr);                                                               \
		if (e_ != hipSuccess) {                                                               \
			set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
			return e_ == hipErrorOutOfMemory ? MC33HIP_ENOMEM : MC33HIP_ERUNTIME;             \
		}                                                                                     \
	} while (0)

// Per-isovalue output of the sweep (SweepLane) and its bookkeeping on the host
constexpr int MC33_LANES = 8;
constexpr int MC33_MANY_PASSES = 4;  // passes of one mc33hip_sweep_many call: 8 isovalues = 4 + 4, 7 = 4 + 2 + 1
struct IsoLane {
	SliceHeader *slice_hdr;   // one record per (wave tile, cell slice) of the sweep
	uint4 *slice_bits;
	uint32_t *slice_compact;
	uint8_t *plane_fmt;
	unsigned long long *slot_part;
	uint4 *edge_bits, *edge_hdr;
	uint64_t slice_cap, edge_cap;
	uint32_t epoch;           // extractions since the slice headers were last cleared
	// a sweep made ahead of time by mc33hip_sweep_many, waiting for the count / extract call of its isovalue
	bool swept, boundary_done;
	bool tail_pending;        // a sweep has added this lane's slices into slot_part and no tail (k_slots) has consumed them yet
	bool tail_done;           // ... and the tail (k_slots ... k_scan_apply) of that sweep has been enqueued too, into the lane's own
	                          // TailSet: the count / extract call finds record ranges, records and prefix sums made
	double iso;
	mc33hip_range range;
	uint32_t pack;            // samples per lane and load of the sweep that filled it (lane_of_column)
	int many_pass, many_ni;   // which pass of mc33hip_sweep_many filled it (its events), and how many isovalues that pass classified
};

// Everything a tail (k_slots ... k_scan_apply) writes and the emit passes read, for ONE isovalue.  Lane k of the sweep buffers
// works with set k.  A single extraction uses lane 0 and set 0; an iso sweep (mc33hip_sweep_many) fills the sets of all its
// isovalues right behind each pass over the grid - one launch of every tail kernel for the up to four isovalues of the pass
// (PerLane) - and the count / extract calls that follow only emit (round 4; until then the one set was shared and every
// isovalue ran its nine tail launches by itself: 72 launches per 8-isovalue step, now 18).
struct TailSet {
	uint32_t *seg_cnt;
	SegDir *seg_dir;
	SegBase *seg_base;
	uint64_t seg_cap;
	uint64_t *bsV, *bsT;
	uint64_t bs_cap;
	EntryA *entries_a;
	EntryB *entries_b;
	EntryC *entries_c;
	uint32_t *entry_seg, *slow_list, *dirty_list;
	uint64_t entry_cap;
	BatchDesc *batches;       // the records in batches of <= 64 of one slice slot (k_cells writes, the emit passes walk)
	uint64_t batch_cap;
	uint32_t *list_cnt;       // [2][LIST_CHUNKS] cursors of the slow / dirty list parts (ListChunks)
	ListChunks lc;            // ... for the range last counted
	uint2 *slot_base;
	uint64_t slot_base_cap;
	uint32_t *live_list;      // [slot_base_cap]: slots with cut cells, k_slots -> k_cells
	uint32_t tail_serial;     // tails enqueued (seg_tagged)
	bool tail_incomplete;     // a tail was begun and did not reach its last launch: Counters::live_cursor may not be zero
	uint32_t records_hint;    // work records of the last extraction whose counters were read (grid of the triangle pass, first guess of a new set)
	uint32_t slow_hint;       // ... and its slow records + 1 (0: not known yet): the grid of k_emit_slow
	bool count_known, count_needed;  // ... and whether it had records waiting for k_slow_count (corners equal to the isovalue): see enqueue_tail
	Counters *d_ctr, *h_ctr;
	bool ctr_published;       // the emit pass enqueued last leaves the counters in h_ctr itself (k_emit_fast_triangles)
};

// The environment switches (developer A/B, the tests that force a code path), read ONCE when a context is created: a getenv walks
// the whole environment, and there were some twenty of them on every call.  0 / -1 / nullptr = not set: the library decides.
struct Switches {
	uint32_t rz, sweep_blocks_per_cu, min_depth, cells_blocks, slow_blocks, emit_blocks, emit_v_blocks_per_cu, slow_slots_max;
	bool no_pack, no_stage, verbose;
	int slow_count, tails_ahead, no_fork, slow_slots, tri_first, slow_merged;  // -1: not set
	char *trace_cells, *trace_file;                               // (developer tracing: file names; copies)
	uint32_t debug, cells_dev, sweep_subtract, tri_below, old_vertex_pass;  // (looked at by -DMC33_DEV builds only)
};

struct mc33hip_ctx {
	mc33hip_grid_desc desc;
	Switches sw;
	int device;
	hipStream_t stream;
	bool own_stream;          // the stream came from the pool (mc33hip_own_stream) and goes back there
	sample_t *d_grid;
	bool owns_grid;
	size_t pitch, slice;  // in samples
	uint16_t *d_lut;
	uint32_t *d_rules;
	uint8_t *d_rule_index;
	uint4 *d_fast;
	EntryB *d_fast_b;
	uint32_t *d_pat;
	IsoLane lanes[MC33_LANES]; // what a sweep leaves behind, per isovalue (lane 0: the single-isovalue calls)
	TailSet ts[MC33_LANES];    // ... and what its tail leaves behind (set k belongs to lane k)
	TailSet *w;                // the set of the lane the last count used: what emit and the counters refer to
	uint32_t cells_blocks;   // blocks of k_cells the GPU holds at once
	uint32_t epoch_wrap;      // the stamps start over at this count (2^30; MC33_HIP_EPOCH_WRAP for the test that crosses it)
	IsoLane *cur_lane;        // the lane the last count used (its epoch is what the emit pass needs)
	bool lane_presweeped;     // ... and it had been filled by mc33hip_sweep_many
	bool lane_pretailed;      // ... tail included
	SweepTile *d_tiles;       // block plan of k_sweep for the current range
	TileBoundary *d_bounds;   // pairs of tiles that meet in z (k_boundary)
	uint64_t tiles_cap, ntiles, nbounds;
	uint32_t tiles_zs, tiles_ze, tiles_depth;
	uint32_t resident_blocks; // k_sweep blocks the device holds at once
	int cus;                  // compute units of the device
	int emit_v_blocks_per_cu; // blocks of k_emit_vertices a CU holds
	hipEvent_t ev[4];
	hipEvent_t ev_many[MC33_MANY_PASSES][3];  // mc33hip_sweep_many's passes: recorded before the sweep, behind it, behind the tails made ahead; read in read_timing (nobody waits)
	hipStream_t aux, aux2;    // the triangle pass and the slow-record pass run beside the vertex pass
	hipStream_t copy;         // mc33hip_download_concurrent
	hipEvent_t ev_fork, ev_join, ev_join2;
	unsigned long long *d_bases;  // {id base, output vertex row, output triangle row} made on the device (mc33hip_bases_from_table)
	bool async_count;         // the last count was enqueued without waiting for its counters (mc33hip_count_async)
	hipEvent_t ev_dl[2];      // mc33hip_emit_download: behind the pass that completes T / behind the one that completes V and N
	bool emit_pending;        // an emit was enqueued after the last timing read
	int timing_level;         // MC33_HIP_TIMING: 0 none (default), 1 whole call, 2 per pass - the event records cost ~20 us per call
	bool inclined, triangular;   // non-orthogonal grid (MC33_spnC): _GRD._A / _GRD.A_ as given
	bool normal_neg;             // front and back exchanged (the reference's MC33_NORMAL_NEG compile-time switch)
	double grd_A[9], grd_Ai[9];
	unsigned long long *trace;  // developer tracing (MC33_HIP_TRACE_FILE)
	uint64_t trace_waves;
	unsigned long long *trace_cells;  // (MC33_HIP_TRACE_CELLS)
	uint64_t trace_cells_n;
	// state of the last count
	bool counted;
	Params P;
	mc33hip_range range;
	uint64_t nsegs, ghost_segs;
	mc33hip_counts counts;
	mc33hip_timing timing;
};

extern "C" const char *mc33hip_last_error(void) { return g_err; }
static uint32_t env_u32(const char *name, uint32_t dflt);

static int env_flag(const char *name) {  // -1: not set
	const char *s = getenv(name);
	return s && *s ? (atoi(s) != 0 ? 1 : 0) : -1;
}
static void read_switches(Switches &w) {
	w.rz = env_u32("MC33_HIP_RZ", 0); w.sweep_blocks_per_cu = env_u32("MC33_HIP_SWEEP_BLOCKS_PER_CU", 0); w.min_depth = env_u32("MC33_HIP_MIN_DEPTH", 0);
	w.cells_blocks = env_u32("MC33_HIP_CELLS_BLOCKS", 0); w.slow_blocks = env_u32("MC33_HIP_SLOW_BLOCKS", 0); w.emit_blocks = env_u32("MC33_HIP_EMIT_BLOCKS", 0);
	w.emit_v_blocks_per_cu = env_u32("MC33_HIP_EMIT_V_BLOCKS_PER_CU", 0); w.slow_slots_max = env_u32("MC33_HIP_SLOW_SLOTS_MAX", 0);
	w.no_pack = env_u32("MC33_HIP_NO_PACK", 0) != 0; w.no_stage = env_u32("MC33_HIP_NO_STAGE", 0) != 0; w.verbose = getenv("MC33_HIP_VERBOSE") != nullptr;
	w.slow_count = env_flag("MC33_HIP_SLOW_COUNT"); w.tails_ahead = env_flag("MC33_HIP_TAILS_AHEAD"); w.no_fork = env_flag("MC33_HIP_NO_FORK");
	w.slow_slots = env_flag("MC33_HIP_SLOW_SLOTS"); w.tri_first = env_flag("MC33_HIP_TRI_FIRST"); w.slow_merged = env_flag("MC33_HIP_SLOW_MERGED");
	w.trace_cells = getenv("MC33_HIP_TRACE_CELLS") ? strdup(getenv("MC33_HIP_TRACE_CELLS")) : nullptr;
	w.trace_file = getenv("MC33_HIP_TRACE_FILE") ? strdup(getenv("MC33_HIP_TRACE_FILE")) : nullptr;
	w.debug = env_u32("MC33_HIP_DEBUG", 0); w.cells_dev = env_u32("MC33_HIP_CELLS_DEV", 0); w.sweep_subtract = env_u32("MC33_HIP_SWEEP_SUBTRACT", 0);
	w.tri_below = env_u32("MC33_HIP_TRI_BELOW", 0); w.old_vertex_pass = env_u32("MC33_HIP_OLD_VERTEX_PASS", 0);
}

static int use_device(mc33hip_ctx *c) {
	HIP_TRY(hipSetDevice(c->device));
	return 0;
}

// Side streams are taken from a process-wide pool and handed back, never destroyed: hipStreamDestroy of a stream
// that events were recorded on leaves the HIP runtime (ROCm 7.x) with a dangling reference - its reference count
// is decremented after the stream object has been freed, which corrupts whatever the heap put there next
// (found with tools/uaf_trap.c under tools/soak.py: thousands of create_MC33 / free_MC33 pairs in one process).
namespace {
struct StreamPool {
	std::mutex m;
	std::vector<std::pair<int, hipStream_t>> idle;
};
StreamPool &stream_pool() {
	static StreamPool *p = new StreamPool;  // never destructed: no ordering problem with the runtime's own teardown
	return *p;
}
hipError_t pool_take(int device, hipStream_t *out) {
	StreamPool &sp = stream_pool();
	{
		std::lock_guard<std::mutex> g(sp.m);
		for (size_t k = 0; k < sp.idle.size(); k++)
			if (sp.idle[k].first == device) {
				*out = sp.idle[k].second;
				sp.idle[k] = sp.idle.back();
				sp.idle.pop_back();
				return hipSuccess;
			}
	}
	return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void pool_give(int device, hipStream_t s) {
	if (!s) return;
	(void)hipStreamSynchronize(s);
	StreamPool &sp = stream_pool();
	std::lock_guard<std::mutex> g(sp.m);
	sp.idle.emplace_back(device, s);
}
}  // namespace

// row pitch (samples) of the library's own copy of the grid: every row starts on a 16-byte boundary, whatever the sample
// type (the sweep loads dwords, k_emit_vertices stages 16-byte chunks of the rows)
static size_t own_pitch(size_t npx) {
	const size_t unit = sizeof(sample_t) >= 4 ? 4 : 16 / sizeof(sample_t);
	return (npx + unit - 1) / unit * unit;
}

static void free_set(TailSet &w) {  // (everything of the set; it can be filled again by ensure_set)
	(void)hipFree(w.seg_cnt); (void)hipFree(w.seg_dir); (void)hipFree(w.seg_base); (void)hipFree(w.bsV);
	(void)hipFree(w.entries_a); (void)hipFree(w.entries_b); (void)hipFree(w.entries_c); (void)hipFree(w.entry_seg); (void)hipFree(w.slow_list); (void)hipFree(w.dirty_list);
	(void)hipFree(w.batches); (void)hipFree(w.list_cnt); (void)hipFree(w.slot_base); (void)hipFree(w.live_list); (void)hipFree(w.d_ctr);
	if (w.h_ctr) (void)hipHostFree(w.h_ctr);
	w = TailSet{};
}

extern "C" int mc33hip_create(mc33hip_ctx **out, const mc33hip_grid_desc *d) {
	if (!out || !d) return MC33HIP_EINVAL;
	*out = nullptr;
	if (d->sample_bytes != MC33_SAMPLE_BYTES) { set_err("sample_bytes %d does not match this library (%d)", d->sample_bytes, MC33_SAMPLE_BYTES); return MC33HIP_EINVAL; }
	if (d->npx < 2 || d->npy < 2 || d->npz_resident < 2 || d->nz_total < 1) { set_err("grid needs at least 2 points per axis"); return MC33HIP_EINVAL; }
	if ((uint64_t)d->plane0 + d->npz_resident > (uint64_t)d->nz_total + 1) { set_err("resident planes exceed the grid"); return MC33HIP_EINVAL; }
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_err("no HIP device available"); return MC33HIP_ENOGPU; }
	mc33hip_ctx *c = (mc33hip_ctx *)calloc(1, sizeof *c);
	if (!c) return MC33HIP_ENOMEM;
	c->desc = *d;
	read_switches(c->sw);
	if (d->device >= 0) c->device = d->device;
	else if (hipGetDevice(&c->device) != hipSuccess) { free(c); set_err("hipGetDevice failed"); return MC33HIP_ENOGPU; }
	*out = c;
	int rc = use_device(c);
	if (rc) { free(c); *out = nullptr; return rc; }
	c->pitch = own_pitch(d->npx);
	c->slice = c->pitch * d->npy;
	hipError_t e;
#define CREATE_TRY(expr)                                                                    \
	if ((e = (expr)) != hipSuccess) {                                                       \
		set_err("%s failed: %s", #expr, hipGetErrorString(e));                              \
		mc33hip_destroy(c);                                                                 \
		*out = nullptr;                                                                     \
		return e == hipErrorOutOfMemory ? MC33HIP_ENOMEM : MC33HIP_ERUNTIME;                \
	}
	CREATE_TRY(hipMalloc(&c->d_lut, sizeof mc33_lut));
	CREATE_TRY(hipMalloc(&c->d_rules, sizeof mc33_rule_words));
	CREATE_TRY(hipMalloc(&c->d_rule_index, sizeof mc33_rule_index));
	CREATE_TRY(hipMemcpy(c->d_lut, mc33_lut, sizeof mc33_lut, hipMemcpyHostToDevice));
	CREATE_TRY(hipMemcpy(c->d_rules, mc33_rule_words, sizeof mc33_rule_words, hipMemcpyHostToDevice));
	CREATE_TRY(hipMemcpy(c->d_rule_index, mc33_rule_index, sizeof mc33_rule_index, hipMemcpyHostToDevice));
	{
		uint32_t fast[256];
		uint4 rec[256];
		build_fast_table(mc33_lut, fast);
		fast_record_table(fast, rec);
		CREATE_TRY(hipMalloc(&c->d_fast, sizeof rec));
		CREATE_TRY(hipMemcpy(c->d_fast, rec, sizeof rec, hipMemcpyHostToDevice));
		EntryB fb[256];
		fast_b_table(fast, fb);
		CREATE_TRY(hipMalloc(&c->d_fast_b, sizeof fb));
		CREATE_TRY(hipMemcpy(c->d_fast_b, fb, sizeof fb, hipMemcpyHostToDevice));
		constexpr uint32_t lut_n = sizeof mc33_lut / sizeof mc33_lut[0];
		uint32_t pat[lut_n];
		build_pattern_info(mc33_lut, lut_n, pat);
		CREATE_TRY(hipMalloc(&c->d_pat, sizeof pat));
		CREATE_TRY(hipMemcpy(c->d_pat, pat, sizeof pat, hipMemcpyHostToDevice));
	}
	c->w = &c->ts[0];  // (the sets get their memory when a range is known: ensure_set)
	for (int k = 0; k < 4; k++) CREATE_TRY(hipEventCreate(&c->ev[k]));
	for (int k = 0; k < MC33_MANY_PASSES; k++)
		for (int j = 0; j < 3; j++) CREATE_TRY(hipEventCreate(&c->ev_many[k][j]));
	CREATE_TRY(pool_take(c->device, &c->aux));
	CREATE_TRY(pool_take(c->device, &c->aux2));
	CREATE_TRY(pool_take(c->device, &c->copy));
	c->timing_level = getenv("MC33_HIP_TIMING") ? atoi(getenv("MC33_HIP_TIMING")) : 0;
	c->epoch_wrap = std::min(1u << 30, std::max(3u, env_u32("MC33_HIP_EPOCH_WRAP", 1u << 30)));
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
	CREATE_TRY(hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming));
	for (int k = 0; k < 2; k++) CREATE_TRY(hipEventCreateWithFlags(&c->ev_dl[k], hipEventDisableTiming));
#undef CREATE_TRY
	return MC33HIP_OK;
}

extern "C" void mc33hip_destroy(mc33hip_ctx *c) {
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	else (void)hipDeviceSynchronize();
	if (c->owns_grid) (void)hipFree(c->d_grid);
	(void)hipFree(c->d_lut); (void)hipFree(c->d_rules); (void)hipFree(c->d_rule_index); (void)hipFree(c->d_fast);
	(void)hipFree(c->d_fast_b); (void)hipFree(c->d_pat);
	for (int k = 0; k < MC33_LANES; k++) free_set(c->ts[k]);
	for (int k = 0; k < MC33_LANES; k++) {
		IsoLane &L = c->lanes[k];
		(void)hipFree(L.slice_hdr); (void)hipFree(L.slice_bits); (void)hipFree(L.slice_compact); (void)hipFree(L.plane_fmt); (void)hipFree(L.slot_part); (void)hipFree(L.edge_bits); (void)hipFree(L.edge_hdr);
	}
	(void)hipFree(c->d_tiles);
	(void)hipFree(c->d_bounds);
	(void)hipFree(c->d_bases);
	(void)hipFree(c->trace); (void)hipFree(c->trace_cells);
	if (c->aux) (void)hipStreamSynchronize(c->aux);
	if (c->aux2) (void)hipStreamSynchronize(c->aux2);
	if (c->copy) (void)hipStreamSynchronize(c->copy);
	for (int k = 0; k < 4; k++) if (c->ev[k]) (void)hipEventDestroy(c->ev[k]);
	for (int k = 0; k < MC33_MANY_PASSES; k++)
		for (int j = 0; j < 3; j++) if (c->ev_many[k][j]) (void)hipEventDestroy(c->ev_many[k][j]);
	if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
	if (c->ev_join) (void)hipEventDestroy(c->ev_join);
	if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
	for (int k = 0; k < 2; k++) if (c->ev_dl[k]) (void)hipEventDestroy(c->ev_dl[k]);
	pool_give(c->device, c->aux); pool_give(c->device, c->aux2); pool_give(c->device, c->copy);  // after the events
	if (c->own_stream) pool_give(c->device, c->stream);
	free(c->sw.trace_cells); free(c->sw.trace_file);
	free(c);
}

static void forget_sweeps(mc33hip_ctx *c) {  // the grid changed: sweeps made ahead of time are worthless
	for (int k = 0; k < MC33_LANES; k++) { c->lanes[k].swept = false; c->lanes[k].tail_done = false; }
}

extern "C" int mc33hip_set_stream(mc33hip_ctx *c, void *s) {
	if (!c) return MC33HIP_EINVAL;
	if (c->own_stream) { pool_give(c->device, c->stream); c->own_stream = false; }
	c->stream = (hipStream_t)s;
	return MC33HIP_OK;
}

extern "C" int mc33hip_own_stream(mc33hip_ctx *c) {
	if (!c) return MC33HIP_EINVAL;
	if (c->own_stream) return MC33HIP_OK;
	int rc = use_device(c);
	if (rc) return rc;
	hipStream_t st = nullptr;
	HIP_TRY(pool_take(c->device, &st));
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	c->stream = st;
	c->own_stream = true;
	return MC33HIP_OK;
}

extern "C" int mc33hip_device_count(void) {
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

static int ensure_grid(mc33hip_ctx *c) {
	if (c->d_grid && c->owns_grid) return 0;
	if (c->d_grid && !c->owns_grid) { c->d_grid = nullptr; }
	c->pitch = own_pitch(c->desc.npx);
	c->slice = c->pitch * c->desc.npy;
	// +64 samples of slack: tile loads clamp their addresses into the row, never past the buffer
	HIP_TRY(hipMalloc(&c->d_grid, (c->slice * c->desc.npz_resident + 64) * sizeof(sample_t)));
	c->owns_grid = true;
	return 0;
}

// Rows packed into the pitched device layout through a pinned staging buffer, one group of planes at a time.
// row(k, j): host address of row j of resident plane k.
template <typename RowFn>
static int upload_staged(mc33hip_ctx *c, RowFn row) {
	const uint32_t npy = c->desc.npy, npz = c->desc.npz_resident;
	const size_t rowb = (size_t)c->desc.npx * sizeof(sample_t);
	const size_t planeb = c->slice * sizeof(sample_t);
	size_t planes_per = (64u << 20) / planeb;
	if (planes_per < 1) planes_per = 1;
	if (planes_per > npz) planes_per = npz;
	char *stage = nullptr;
	HIP_TRY(hipHostMalloc(&stage, planes_per * planeb, hipHostMallocDefault));
	for (uint32_t k0 = 0; k0 < npz; k0 += (uint32_t)planes_per) {
		const uint32_t kn = (uint32_t)((k0 + planes_per <= npz) ? planes_per : npz - k0);
		for (uint32_t k = 0; k < kn; k++)
			for (uint32_t j = 0; j < npy; j++)
				memcpy(stage + k * planeb + (size_t)j * c->pitch * sizeof(sample_t), row(k0 + k, j), rowb);
		hipError_t e = hipMemcpy((char *)c->d_grid + (size_t)k0 * planeb, stage, (size_t)kn * planeb, hipMemcpyHostToDevice);
		if (e != hipSuccess) { (void)hipHostFree(stage); set_err("grid upload failed: %s", hipGetErrorString(e)); return MC33HIP_ERUNTIME; }
	}
	(void)hipHostFree(stage);
	return MC33HIP_OK;
}

extern "C" int mc33hip_upload_contiguous(mc33hip_ctx *c, const void *host) {
	if (!c || !host) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = ensure_grid(c))) return rc;
	const size_t rowb = (size_t)c->desc.npx * sizeof(sample_t);
	if (c->pitch != c->desc.npx && rowb % 4 != 0) {
		// rows that are not a whole number of dwords (odd-length uchar / ushort rows): the runtime's pitched copy from
		// pageable memory falls to 0.1-0.3 GB/s (3 s for a 0.8 GB grid); packing the rows ourselves runs at memcpy speed
		const char *h = (const char *)host;
		const size_t npy = c->desc.npy;
		if ((rc = upload_staged(c, [=](uint32_t k, uint32_t j) { return h + ((size_t)k * npy + j) * rowb; }))) return rc;
	} else
		HIP_TRY(hipMemcpy2D(c->d_grid, c->pitch * sizeof(sample_t), host, rowb, rowb, (size_t)c->desc.npy * c->desc.npz_resident,
		                    hipMemcpyHostToDevice));
	c->counted = false;
	forget_sweeps(c);
	return MC33HIP_OK;
}

extern "C" int mc33hip_upload_rows(mc33hip_ctx *c, const void *const *const *F) {
	if (!c || !F) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = ensure_grid(c))) return rc;
	const uint32_t npy = c->desc.npy, npz = c->desc.npz_resident;
	const size_t rowb = (size_t)c->desc.npx * sizeof(sample_t);
	// fast path: rows laid out back to back (grid_from_data_pointer, reference MC33_util_grd.c:609-611)
	bool contiguous = true;
	const char *expect = (const char *)F[0][0];
	for (uint32_t k = 0; k < npz && contiguous; k++)
		for (uint32_t j = 0; j < npy; j++, expect += rowb)
			if ((const char *)F[k][j] != expect) { contiguous = false; break; }
	if (contiguous) return mc33hip_upload_contiguous(c, F[0][0]);
	// rows are separate allocations (alloc_F, reference MC33_util_grd.c:147-169)
	if ((rc = upload_staged(c, [=](uint32_t k, uint32_t j) { return (const char *)F[k][j]; }))) return rc;
	c->counted = false;
	forget_sweeps(c);
	return MC33HIP_OK;
}

extern "C" int mc33hip_adopt_device(mc33hip_ctx *c, const void *dptr, size_t pitch, size_t slice) {
	if (!c || !dptr || pitch < c->desc.npx || slice < pitch * c->desc.npy) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if (c->owns_grid) (void)hipFree(c->d_grid);
	c->d_grid = (sample_t *)dptr;
	c->owns_grid = false;
	c->pitch = pitch;
	c->slice = slice;
	c->tiles_ze = 0u;  // (the plan counts its work in batches of the form - packed or not - this buffer's alignment allows)
	c->counted = false;
	forget_sweeps(c);
	return MC33HIP_OK;
}

static int check_range(mc33hip_ctx *c, const mc33hip_range *r) {
	const mc33hip_grid_desc &d = c->desc;
	if (!r || r->z_begin >= r->z_end || r->z_end > d.nz_total) { set_err("bad z range"); return MC33HIP_EINVAL; }
	if (r->ghost_below && r->z_begin == 0) { set_err("ghost slice below z = 0"); return MC33HIP_EINVAL; }
	const uint32_t zs = r->z_begin - (r->ghost_below ? 1u : 0u);
	// planes the passes touch: cells need planes z and z+1; normals read z+2 (MC:888, 1036, 1182, 1217)
	// when it exists, and z-1 for vertices on grid points (MC:643-647, 836, 909, 980, 1058)
	const uint32_t lo = zs ? zs - 1 : 0, hi = (r->z_end + 1 <= d.nz_total) ? r->z_end + 1 : d.nz_total;
	if (lo < d.plane0 || hi > d.plane0 + d.npz_resident - 1) {
		set_err("range needs planes %u..%u, resident are %u..%u", lo, hi, d.plane0, d.plane0 + d.npz_resident - 1);
		return MC33HIP_EINVAL;
	}
	if (!c->d_grid) { set_err("no grid uploaded"); return MC33HIP_EINVAL; }
	return 0;
}

static void fill_params(mc33hip_ctx *c, double iso, const mc33hip_range *r) {
	const mc33hip_grid_desc &d = c->desc;
	Params &P = c->P;
	P.nx = d.npx - 1; P.ny = d.npy - 1; P.nz = d.nz_total;
	P.nseg = (P.nx + SEG_CELLS - 1) / SEG_CELLS;
	P.zs = r->z_begin - (r->ghost_below ? 1u : 0u);
	P.iso = (real_t)iso;
	// store selection and float copies: MC:1772-1782
	if (d.d[0] != d.d[1] || d.d[1] != d.d[2]) { P.store_mode = 2; P.ca = (real_t)(d.d[2] / d.d[0]); P.cb = (real_t)(d.d[2] / d.d[1]); }
	else { P.store_mode = (d.d[0] == 1 && d.r0[0] == 0 && d.r0[1] == 0 && d.r0[2] == 0) ? 0 : 1; P.ca = P.cb = 1.0f; }
	P.triangular = 0;
	P.normal_neg = c->normal_neg ? 1 : 0;
	P.negzero_iso = (P.iso == 0 && sign_of(P.iso)) ? 1 : 0;
	for (int k = 0; k < 9; k++) P.A[k] = P.Ai[k] = 0.0;
	if (c->inclined) {  // G->nonortho: MC:1763-1770
		P.store_mode = 3;
		P.triangular = c->triangular;
		for (int j = 0; j < 3; j++)
			for (int i = 0; i < 3; i++) {
				P.A[3 * j + i] = c->grd_A[3 * j + i] * d.d[i];
				P.Ai[3 * j + i] = c->grd_Ai[3 * j + i] / d.d[j];
			}
	}
	for (int k = 0; k < 3; k++) { P.O[k] = (real_t)d.r0[k]; P.D[k] = (real_t)d.d[k]; }
	c->range = *r;
	c->nsegs = (uint64_t)(r->z_end - P.zs) * P.ny * P.nseg;
	c->ghost_segs = r->ghost_below ? (uint64_t)P.ny * P.nseg : 0;
}

static uint32_t env_u32(const char *name, uint32_t dflt) {
	const char *s = getenv(name);
	if (!s || !*s) return dflt;
	long v = strtol(s, nullptr, 10);
	return v > 0 ? (uint32_t)v : dflt;
}

static int alloc_entries(TailSet &w, uint64_t cap) {
	(void)hipFree(w.entries_a); (void)hipFree(w.entries_b); (void)hipFree(w.entries_c); (void)hipFree(w.entry_seg); (void)hipFree(w.slow_list); (void)hipFree(w.dirty_list);
	w.entries_a = nullptr; w.entries_b = nullptr; w.entries_c = nullptr; w.entry_seg = nullptr; w.slow_list = nullptr; w.dirty_list = nullptr;
	w.entry_cap = 0;
	if (cap > 0xFFFFFF00ull) cap = 0xFFFFFF00ull;
	const hipError_t e = [&]() -> hipError_t {
		hipError_t r;
		if ((r = hipMalloc(&w.entries_a, (cap + 2) * sizeof(EntryA))) != hipSuccess) return r;  // (+ 2: the triangle pass reads records in pairs)
		if ((r = hipMalloc(&w.entries_b, cap * sizeof(EntryB))) != hipSuccess) return r;        // (touched for tested and slow records only)
		if ((r = hipMalloc(&w.entries_c, cap * sizeof(EntryC))) != hipSuccess) return r;        // (... for slow records only)
		if ((r = hipMalloc(&w.entry_seg, cap * 4)) != hipSuccess) return r;
		if ((r = hipMalloc(&w.slow_list, cap * 4)) != hipSuccess) return r;
		return hipMalloc(&w.dirty_list, cap * 4);
	}();
	if (e != hipSuccess) {  // all or nothing: a set with some of its arrays would pass for a complete one (ensure_set looks at entries_a)
		(void)hipFree(w.entries_a); (void)hipFree(w.entries_b); (void)hipFree(w.entries_c); (void)hipFree(w.entry_seg); (void)hipFree(w.slow_list); (void)hipFree(w.dirty_list);
		w.entries_a = nullptr; w.entries_b = nullptr; w.entries_c = nullptr; w.entry_seg = nullptr; w.slow_list = nullptr; w.dirty_list = nullptr;
		set_err("work-record buffers (%llu records) failed: %s", (unsigned long long)cap, hipGetErrorString(e));
		return e == hipErrorOutOfMemory ? MC33HIP_ENOMEM : MC33HIP_ERUNTIME;
	}
	w.entry_cap = cap;
	return 0;
}

// the buffers of set w for the current range (c->nsegs, c->P, c->range); hint: work records to make room for at first (0: a guess from the range)
static int ensure_set(mc33hip_ctx *c, TailSet &w, uint64_t hint = 0) {
	if (!w.d_ctr) {
		HIP_TRY(hipMalloc(&w.d_ctr, sizeof(Counters)));
		HIP_TRY(hipMemset(w.d_ctr, 0, sizeof(Counters)));  // (live_cursor: every tail leaves it zero for the next)
		HIP_TRY(hipMalloc(&w.list_cnt, 2 * LIST_CHUNKS * sizeof(uint32_t)));
		HIP_TRY(hipHostMalloc(&w.h_ctr, sizeof(Counters), hipHostMallocDefault));
	}
	if (w.seg_cap < c->nsegs) {
		(void)hipFree(w.seg_cnt); (void)hipFree(w.seg_dir); (void)hipFree(w.seg_base);
		w.seg_cnt = nullptr; w.seg_dir = nullptr; w.seg_base = nullptr;
		w.seg_cap = 0;
		HIP_TRY(hipMalloc(&w.seg_cnt, c->nsegs * 4));
		w.tail_serial = 0;  // (the first tail clears the new array)
		HIP_TRY(hipMalloc(&w.seg_dir, c->nsegs * sizeof(SegDir)));
		HIP_TRY(hipMalloc(&w.seg_base, (c->nsegs + 1) * sizeof(SegBase)));  // (+ 1: the triangle pass reads bases in pairs)
		w.seg_cap = c->nsegs;
	}
	const uint64_t nb = (c->nsegs + SCAN_CHUNK - 1) / SCAN_CHUNK;
	if (w.bs_cap < nb) {
		(void)hipFree(w.bsV);
		w.bsV = w.bsT = nullptr;
		w.bs_cap = 0;
		HIP_TRY(hipMalloc(&w.bsV, 2 * (nb + scan_groups(nb)) * 8));  // chunk sums V, T; then group sums V, T
		w.bsT = w.bsV + nb;
		w.bs_cap = nb;
	}
	if (!w.entries_a || !w.entry_cap) {
		// first guess: one cell in 32 is cut (BASELINE fields: 0.4-6 % of the cells); grown on demand
		const uint64_t cells = (uint64_t)c->P.nx * c->P.ny * (c->range.z_end - c->P.zs);
		return alloc_entries(w, hint ? hint + hint / 4 + 65536 : cells / 32 + 65536);
	}
	return 0;
}
static int ensure_workspaces(mc33hip_ctx *c) { return ensure_set(c, c->ts[0]); }

static int grow_entries(TailSet &w, uint64_t need) {
	const uint64_t cap = need + need / 8 + 65536;
	if (cap > 0xFFFFFF00ull) { set_err("more than 2^32 work records"); return MC33HIP_EOVERFLOW; }
	return alloc_entries(w, cap);
}

// Block plan of k_sweep for cell slices [zs, ze): every (segment group, y tile) column is cut along z into
// chunks of about equal work (sample rows x waves x planes; a chunk re-reads one plane, so they are kept
// about `depth` slices deep), and the number of chunks is a whole multiple of what the device holds at
// once whenever the grid is large enough.  Measured on MI355X at 1024^3: with 1088 equal tiles on 256 CUs
// the 64 CUs that got a fifth block finished 12 % after the others.
static bool sweep_packed(const mc33hip_ctx *c);
static int plan_sweep(mc33hip_ctx *c, uint32_t zs, uint32_t ze) {
	const Params &P = c->P;
	const uint32_t depth = c->sw.rz ? c->sw.rz : 16u;
	if (c->d_tiles && c->tiles_zs == zs && c->tiles_ze == ze && c->tiles_depth == depth) return 0;
	if (!c->resident_blocks) {
		int per_cu = 0, cus = 0;
		HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sweep<1, 1>, 256, 0));
		HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
		const uint32_t want = c->sw.sweep_blocks_per_cu ? c->sw.sweep_blocks_per_cu : 4u;
		c->resident_blocks = (uint32_t)std::max(1, cus) * (uint32_t)std::max(1, std::min(per_cu, (int)want));
	}
	// A tile is what ONE wave streams: a row segment (256 samples in x) x a y tile (64 sample rows) x a run of planes.
	// Planning is done per group of up to 4 neighbouring segments (they read the same 4 KiB rows and are launched
	// side by side), but every wave gets a tile of its own, so a group with fewer than 4 segments (grids that are
	// not a multiple of 1024 samples wide, narrow grids) does not leave the waves of a block idle.
	const uint32_t nXG = (P.nseg + 3) / 4, nYT = (P.ny + 62) / 63, nzc = ze - zs;
	const uint64_t ncol = (uint64_t)nXG * nYT;  // groups
	std::vector<double> w(ncol);       // work of one WAVE of the group per slice: its sample rows
	std::vector<uint32_t> waves(ncol);
	double W = 0;                      // ... summed over all waves
	for (uint32_t yt = 0; yt < nYT; yt++)
		for (uint32_t xg = 0; xg < nXG; xg++) {
			const uint64_t i = (uint64_t)yt * nXG + xg;
			waves[i] = std::min(4u, P.nseg - xg * 4);
			// (in whole BATCHES of sample rows - 4 rows of float samples, 8 / 16 of packed ushort / uchar ones: a wave's time is the
			// number of batches it waits for, and a y tile of 2 rows costs a batch per plane like one of 4.  Counted in rows, the
			// 2-row last y tile of a 128^3 grid got pieces twice as deep as its time allows and the sweep took 0.060 ms where the
			// 256^3 one takes 0.035: VERDICT r4 weak 12)
			const uint32_t rb = sweep_packed(c) ? 4u * (uint32_t)SWEEP_PACK : 4u;
			w[i] = (double)((std::min(64u, P.ny + 1 - yt * 63u) + rb - 1u) / rb * rb);
			W += w[i] * waves[i];
		}
	// wave tiles wanted: W * nzc / (64 * depth), rounded to whole rounds of the resident set
	const double pref = W * nzc / (64.0 * depth);
	uint64_t B = (uint64_t)c->resident_blocks * 4;
	if (pref >= (double)B) B *= (uint64_t)(pref / (double)B + 0.5);
	else B = std::max<uint64_t>(1, std::min<uint64_t>(B, (uint64_t)(W * nzc / (64.0 * std::max(1u, c->sw.min_depth)))));  // small grid: fill the GPU, tiles down to one plane deep (k_boundary then does the slices)
	std::vector<uint32_t> chunks(ncol);  // z pieces of the group (each is one tile per wave of the group)
	std::vector<std::pair<double, uint64_t>> frac(ncol);
	uint64_t total = 0;
	for (uint64_t i = 0; i < ncol; i++) {
		const double share = (double)B * w[i] / W;
		const uint32_t n = (uint32_t)std::min<double>(std::max(1.0, std::floor(share)), (double)nzc);
		chunks[i] = n; total += (uint64_t)n * waves[i];
		frac[i] = {share - std::floor(share), i};
	}
	std::sort(frac.begin(), frac.end(), [](const std::pair<double, uint64_t> &x, const std::pair<double, uint64_t> &y) { return x.first > y.first; });
	for (uint64_t k = 0; k < ncol && total < B; k++)
		if (chunks[frac[k].second] < nzc) { chunks[frac[k].second]++; total += waves[frac[k].second]; }
	if (total > 0x3FFFFFFFull) { set_err("grid too large for one launch"); return MC33HIP_EINVAL; }
	struct Planned { SweepTile t; uint64_t col; };  // col: the wave's column (yt, seg)
	std::vector<Planned> planned;
	planned.reserve(total);
	for (uint64_t i = 0; i < ncol; i++)
		for (uint32_t k = 0; k < chunks[i]; k++) {
			const uint32_t lo = zs + (uint32_t)((uint64_t)nzc * k / chunks[i]), hi = zs + (uint32_t)((uint64_t)nzc * (k + 1) / chunks[i]);
			if (hi <= lo) continue;
			const uint32_t yt = (uint32_t)(i / nXG), xg = (uint32_t)(i % nXG);
			for (uint32_t sgm = xg * 4; sgm < xg * 4 + waves[i]; sgm++)
				planned.push_back(Planned{SweepTile{sgm, yt, lo, hi}, (uint64_t)yt * P.nseg + sgm});
		}
	// launch order: by depth first, so that waves running together read neighbouring memory (the segments of a group
	// stay next to each other: the sort is stable)
	std::stable_sort(planned.begin(), planned.end(), [](const Planned &x, const Planned &y) { return x.t.z_lo < y.t.z_lo; });
	std::vector<SweepTile> tiles(planned.size());
	std::vector<TileBoundary> bounds;
	{
		std::vector<uint32_t> below((uint64_t)nYT * P.nseg, 0xFFFFFFFFu);  // the tile of the column that ends where the next one begins
		for (uint32_t b = 0; b < planned.size(); b++) {   // (ascending z_lo: a column's tiles come in order)
			const SweepTile &t = planned[b].t;
			tiles[b] = t;
			if (below[planned[b].col] != 0xFFFFFFFFu) bounds.push_back(TileBoundary{below[planned[b].col], b, t.z_lo, t.yt, t.seg, {0, 0, 0}});
			below[planned[b].col] = b;
		}
	}
	// A sweep made ahead by mc33hip_sweep_many may still be reading the old plan: the copies below go through the null stream,
	// which a non-blocking stream (any torch.cuda.Stream) is not ordered with
	HIP_TRY(hipStreamSynchronize(c->stream));
	if (c->tiles_cap < tiles.size()) {
		(void)hipFree(c->d_tiles);
		c->d_tiles = nullptr; c->tiles_cap = 0;
		HIP_TRY(hipMalloc(&c->d_tiles, tiles.size() * sizeof(SweepTile)));
		c->tiles_cap = tiles.size();
	}
	HIP_TRY(hipMemcpy(c->d_tiles, tiles.data(), tiles.size() * sizeof(SweepTile), hipMemcpyHostToDevice));
	c->ntiles = tiles.size();
	(void)hipFree(c->d_bounds);
	c->d_bounds = nullptr;
	c->nbounds = bounds.size();
	if (c->nbounds) {
		HIP_TRY(hipMalloc(&c->d_bounds, bounds.size() * sizeof(TileBoundary)));
		HIP_TRY(hipMemcpy(c->d_bounds, bounds.data(), bounds.size() * sizeof(TileBoundary), hipMemcpyHostToDevice));
	}
	c->tiles_zs = zs; c->tiles_ze = ze; c->tiles_depth = depth;
	if (c->sw.verbose)
		fprintf(stderr, "[mc33hip] sweep plan: %llu wave tiles (%u resident), %llu columns, depth %.1f\n", (unsigned long long)c->ntiles,
		        c->resident_blocks * 4, (unsigned long long)nYT * P.nseg, (double)nzc * nYT * P.nseg / (double)c->ntiles);
	return 0;
}

// slot geometry of the range being classified
struct SlotGeom {
	uint32_t nYT, nseg;
	SlotDims sd;
	uint64_t cell_blocks, nslots, nchunks;
};
static int slot_geometry(mc33hip_ctx *c, SlotGeom &g) {
	const Params &P = c->P;
	g.nYT = (P.ny + 62) / 63;
	g.nseg = P.nseg;  // (the slots of a slice group are its real row segments)
	g.sd = SlotDims{(c->range.z_end - P.zs + 1 + 3) / 4, g.nYT, g.nseg};  // the plane above the last slice has a slot too
	g.cell_blocks = (uint64_t)g.sd.nZG * g.nYT * g.nseg;
	if (g.cell_blocks > 0x3FFFFFFFull) { set_err("grid too large for one launch"); return MC33HIP_EINVAL; }
	g.nslots = g.cell_blocks * 4;
	g.nchunks = 0;
	return 0;
}

// buffers of one isovalue lane for the current range and tile plan; a new extraction number (epoch)
static int begin_lane(mc33hip_ctx *c, IsoLane &L, const SlotGeom &g, hipStream_t st) {
	if (L.slice_cap < g.nslots) {
		(void)hipFree(L.slice_hdr); (void)hipFree(L.slice_bits); (void)hipFree(L.slice_compact); (void)hipFree(L.plane_fmt); (void)hipFree(L.slot_part);
		L.slice_hdr = nullptr; L.slice_bits = nullptr; L.slice_compact = nullptr; L.plane_fmt = nullptr; L.slot_part = nullptr; L.slice_cap = 0;
		HIP_TRY(hipMalloc(&L.slice_hdr, g.nslots * sizeof(SliceHeader)));
		HIP_TRY(hipMalloc(&L.slice_bits, g.nslots * 2048));  // (slots of planes: SlotDims)
		HIP_TRY(hipMalloc(&L.slice_compact, g.nslots * 256));
		HIP_TRY(hipMalloc(&L.plane_fmt, g.nslots));
		const uint64_t part_bytes = ((g.nslots + SLOT_CHUNK - 1) / SLOT_CHUNK) * 8;
		HIP_TRY(hipMalloc(&L.slot_part, 2 * part_bytes));  // two halves, used by alternate extractions
		HIP_TRY(hipMemsetAsync(L.slice_hdr, 0, g.nslots * sizeof(SliceHeader), st));
		HIP_TRY(hipMemsetAsync(L.slot_part, 0, 2 * part_bytes, st));
		L.epoch = 0;
		L.tail_pending = false;
		L.slice_cap = g.nslots;
	}
	if (L.edge_cap < c->ntiles) {
		(void)hipFree(L.edge_bits); (void)hipFree(L.edge_hdr);
		L.edge_bits = nullptr; L.edge_hdr = nullptr; L.edge_cap = 0;
		HIP_TRY(hipMalloc(&L.edge_bits, c->ntiles * 2 * 128 * sizeof(uint4)));
		HIP_TRY(hipMalloc(&L.edge_hdr, c->ntiles * 2 * 2 * sizeof(uint4)));
		L.edge_cap = c->ntiles;
	}
	const uint64_t nchunks = (L.slice_cap + SLOT_CHUNK - 1) / SLOT_CHUNK;  // (capacity: the halves keep their place)
	if (++L.epoch >= c->epoch_wrap) {  // stamps wrap: start over with clean headers AND clean partial sums - the call before
		// accumulated into the half an odd epoch selects and cleared only the other one, and epoch 1 is odd again
		HIP_TRY(hipMemsetAsync(L.slice_hdr, 0, L.slice_cap * sizeof(SliceHeader), st));
		HIP_TRY(hipMemsetAsync(L.slot_part, 0, 2 * nchunks * 8, st));
		L.epoch = 1;
	}
	if (L.tail_pending) {
		// The sweep before this one added its slices into the half its epoch selected, and no k_slots ever consumed them and
		// cleared the other half for this epoch (a sweep made ahead by mc33hip_sweep_many whose isovalue was never asked for,
		// the grid was re-uploaded, a call failed in between): the half this epoch accumulates into still holds the sums of
		// two extractions ago.  Start from clean sums.
		HIP_TRY(hipMemsetAsync(L.slot_part, 0, 2 * nchunks * 8, st));
		L.tail_pending = false;
	}
	L.swept = false;
	L.tail_done = false;
	L.boundary_done = false;
	return 0;
}
static unsigned long long *lane_part(const IsoLane &L, bool next) {
	const uint64_t nchunks = (L.slice_cap + SLOT_CHUNK - 1) / SLOT_CHUNK;
	return L.slot_part + ((L.epoch + (next ? 1u : 0u)) & 1u) * nchunks;
}

static void sweep_args(mc33hip_ctx *c, const SlotGeom &g, SweepArgs &a) {
	a.G.p = c->d_grid; a.G.pitch = (uint32_t)c->pitch; a.G.z0 = c->desc.plane0; a.G.slice = c->slice;
	a.P = c->P;
	a.sd = g.sd;
	a.tiles = c->d_tiles;
	a.ntiles = (uint32_t)c->ntiles;
	a.z_end = c->range.z_end;
	a.trace = nullptr;
	a.debug = 0;
	for (int q = 0; q < SWEEP_MAXNI; q++) a.lane[q] = SweepLane{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, (real_t)0, 0, 0xFFFFFFFFu};
}
// SweepLane::iso_gt / iso_eq of an isovalue for packed samples of `top` as their largest value
static void sweep_iso_words(real_t v, real_t top, int32_t &gt, uint32_t &eq) {
	const real_t fl = std::floor(v);
	gt = !(v == v) ? 0x7FFFFFFF : fl < 0 ? -1 : fl >= top ? (int32_t)top : (int32_t)fl;
	eq = (v == fl && fl >= 0 && fl <= top) ? (uint32_t)fl : 0xFFFFFFFFu;
}
static void set_lane(SweepArgs &a, int q, const IsoLane &L, double iso) {
	a.lane[q] = SweepLane{L.slice_hdr, L.slice_bits, L.slice_compact, L.plane_fmt, lane_part(L, false), L.edge_bits, L.edge_hdr, L.epoch, (real_t)iso, 0, 0xFFFFFFFFu};
	if (SWEEP_PACK > 1) sweep_iso_words((real_t)iso, SWEEP_PACK == 2 ? (real_t)65535 : (real_t)255, a.lane[q].iso_gt, a.lane[q].iso_eq);
}

// narrow samples are loaded as dwords when every row of the grid starts on a dword boundary (always true for the
// library's own pitched copy; a caller's device buffer may have any pitch)
static bool sweep_packed(const mc33hip_ctx *c) {
	return SWEEP_PACK > 1 && !c->sw.no_pack && ((uintptr_t)c->d_grid % 4u) == 0 && (c->pitch * sizeof(sample_t)) % 4u == 0 &&
	       (c->slice * sizeof(sample_t)) % 4u == 0;
}

// one k_sweep launch over NI = 1, 2 or 4 lanes that begin_lane has prepared
// (all return the samples per lane and load of the form that was launched: k_cells needs it - lane_of_column)
template <int NI, int ZM>
static uint32_t launch_sweep_zm(mc33hip_ctx *c, const SweepArgs &a, hipStream_t st) {
	const uint64_t blocks = (c->ntiles + 3) / 4;
	// (packed narrow samples, several isovalues, classified by subtraction - an isovalue of -0.0 among them: the conversions of
	// a batch and the sets of bit rows do not fit the registers of 3 waves per SIMD - 468 bytes of scratch per lane for uchar
	// with four isovalues, 20 for ushort, 8 for uchar with two; that corner takes the unpacked form, which has none)
	constexpr bool packed_form = !(SWEEP_PACK >= 2 && NI >= 2 && ZM == 0);
	if (packed_form && sweep_packed(c)) {
		hipLaunchKernelGGL((k_sweep<packed_form ? SWEEP_PACK : 1, NI, ZM>), dim3((uint32_t)blocks), dim3(256), 0, st, a);
		return (uint32_t)SWEEP_PACK;
	}
	hipLaunchKernelGGL((k_sweep<1, NI, ZM>), dim3((uint32_t)blocks), dim3(256), 0, st, a);
	return 1u;
}
template <int NI>
static uint32_t launch_sweep_ni(mc33hip_ctx *c, const SweepArgs &a, hipStream_t st) {
#ifdef MC33_INT_SAMPLES
	// which classification the isovalues of this pass allow (k_sweep's ZM)
	bool negzero = false, can_equal = false;
	for (int q = 0; q < NI; q++) {
		const real_t iso = a.lane[q].iso;
		negzero |= iso == 0 && sign_of(iso);
		can_equal |= iso >= 0 && iso <= (real_t)std::numeric_limits<sample_t>::max() && iso == std::floor(iso);
	}
	bool subtract = negzero;
#ifdef MC33_DEV
	subtract |= c->sw.sweep_subtract != 0;  // (A/B of the two forms; same results)
#endif
	if (!subtract) {
		if (can_equal) return launch_sweep_zm<NI, 1>(c, a, st);
		return launch_sweep_zm<NI, 2>(c, a, st);
	}
#endif
	return launch_sweep_zm<NI, 0>(c, a, st);
}

// Parameters of the passes for the isovalue of one lane (fill_params made c->P for the call's own isovalue)
static Params lane_params(const mc33hip_ctx *c, double iso) {
	Params P = c->P;
	P.iso = (real_t)iso;
	P.negzero_iso = (P.iso == 0 && sign_of(P.iso)) ? 1 : 0;
	return P;
}

// Everything after the sweep for the slices the lanes idx[0 .. n) hold (n <= SWEEP_MAXNI; lane idx[q] works into set sidx[q]): tile
// boundaries, record ranges, cell records, slow-cell planning, scans - ONE launch of each kernel for all n isovalues
// (PerLane, blockIdx.y), on the context's stream, no synchronisation.  isos[q]: the isovalue of lane idx[q].
static int enqueue_tail(mc33hip_ctx *c, const int *idx, const int *sidx, const double *isos, int n, const SlotGeom &g) {
	hipStream_t st = c->stream;
	const uint32_t ze = c->range.z_end;
	SweepArgs a;
	sweep_args(c, g, a);
#ifdef MC33_DEV
	a.debug = c->sw.debug;
#endif
	PerLane<SlotsArgs> SA;
	PerLane<CellsArgs> CA;
	PerLane<SlowArgs> WA;
	PerLane<ScanArgs> NA;
	static_assert(sizeof(PerLane<CellsArgs>) <= 3584 && sizeof(PerLane<SlowArgs>) <= 3584, "kernel argument segment");
	memset(&SA, 0, sizeof SA); memset(&CA, 0, sizeof CA); memset(&WA, 0, sizeof WA); memset(&NA, 0, sizeof NA);
	if (!c->cells_blocks) {  // (asked before anything of the tail is launched: nothing below can return between k_slots and k_scan_apply)
		int per_cu = 0, cus = 0;
		HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_cells, 256, 0));
		HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
		c->cells_blocks = (uint32_t)std::max(1, per_cu) * (uint32_t)std::max(1, cus);
	}
	const uint32_t nb = (uint32_t)((c->nsegs + SCAN_CHUNK - 1) / SCAN_CHUNK);
	bool boundaries = false;
	for (int q = 0; q < n; q++) {
		IsoLane &L = c->lanes[idx[q]];
		TailSet &w = c->ts[sidx[q]];
		const Params P = lane_params(c, isos[q]);
		if (w.slot_base_cap < g.nslots) {
			(void)hipFree(w.slot_base); (void)hipFree(w.live_list);
			w.slot_base = nullptr; w.live_list = nullptr; w.slot_base_cap = 0;
			HIP_TRY(hipMalloc(&w.slot_base, g.nslots * sizeof(uint2)));
			HIP_TRY(hipMalloc(&w.live_list, g.nslots * sizeof(uint32_t)));
			w.slot_base_cap = g.nslots;
		}
		{  // batch descriptors: every 64 records one, plus at most one partly filled batch per slice slot
			const uint64_t need = w.entry_cap / 64 + g.nslots + 64;
			if (w.batch_cap < need) {
				(void)hipFree(w.batches);
				w.batches = nullptr; w.batch_cap = 0;
				HIP_TRY(hipMalloc(&w.batches, need * sizeof(BatchDesc)));
				w.batch_cap = need;
			}
		}
		set_lane(a, q, L, isos[q]);
		boundaries |= !L.boundary_done;
		CellsArgs &ca = CA.a[q];
		ca.pack = L.pack ? L.pack : 1u;
		ca.dev = 0;
#ifdef MC33_DEV
		ca.dev = c->sw.cells_dev;
#endif
		ca.G.p = c->d_grid; ca.G.pitch = (uint32_t)c->pitch; ca.G.z0 = c->desc.plane0; ca.G.slice = c->slice;
		ca.P = P; ca.fast = c->d_fast; ca.pat = c->d_pat;
		ca.ze = ze; ca.sd = g.sd;
		ca.slice_hdr = L.slice_hdr; ca.slice_bits = L.slice_bits; ca.slice_compact = L.slice_compact; ca.plane_fmt = L.plane_fmt; ca.slot_base = w.slot_base;
		// the tag of this tail's row-segment counts; the array is cleared whenever the tags start over
		if (w.tail_serial % SEG_TAGS == 0) HIP_TRY(hipMemsetAsync(w.seg_cnt, 0, w.seg_cap * 4, st));
		const uint32_t seg_tag = w.tail_serial % SEG_TAGS + 1u;
		w.tail_serial++;
		ca.seg_tag = seg_tag;
		ca.live_list = w.live_list; ca.live_cap = (uint32_t)std::min<uint64_t>(g.nslots, 0xFFFFFFFFull);
		ca.epoch = L.epoch;
		ca.seg_cnt = w.seg_cnt; ca.seg_dir = w.seg_dir;
		ca.entries_a = w.entries_a; ca.entries_b = w.entries_b; ca.entry_seg = w.entry_seg; ca.slow_list = w.slow_list; ca.dirty_list = w.dirty_list;
		ca.entry_cap = (uint32_t)w.entry_cap;
		ca.batches = w.batches; ca.batch_cap = (uint32_t)std::min<uint64_t>(w.batch_cap, 0xFFFFFFFFull);
		ca.ctr = w.d_ctr;
		ca.trace = nullptr;
		if (n == 1 && c->sw.trace_cells) {
			(void)hipFree(c->trace_cells);
			c->trace_cells = nullptr;
			c->trace_cells_n = g.nslots;
			HIP_TRY(hipMalloc(&c->trace_cells, g.nslots * 32));
			HIP_TRY(hipMemsetAsync(c->trace_cells, 0, g.nslots * 32, st));
			ca.trace = c->trace_cells;
		}
		{  // groups of slots for the slow / dirty lists: at most LIST_CHUNKS
			uint32_t shift = 6;
			while (((g.nslots + (1ull << shift) - 1) >> shift) > LIST_CHUNKS) shift++;
			w.lc = ListChunks{w.list_cnt, w.list_cnt + LIST_CHUNKS, (uint32_t)((g.nslots + (1ull << shift) - 1) >> shift), shift};
		}
		ca.lc = w.lc;
		const uint64_t nchunks = (L.slice_cap + SLOT_CHUNK - 1) / SLOT_CHUNK;
		SA.a[q] = SlotsArgs{L.slice_hdr, lane_part(L, false), lane_part(L, true), (uint32_t)nchunks, L.epoch, w.slot_base, w.d_ctr, w.lc,
		                    (unsigned long long *)(w.bsV + 2 * w.bs_cap), (uint32_t)(2 * scan_groups(w.bs_cap)), w.live_list, ca.live_cap};
		SlowArgs &sa = WA.a[q];
		sa.G = ca.G; sa.P = P;
		sa.tab.lut = c->d_lut; sa.tab.rule_words = c->d_rules; sa.tab.rule_index = c->d_rule_index;
		sa.z_emit = c->range.z_begin;
		sa.entries_a = w.entries_a; sa.entries_b = w.entries_b; sa.entries_c = w.entries_c; sa.fast_b = c->d_fast_b; sa.entry_seg = w.entry_seg; sa.slow_list = w.slow_list;
		sa.seg_cnt = w.seg_cnt; sa.seg_tag = seg_tag; sa.seg_dir = w.seg_dir; sa.dirty_list = w.dirty_list;
		sa.lc = w.lc; sa.slot_base = w.slot_base;
		sa.entry_cap = (uint32_t)w.entry_cap; sa.ctr = w.d_ctr;
		uint64_t *grV = nb >= SCAN_GROUPED_FROM ? w.bsV + 2 * w.bs_cap : nullptr, *grT = grV ? grV + scan_groups(w.bs_cap) : nullptr;
		NA.a[q] = ScanArgs{w.seg_cnt, seg_tag, w.bsV, w.bsT, grV, grT, w.seg_base, w.d_ctr};
		// k_slots appends to live_list from Counters::live_cursor on, and k_scan_apply - the last kernel of a tail - leaves the
		// cursor zero for the next.  A tail that was cut short (a launch error) leaves it wherever it was: the next one starts clean.
		if (w.tail_incomplete) HIP_TRY(hipMemsetAsync(&w.d_ctr->live_cursor, 0, sizeof(uint32_t), st));
		w.tail_incomplete = true;
		w.ctr_published = false;  // (the counters of THIS tail are on the device until somebody brings them over)
	}
	const uint32_t ny = (uint32_t)n;
	if (c->nbounds && boundaries && !(MC33_DEBUG_BITS(a) & 2u))  // (the lanes of a call are all fresh, or it is one lane)
		hipLaunchKernelGGL(k_boundary, dim3((uint32_t)((c->nbounds + 3) / 4), ny), dim3(256), 0, st, a, c->d_bounds, (uint32_t)c->nbounds);
	for (int q = 0; q < n; q++) c->lanes[idx[q]].boundary_done = true;  // (its slices are in the partial sums now: a repeated tail - more room for records - must not add them again)
	hipLaunchKernelGGL(k_slots, dim3((uint32_t)((g.nslots + SLOT_CHUNK - 1) / SLOT_CHUNK), ny), dim3(256), 0, st, SA, g.nslots);
	for (int q = 0; q < n; q++) c->lanes[idx[q]].tail_pending = false;  // k_slots has read this epoch's partial sums and cleared the half of the next one
	// (four times what the GPU holds at once: slices differ in length, and a block that starts late evens the waves out -
	// 76 -> 66 us at 1024^3; a block per group of four slots, as until round 3, is 17 408 blocks there)
	hipLaunchKernelGGL(k_cells, dim3((uint32_t)std::min<uint64_t>(g.cell_blocks, c->sw.cells_blocks ? c->sw.cells_blocks : 4u * c->cells_blocks), ny), dim3(256), 0, st, CA);
	// (blocks beyond the lists end at once.  A grid sized from the last extraction's counts - 69 blocks instead of 1024 at 1024^3,
	// whose 8 820 slow cells are 35 blocks' worth - changes nothing: 13.5 / 6 / 8.7 us either way.  What these kernels take is
	// the chain of dependent loads of the cells that ARE slow, not their empty blocks; round 3)
	const uint32_t slow_blocks = c->sw.slow_blocks ? c->sw.slow_blocks : 1024u;
	// planning, identity counts and segment offsets: three launches - or, with MC33_HIP_SLOW_MERGED=1, ONE whose blocks wait for
	// each other (k_slow_all: 64 blocks at most, all resident together)
	bool merged = false;  // (measured and lost: see k_slow_all)
	uint32_t hint = 0;
	for (int q = 0; q < n; q++) hint = std::max(hint, c->ts[sidx[q]].slow_hint);
	if (c->sw.slow_merged >= 0) merged = c->sw.slow_merged != 0;
	if (merged) {
		const uint32_t blocks = std::min(SLOW_ALL_MAX_BLOCKS, std::max(1u, (hint + hint / 4u + 255u) / 256u));  // (any number is right: the phases stride)
		hipLaunchKernelGGL(k_slow_all, dim3(blocks, ny), dim3(256), 0, st, WA);
	} else {
	hipLaunchKernelGGL(k_slow_plan, dim3(slow_blocks, ny), dim3(256), 0, st, WA);
	{  // k_slow_count only when the last extraction of (one of) the set(s) had records for it, or nothing is known: k_seg_fix counts what is left over
		bool wanted = false;
		for (int q = 0; q < n; q++) wanted |= !c->ts[sidx[q]].count_known || c->ts[sidx[q]].count_needed;
		if (c->sw.slow_count >= 0) wanted = c->sw.slow_count != 0;
		if (wanted) hipLaunchKernelGGL(k_slow_count, dim3(slow_blocks, ny), dim3(256), 0, st, WA);
	}
	hipLaunchKernelGGL(k_seg_fix, dim3(slow_blocks, ny), dim3(256), 0, st, WA);
	}
	hipLaunchKernelGGL(k_scan_reduce, dim3(nb, ny), dim3(256), 0, st, NA, c->nsegs, c->P);
	hipLaunchKernelGGL(k_scan_apply, dim3(nb, ny), dim3(256), 0, st, NA, c->nsegs, c->P, c->ghost_segs);
	HIP_TRY(hipGetLastError());
	for (int q = 0; q < n; q++) c->ts[sidx[q]].tail_incomplete = false;
	return 0;
}

// the isovalue a lane was swept for, compared by bit pattern: +0.0 and -0.0 compare equal but classify samples equal to
// them differently (v = iso - F = -0 is "zero AND negative", Params::negzero_iso), and a NaN is the same NaN
static bool same_bits(real_t a, real_t b) { return memcmp(&a, &b, sizeof a) == 0; }
static bool same_range(const mc33hip_range &x, const mc33hip_range &y) {
	return x.z_begin == y.z_begin && x.z_end == y.z_end && (x.ghost_below != 0) == (y.ghost_below != 0);
}

// enqueue sweep + cell records + slow-cell planning + scans on the context's stream (no synchronisation).  When
// mc33hip_sweep_many has already classified this isovalue over this range, its lane is used and nothing is streamed - and
// when it has made the tail ahead as well (the lane's own TailSet), nothing is enqueued at all: the counters are waiting.
static int enqueue_count(mc33hip_ctx *c, bool rerun = false) {
	const Params &P = c->P;
	hipStream_t st = c->stream;
	if (int rc = plan_sweep(c, P.zs, c->range.z_end)) return rc;
	SlotGeom g;
	if (int rc = slot_geometry(c, g)) return rc;
	IsoLane *L = nullptr;
	bool tail_made = false;
	if (rerun && c->cur_lane && c->lane_presweeped) L = c->cur_lane;  // same call, more room for records: the sweep's result stands, the tail is made again
	else
		for (int k = 0; k < MC33_LANES && !L; k++)
			if (c->lanes[k].swept && same_bits((real_t)c->lanes[k].iso, P.iso) && same_range(c->lanes[k].range, c->range) &&
			    c->lanes[k].slice_cap >= g.nslots && c->lanes[k].edge_cap >= c->ntiles) {
				L = &c->lanes[k];
				tail_made = L->tail_done;
			}
	if (c->timing_level > 0) HIP_TRY(hipEventRecord(c->ev[0], st));
	if (L) {
		// a sweep made ahead is used once when its tail is made here (in the shared set 0, which the next isovalue overwrites); a
		// lane whose tail was made ahead as well - in a set of its own - stays good for any number of count / emit calls until the
		// next mc33hip_sweep_many or a change of the grid (slabs: count all isovalues, ONE exchange of counts, then the emits)
		if (!tail_made) { L->swept = false; L->tail_done = false; }
		c->lane_presweeped = true;
	} else {
		L = &c->lanes[0];
		if (int rc = begin_lane(c, *L, g, st)) return rc;
		c->lane_presweeped = false;
		SweepArgs a;
		sweep_args(c, g, a);
		set_lane(a, 0, *L, P.iso);
#ifdef MC33_DEV
		a.debug = c->sw.debug;
		if (a.debug) {  // never silent: with this set the call measures the sweep's read stream and finds no surface
			static bool warned = false;
			if (!warned) fprintf(stderr, "[mc33hip] MC33_HIP_DEBUG=%u: timing experiment, every extraction returns an EMPTY surface\n", a.debug);
			warned = true;
		}
#endif
		if (c->sw.trace_file) {
			(void)hipFree(c->trace);
			c->trace = nullptr;
			c->trace_waves = ((c->ntiles + 3) / 4) * 4;
			HIP_TRY(hipMalloc(&c->trace, c->trace_waves * 32));
			HIP_TRY(hipMemsetAsync(c->trace, 0, c->trace_waves * 32, st));
			a.trace = c->trace;
			if (c->timing_level > 0) HIP_TRY(hipEventRecord(c->ev[0], st));
		}
		// nothing is cleared between extractions: headers carry the epoch, k_slots resets the counters and the
		// partial sums of the next call, the counts of the row segments carry the tag of the tail that wrote them (seg_tagged)
		L->pack = launch_sweep_ni<1>(c, a, st);
		L->tail_pending = true;
		HIP_TRY(hipGetLastError());
	}
	c->cur_lane = L;
	if (!rerun) c->lane_pretailed = tail_made;  // (a repeated count of the same call: the lane still has its own set)
	const int li = (int)(L - c->lanes);
	if (c->timing_level > 1) HIP_TRY(hipEventRecord(c->ev[1], st));
	if (tail_made) c->w = &c->ts[li];  // (made behind its sweep pass, in the lane's own set)
	else {
		// a tail made here works in the lane's own set when it has one (a repeated tail - more room for records - of a lane that had
		// been made ahead), in set 0 otherwise
		const int si = (rerun && c->lane_pretailed) ? li : 0;
		c->w = &c->ts[si];
		if (int rc = ensure_set(c, *c->w)) return rc;
		const double iso = (double)P.iso;
		if (int rc = enqueue_tail(c, &li, &si, &iso, 1, g)) return rc;
		if (rerun && c->lane_pretailed) { L->swept = true; L->tail_done = true; }  // (made again in its own set: good for further calls, as before)
	}
	if (c->timing_level > 1) HIP_TRY(hipEventRecord(c->ev[2], st));
	return 0;
}

// Sweeps for n isovalues over one range, SWEEP_MAXNI isovalues per pass over the grid; the count / extract calls that follow
// (same isovalue, same range) find their lane and go straight to the tail.  tails_ahead (mc33hip_prepare_many): the tails
// too, right behind each pass, one launch of each tail kernel per pass, every isovalue into a TailSet of its own - the calls
// then have nothing left to do but fetch the counters / emit, in any order, any number of times (when there is not enough
// device memory for the sets: sweeps only, as without the flag).
// Measured (round 4, 2048 x 2048 x 1024 ushort, 8 isovalues, profiles/r04_tails_ahead.txt): the batched tails take 0.29 ms per
// isovalue instead of 0.345 (k_cells 188 us per isovalue instead of 214, k_boundary 16 instead of 23; the scans gain nothing),
// but the vertex pass of an isovalue whose records were written eight tails ago instead of just now takes 30 % longer (avg
// 575 instead of 442 us: its per-batch chain of record and row-base fetches finds them in HBM instead of the 256 MB
// last-level cache) - 14.2 ms per step against 13.2.  So a single GPU keeps tail and emit of an isovalue adjacent
// (mc33hip_sweep_many), and the sets are for callers that need ALL counts before the first emit (z-slabs over several GPUs:
// one exchange of counts per step instead of one per isovalue).
static int enqueue_sweep_many(mc33hip_ctx *c, const double *isos, int n, bool tails_ahead) {
	hipStream_t st = c->stream;
	if (int rc = plan_sweep(c, c->P.zs, c->range.z_end)) return rc;
	SlotGeom g;
	if (int rc = slot_geometry(c, g)) return rc;
	if (c->sw.tails_ahead >= 0) tails_ahead = c->sw.tails_ahead != 0;  // (developer A/B of the two flows)
	if (tails_ahead) {
		// records to make room for in a new set: what the last extraction of this context needed
		uint64_t hint = 0;
		for (int k = 0; k < MC33_LANES; k++) hint = std::max<uint64_t>(hint, c->ts[k].records_hint);
		for (int k = 0; k < n && tails_ahead; k++)
			if (ensure_set(c, c->ts[k], hint) != 0) {  // (out of device memory: no sets beyond the first, no tails ahead)
				forget_sweeps(c);  // (first: a lane that still said "tail made" would send a later count to a set that is gone)
				for (int j = 1; j <= k; j++) free_set(c->ts[j]);
				(void)hipGetLastError();
				c->w = &c->ts[0];  // (the last count may have worked in one of the sets that are gone)
				c->counted = false;
				tails_ahead = false;
			}
		if (!tails_ahead) { if (int rc = ensure_set(c, c->ts[0])) return rc; }
	}
	int k = 0, pass = 0;
	while (k < n) {
		const int ni = (n - k >= 4) ? 4 : (n - k >= 2) ? 2 : 1;
		SweepArgs a;
		sweep_args(c, g, a);
#ifdef MC33_DEV
		a.debug = c->sw.debug;
#endif
		for (int q = 0; q < ni; q++) {
			IsoLane &L = c->lanes[k + q];
			if (int rc = begin_lane(c, L, g, st)) return rc;
			set_lane(a, q, L, isos[k + q]);
		}
		// (events around the pass are only recorded; read_timing asks for the elapsed time when the lane is consumed)
		const bool timed = c->timing_level > 0 && pass < MC33_MANY_PASSES;
		if (timed) HIP_TRY(hipEventRecord(c->ev_many[pass][0], st));
		const uint32_t pack = ni == 4 ? launch_sweep_ni<4>(c, a, st) : ni == 2 ? launch_sweep_ni<2>(c, a, st) : launch_sweep_ni<1>(c, a, st);
		HIP_TRY(hipGetLastError());
		if (timed) HIP_TRY(hipEventRecord(c->ev_many[pass][1], st));
		int idx[SWEEP_MAXNI];
		for (int q = 0; q < ni; q++) {
			IsoLane &L = c->lanes[k + q];
			L.swept = true; L.tail_pending = true; L.tail_done = false; L.iso = isos[k + q]; L.range = c->range; L.pack = pack;
			L.many_pass = timed ? pass : -1; L.many_ni = ni;
			idx[q] = k + q;
		}
		if (tails_ahead) {
			if (int rc = enqueue_tail(c, idx, idx, isos + k, ni, g)) return rc;
			for (int q = 0; q < ni; q++) c->lanes[k + q].tail_done = true;
		}
		if (timed) HIP_TRY(hipEventRecord(c->ev_many[pass][2], st));
		k += ni;
		pass++;
	}
	return 0;
}

// Host destinations of a pipelined download (mc33hip_emit_download): every array is copied on the context's copy stream as soon as
// the passes that write it have been through, while the remaining passes still run.
struct DownloadPlan { void *hV, *hN, *hT; size_t bV, bN, bT; };

static int enqueue_emit(mc33hip_ctx *c, void *dV, void *dN, void *dT, uint64_t capV, uint64_t capT, const DownloadPlan *dl = nullptr,
                        const unsigned long long *dev_base = nullptr) {
	EmitArgs a;
	a.dev_base = dev_base;
	a.c.tab.lut = c->d_lut; a.c.tab.rule_words = c->d_rules; a.c.tab.rule_index = c->d_rule_index;
	a.c.P = c->P;
	a.c.G.p = c->d_grid; a.c.G.pitch = (uint32_t)c->pitch; a.c.G.z0 = c->desc.plane0; a.c.G.slice = c->slice;
	a.c.seg_base = c->w->seg_base; a.c.seg_dir = c->w->seg_dir;
	a.c.entries_a = c->w->entries_a; a.c.entries_b = c->w->entries_b; a.c.entries_c = c->w->entries_c; a.c.fast_b = c->d_fast_b; a.c.fast_b_in_lds = false; a.c.entry_seg = c->w->entry_seg;
	a.c.V = (real_t *)dV; a.c.N = (float *)dN; a.c.Tri = (uint32_t *)dT;
	a.c.z_emit = c->range.z_begin; a.c.v_skip = a.c.t_skip = a.c.id_delta = 0;
	a.ctr = c->w->d_ctr;
	a.slow_list = c->w->slow_list;
	a.lc = c->w->lc; a.slot_base = c->w->slot_base;
	a.entry_cap = (uint32_t)c->w->entry_cap;
	a.capV = capV; a.capT = capT;
	a.ghost_segs = c->ghost_segs;
	a.id_base = c->range.id_base;
	a.batches = c->w->batches; a.batch_cap = (uint32_t)std::min<uint64_t>(c->w->batch_cap, 0xFFFFFFFFull);
	a.host_ctr = c->w->h_ctr;  // (hipHostMalloc'ed: the same address on the device)
	c->w->ctr_published = true;
	// rows may be staged in 16-byte chunks when every row of the grid starts on a 16-byte boundary (always so for the library's
	// own copy; a caller's device buffer may have any pitch: its records then load for themselves)
	a.stage_rows = ((uintptr_t)c->d_grid % 16u) == 0 && (c->pitch * sizeof(sample_t)) % 16u == 0 && (c->slice * sizeof(sample_t)) % 16u == 0 &&
	               !c->sw.no_stage;
	// The triangle pass is fastest with a thread per record (C5, 14.4 M records: 16 384 / 32 768 / 65 536 blocks 392 / 363 /
	// 352 us; C3, 3.9 M: 2 048 / 4 096 / 8 192 / 16 384 blocks 106 / 98 / 93 / 88 us).  How many records this extraction has is
	// on the device only: the grid follows the last extraction whose counters were read, 16 384 blocks at least.
	const uint32_t blocks = c->sw.emit_blocks ? c->sw.emit_blocks : std::max(256u * 64u, std::min(1u << 20, ((c->w->records_hint + 255u) / 256u + 7u) & ~7u));
	if (!c->cus) HIP_TRY(hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, c->device));
	// The three emit passes are independent (V/N vs T, fast vs slow records).  While each of them waited through a chain of
	// dependent loads (rounds 1 and most of 2) running them side by side on three streams paid on large grids (0.15 instead
	// of 0.18 ms at 768^3); with the loads of a round trip asked for together they keep the GPU busy by themselves and
	// one after the other is as fast or faster (C3 tail 0.401 against 0.404 - 0.409 ms, C5 step 15.97 against 16.18 ms), without
	// the events between the streams.  MC33_HIP_NO_FORK=0 still runs them side by side.
	// MC33_HIP_NO_FORK: 0 = all three side by side, 1 = all in sequence, unset = the two fast passes in sequence and the slow one
	// behind them (few slow records) or, on large grids, beside them on a second stream (many: see below).
	const bool fork_env = c->sw.no_fork >= 0;
	const uint64_t range_cells = (uint64_t)c->P.nx * c->P.ny * (c->range.z_end - c->P.zs);
	const bool fork_all = c->sw.no_fork == 0;
	// Which fast pass goes first (round 4).  The triangle pass lives on dependent look-ups in what the tail has just written - records,
	// directory lines, segment bases: 144 MB at 1024^3 float - and right behind the tail it finds them in the 256 MB last-level cache;
	// behind the vertex pass, which pulls 0.45 GB of sample lines through that cache, it does not: 85 -> 68 - 70 us at 1024^3 with
	// the triangles first, the vertex pass unchanged (110 - 115), the step 1.05 - 1.06 -> 1.02 - 1.03 ms.  On the 2048 x 2048 x 1024
	// ushort grid (14.4 M records per isovalue: the set does not fit either way) the order costs the triangle pass 20 - 25 us and
	// gives the vertex pass 14: the vertex pass stays first there.  MC33_HIP_TRI_FIRST=0 / 1 forces the order.
	// The slow pass.  FEW slow records - the usual case: cells on the grid's faces, a corner equal to the isovalue here and there - go
	// through k_emit_slow_slots in sequence behind the two fast passes (8 800 records of the 1024^3 cos field in 14 us with the GPU to
	// itself; a thread per record: 31).  Until round 4 the pass always ran beside the fast passes on a second stream: an event at the
	// fork and a cross-queue wait at the join (6 - 7 us each inside an emit stage of 200), and once the triangle pass went first, slow
	// blocks still resident when the vertex pass placed its own - that kernel is as many blocks as the device holds (3 per CU by its
	// LDS image) with a fixed share of the batches each, and 26 KB of a slow block in the middle of a CU's LDS kept the CU's third
	// vertex block out until one of the other two had finished: 170 instead of 112 us in a third of the calls whenever the slow pass
	// ended 0 - 3 us behind the triangle pass (profiles/r04_vertex_pass_bimodal.txt).  MANY slow records (noise, integer isovalues on
	// integer grids) - or an unknown number - take a thread each (k_emit_slow), on large grids beside the fast passes on the second
	// stream as before, the vertex pass first as before.  MC33_HIP_SLOW_SLOTS=0 / 1 forces the kernel, MC33_HIP_NO_FORK the streams.
	const bool slow_slots = c->sw.slow_slots >= 0 ? c->sw.slow_slots != 0 : (c->w->slow_hint != 0u && c->w->slow_hint <= (c->sw.slow_slots_max ? c->sw.slow_slots_max : 32768u));
	const bool fork_slow = fork_all || (!fork_env && !slow_slots && range_cells >= 300000000ull);  // (small grids: the events cost more than they gain)
	const bool tri_first = !fork_all && (c->sw.tri_first >= 0 ? c->sw.tri_first != 0 : (c->w->records_hint <= 6000000u && !fork_slow));
	hipStream_t sv = fork_all ? c->aux : c->stream, ss = fork_slow ? c->aux2 : c->stream;
	if (fork_slow) {
		HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
		if (fork_all) HIP_TRY(hipStreamWaitEvent(c->aux, c->ev_fork, 0));
		HIP_TRY(hipStreamWaitEvent(c->aux2, c->ev_fork, 0));
	}
	// Both slow kernels walk the slow list with a grid stride: any grid is right.  A thread per record: as many blocks as the last
	// extraction's slow records fill four times over, 64 at least (waves that are started beside the vertex pass only to find the list
	// exhausted cost it), 1 024 at most and while nothing is known.  A lane per slot, 16 records per block and round: as many blocks as
	// the records need and an eighth more, 1 024 at least (a block beyond the list leaves at once, and should the count have been an
	// earlier isovalue's and far too small, the kernel goes through the list with a thread per record: it needs the threads then).
	const uint32_t slow_grid = c->sw.slow_blocks ? c->sw.slow_blocks
	                           : slow_slots ? std::max(1024u, (c->w->slow_hint + c->w->slow_hint / 8u + 15u) / 16u)
	                           : c->w->slow_hint ? std::min(1024u, std::max(64u, (c->w->slow_hint + 255u) / 256u * 4u)) : 1024u;
#define MC33_LAUNCH_SLOW(st) do { if (slow_slots) hipLaunchKernelGGL(k_emit_slow_slots, dim3(slow_grid), dim3(256), 0, st, a); else hipLaunchKernelGGL(k_emit_slow, dim3(slow_grid), dim3(256), 0, st, a); } while (0)
	if (fork_slow) {  // (first: it is the one with the long chains)
		MC33_LAUNCH_SLOW(ss);
		HIP_TRY(hipEventRecord(c->ev_join2, c->aux2));
	}
	// A pipelined download wants every array complete as early as possible: the slow records - which write V, N AND T - go first
	// then (in sequence all the same: 14 us with the GPU to itself), and each fast pass is followed by the copies it completes.
	const bool dl_split = dl && !fork_all && !fork_slow;
	if (dl_split) MC33_LAUNCH_SLOW(ss);
	auto copy_T = [&]() -> int {
		HIP_TRY(hipEventRecord(c->ev_dl[0], c->stream));
		HIP_TRY(hipStreamWaitEvent(c->copy, c->ev_dl[0], 0));
		if (dl->bT) HIP_TRY(hipMemcpyAsync(dl->hT, dT, dl->bT, hipMemcpyDeviceToHost, c->copy));
		return 0;
	};
	auto copy_VN = [&]() -> int {
		HIP_TRY(hipEventRecord(c->ev_dl[1], c->stream));
		HIP_TRY(hipStreamWaitEvent(c->copy, c->ev_dl[1], 0));
		if (dl->bV) HIP_TRY(hipMemcpyAsync(dl->hV, dV, dl->bV, hipMemcpyDeviceToHost, c->copy));
		if (dl->bN) HIP_TRY(hipMemcpyAsync(dl->hN, dN, dl->bN, hipMemcpyDeviceToHost, c->copy));
		return 0;
	};
#ifdef MC33_DEV
	// developer experiment (MC33_HIP_TRI_BELOW=1): what would the triangle pass take if every record knew where its owners' records
	// are?  A first pass keeps the positions it finds through the directory (k_emit_fast_triangles<1>), a second one takes them from
	// that array with the record and never looks at the directory (<2>: same triangles) - the second is the one to time.
	static uint32_t *s_below = nullptr; static uint64_t s_below_cap = 0;
	const bool tri_below = c->sw.tri_below != 0;
	a.below_idx = nullptr;
	if (tri_below) {
		if (s_below_cap < c->w->entry_cap) { (void)hipFree(s_below); s_below = nullptr; HIP_TRY(hipMalloc(&s_below, c->w->entry_cap * 12ull)); s_below_cap = c->w->entry_cap; }
		a.below_idx = s_below;
	}
#define MC33_LAUNCH_TRI(st) do { if (tri_below) { hipLaunchKernelGGL(k_emit_fast_triangles<1>, dim3(blocks), dim3(256), 0, st, a); hipLaunchKernelGGL(k_emit_fast_triangles<2>, dim3(blocks), dim3(256), 0, st, a); } else hipLaunchKernelGGL(k_emit_fast_triangles<0>, dim3(blocks), dim3(256), 0, st, a); } while (0)
#else
#define MC33_LAUNCH_TRI(st) hipLaunchKernelGGL(k_emit_fast_triangles, dim3(blocks), dim3(256), 0, st, a)
#endif
	if (tri_first) {
		MC33_LAUNCH_TRI(sv);
		if (dl_split) { if (int rc = copy_T()) return rc; }
	}
#ifdef MC33_DEV
	if (c->sw.old_vertex_pass) hipLaunchKernelGGL(k_emit_fast_vertices, dim3(blocks), dim3(256), 0, c->stream, a);  // (the round-2 pass, for A/B timing)
	else
#endif
	{
		// as many blocks as the device holds at once (one more round of blocks would run with most of the GPU idle); every wave
		// walks many batches, its next batch's records in flight while it works on one
		if (!c->emit_v_blocks_per_cu) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->emit_v_blocks_per_cu, k_emit_vertices<3>, 256, 0));
		const uint32_t vblocks = (uint32_t)std::max(1, c->cus) * (c->sw.emit_v_blocks_per_cu ? c->sw.emit_v_blocks_per_cu : (uint32_t)std::max(1, c->emit_v_blocks_per_cu));
		const dim3 vgrid((vblocks + 7u) & ~7u);
		switch (c->P.store_mode) {
		case 0: hipLaunchKernelGGL(k_emit_vertices<0>, vgrid, dim3(256), 0, c->stream, a); break;
		case 1: hipLaunchKernelGGL(k_emit_vertices<1>, vgrid, dim3(256), 0, c->stream, a); break;
		case 2: hipLaunchKernelGGL(k_emit_vertices<2>, vgrid, dim3(256), 0, c->stream, a); break;
		default: hipLaunchKernelGGL(k_emit_vertices<3>, vgrid, dim3(256), 0, c->stream, a); break;
		}
	}
	if (dl_split) { if (int rc = copy_VN()) return rc; }
	if (!tri_first) {
		MC33_LAUNCH_TRI(sv);
		if (dl_split) { if (int rc = copy_T()) return rc; }
	}
#undef MC33_LAUNCH_TRI
	if (fork_all) HIP_TRY(hipEventRecord(c->ev_join, c->aux));
	if (!fork_slow && !dl_split) MC33_LAUNCH_SLOW(ss);
#undef MC33_LAUNCH_SLOW
	HIP_TRY(hipGetLastError());
	if (fork_all) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
	if (fork_slow) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
	if (dl && !dl_split) {  // (passes on several streams: everything behind the join)
		if (int rc = copy_T()) return rc;
		if (int rc = copy_VN()) return rc;
	}
	if (c->timing_level > 0) HIP_TRY(hipEventRecord(c->ev[3], c->stream));
	c->emit_pending = true;
	return 0;
}

static int fetch_counters(mc33hip_ctx *c) {
	// (after an emit pass the triangle kernel has already written them into h_ctr: only the wait is left)
	if (!c->w->ctr_published) HIP_TRY(hipMemcpyAsync(c->w->h_ctr, c->w->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
	c->w->ctr_published = true;  // (h_ctr matches the set's counters until the next tail into this set: enqueue_tail)
	HIP_TRY(hipStreamSynchronize(c->stream));  // (polling the stream before blocking - hipStreamQuery for up to 3 ms - gains nothing: 1.086 / 1.087 / 1.097 against 1.063 / 1.094 / 1.126 ms per step at 1024^3, round 4)
	c->w->records_hint = c->w->h_ctr->entry_cursor == 0xFFFFFFFFu ? 0u : c->w->h_ctr->entry_cursor;
	c->w->slow_hint = c->w->h_ctr->slow_cursor + 1u;
	c->w->count_known = true; c->w->count_needed = c->w->h_ctr->count_pending != 0u;
	if (c->sw.verbose)
		fprintf(stderr, "[mc33hip] cut cells %u (slow %u, dirty segments %u, record batches %u)\n", c->w->h_ctr->entry_cursor,
		        c->w->h_ctr->slow_cursor, c->w->h_ctr->dirty_cursor, c->w->h_ctr->batch_cursor);
	if (c->w->h_ctr->debug[0])
		fprintf(stderr, "[mc33hip] DEBUG words %u: first %u count %u z %u y0 %u xbase %u batch %u of %u\n", c->w->h_ctr->debug[0], c->w->h_ctr->debug[1], c->w->h_ctr->debug[2],
		        c->w->h_ctr->debug[3], c->w->h_ctr->debug[4], c->w->h_ctr->debug[5], c->w->h_ctr->debug[6], c->w->h_ctr->debug[7]);
	if (c->trace_cells && c->sw.trace_cells) {
		void *h = malloc(c->trace_cells_n * 32);
		if (h && hipMemcpy(h, c->trace_cells, c->trace_cells_n * 32, hipMemcpyDeviceToHost) == hipSuccess) {
			FILE *f = fopen(c->sw.trace_cells, "wb");
			if (f) { fwrite(h, 32, c->trace_cells_n, f); fclose(f); }
		}
		free(h);
	}
	if (c->trace && c->sw.trace_file) {  // developer tracing: per-wave stamps of the last sweep
		void *h = malloc(c->trace_waves * 32);
		if (h && hipMemcpy(h, c->trace, c->trace_waves * 32, hipMemcpyDeviceToHost) == hipSuccess) {
			FILE *f = fopen(c->sw.trace_file, "wb");
			if (f) { fwrite(h, 32, c->trace_waves, f); fclose(f); }
		}
		free(h);
	}
	return 0;
}

static int finish_counts(mc33hip_ctx *c, mc33hip_counts *out) {
	const Counters &h = *c->w->h_ctr;
	const uint64_t gV = c->ghost_segs ? h.ghostV : 0, gT = c->ghost_segs ? h.ghostT : 0;
	c->counts.nV = h.totV - gV; c->counts.nT = h.totT - gT;
	c->counts.nV_ghost = gV; c->counts.nT_ghost = gT;
	c->counts.active_cells = h.entry_cursor;
	if (out) *out = c->counts;
	if (h.totV > 0xFFFFFFFFull || h.totT > 0xFFFFFFFFull || (uint64_t)c->range.id_base + c->counts.nV > 0xFFFFFFFFull) {
		set_err("surface exceeds 2^32-1 vertices or triangles");
		return MC33HIP_EOVERFLOW;
	}
	return 0;
}

static void read_timing(mc33hip_ctx *c, bool with_emit, unsigned launches) {
	mc33hip_timing &t = c->timing;
	c->emit_pending = false;
	t.sweep_ms = t.scan_ms = t.emit_ms = t.total_ms = 0.f;
	t.sweep_launches = launches;
	if (c->timing_level > 1) {
		(void)hipEventElapsedTime(&t.sweep_ms, c->ev[0], c->ev[1]);
		(void)hipEventElapsedTime(&t.scan_ms, c->ev[1], c->ev[2]);
		if (with_emit) (void)hipEventElapsedTime(&t.emit_ms, c->ev[2], c->ev[3]);
	}
	if (c->timing_level > 0 && with_emit) (void)hipEventElapsedTime(&t.total_ms, c->ev[0], c->ev[3]);
	else if (c->timing_level > 1) (void)hipEventElapsedTime(&t.total_ms, c->ev[0], c->ev[2]);
	if (c->lane_presweeped && c->cur_lane && c->timing_level > 0 && c->cur_lane->many_pass >= 0) {
		// the sweep was made ahead of time, NI isovalues per pass: its share.  (The pass lies before this call's work on the same
		// stream, which fetch_counters has waited for: its events are complete)
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, c->ev_many[c->cur_lane->many_pass][0], c->ev_many[c->cur_lane->many_pass][1]) == hipSuccess) {
			ms /= (float)c->cur_lane->many_ni;
			t.sweep_ms += ms;
			t.total_ms += ms;
		}
		// ... and of the tails made behind that pass, one launch of each kernel for its NI isovalues
		if (c->lane_pretailed && hipEventElapsedTime(&ms, c->ev_many[c->cur_lane->many_pass][1], c->ev_many[c->cur_lane->many_pass][2]) == hipSuccess) {
			ms /= (float)c->cur_lane->many_ni;
			t.scan_ms += ms;
			t.total_ms += ms;
		}
	}
}

extern "C" int mc33hip_count(mc33hip_ctx *c, double iso, const mc33hip_range *range, mc33hip_counts *out) {
	if (!c) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = check_range(c, range))) return rc;
	c->counted = false;
	fill_params(c, iso, range);
	if ((rc = ensure_workspaces(c))) return rc;
	unsigned launches = 0;
	for (;;) {
		if ((rc = enqueue_count(c, launches > 0))) return rc;
		launches++;
		if ((rc = fetch_counters(c))) return rc;
		if (c->w->h_ctr->entry_cursor <= c->w->entry_cap) break;
		if ((rc = grow_entries(*c->w, c->w->h_ctr->entry_cursor))) return rc;
	}
	read_timing(c, false, launches);
	if ((rc = finish_counts(c, out))) return rc;
	c->counted = true;
	return MC33HIP_OK;
}

extern "C" int mc33hip_sweep_many(mc33hip_ctx *c, const double *isos, int n, const mc33hip_range *range) {
	if (!c || !isos || n < 1 || n > MC33_LANES) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = check_range(c, range))) return rc;
	c->counted = false;
	fill_params(c, isos[0], range);
	if ((rc = ensure_workspaces(c))) return rc;
	return enqueue_sweep_many(c, isos, n, false);
}

extern "C" int mc33hip_prepare_many(mc33hip_ctx *c, const double *isos, int n, const mc33hip_range *range) {
	if (!c || !isos || n < 1 || n > MC33_LANES) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = check_range(c, range))) return rc;
	c->counted = false;
	fill_params(c, isos[0], range);
	if ((rc = ensure_workspaces(c))) return rc;
	return enqueue_sweep_many(c, isos, n, true);
}

extern "C" int mc33hip_set_inclined(mc33hip_ctx *c, const double *grd_A, const double *grd_Ai, int triangular) {
	if (!c) return MC33HIP_EINVAL;
	c->inclined = grd_A && grd_Ai;
	c->triangular = triangular != 0;
	if (c->inclined) { memcpy(c->grd_A, grd_A, sizeof c->grd_A); memcpy(c->grd_Ai, grd_Ai, sizeof c->grd_Ai); }
	c->counted = false;
	return MC33HIP_OK;
}

extern "C" int mc33hip_set_normal_neg(mc33hip_ctx *c, int on) {
	if (!c) return MC33HIP_EINVAL;
	c->normal_neg = on != 0;
	c->counted = false;
	return MC33HIP_OK;
}

extern "C" int mc33hip_set_timing(mc33hip_ctx *c, int level) {
	if (!c || level < 0 || level > 2) return MC33HIP_EINVAL;
	c->timing_level = level;
	return MC33HIP_OK;
}

extern "C" int mc33hip_set_id_base(mc33hip_ctx *c, unsigned int id_base) {
	if (!c || !c->counted) return MC33HIP_EINVAL;
	if ((uint64_t)id_base + c->counts.nV > 0xFFFFFFFFull) { set_err("vertex ids exceed 2^32-1"); return MC33HIP_EOVERFLOW; }
	c->range.id_base = id_base;
	return MC33HIP_OK;
}

extern "C" int mc33hip_emit(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT) {
	if (!c || !c->counted) { set_err("mc33hip_emit needs a successful mc33hip_count first"); return MC33HIP_EINVAL; }
	if (capV < c->counts.nV || capT < c->counts.nT) { set_err("output buffers too small"); return MC33HIP_ECAPACITY; }
	if ((c->counts.nV && (!dV || !dN)) || (c->counts.nT && !dT)) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	return enqueue_emit(c, dV, dN, dT, capV, capT);
}

// ---- a z-slab's count, exchange and emit without a host round trip in between (slabs.py: extract_slab) ----
extern "C" int mc33hip_count_async(mc33hip_ctx *c, double iso, const mc33hip_range *range) {
	if (!c) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = check_range(c, range))) return rc;
	c->counted = false;
	fill_params(c, iso, range);
	if ((rc = ensure_workspaces(c))) return rc;
	if ((rc = enqueue_count(c, false))) return rc;
	memset(&c->counts, 0, sizeof c->counts);  // (not known on the host until mc33hip_count_finish)
	c->counted = true;
	c->async_count = true;
	return MC33HIP_OK;
}

extern "C" int mc33hip_counts_to_device(mc33hip_ctx *c, long long *device_dst) {
	if (!c || !c->counted || !device_dst) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(1), 0, c->stream, (const Counters *)c->w->d_ctr, c->ghost_segs, device_dst);
	HIP_TRY(hipGetLastError());
	return MC33HIP_OK;
}

extern "C" int mc33hip_bases_from_table(mc33hip_ctx *c, const long long *device_table, int stride, int rank, int concatenated) {
	if (!c || !device_table || stride < 2 || rank < 0) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if (!c->d_bases) HIP_TRY(hipMalloc(&c->d_bases, 3 * sizeof(unsigned long long)));
	hipLaunchKernelGGL(k_slab_bases, dim3(1), dim3(1), 0, c->stream, device_table, stride, rank, concatenated, c->d_bases);
	HIP_TRY(hipGetLastError());
	return MC33HIP_OK;
}

extern "C" int mc33hip_emit_at_device_bases(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT) {
	if (!c || !c->counted || !c->d_bases) { set_err("mc33hip_emit_at_device_bases needs a count and mc33hip_bases_from_table first"); return MC33HIP_EINVAL; }
	if (!dV || !dN || !dT) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	return enqueue_emit(c, dV, dN, dT, capV, capT, nullptr, c->d_bases);  // (capacities are checked on the device: emit_prepare)
}

extern "C" int mc33hip_count_finish(mc33hip_ctx *c, mc33hip_counts *out) {
	if (!c || !c->counted) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = fetch_counters(c))) return rc;
	c->async_count = false;
	if (c->w->h_ctr->entry_cursor > c->w->entry_cap) {  // the records did not fit: nothing was emitted; the synchronous path makes room
		const uint32_t need = c->w->h_ctr->entry_cursor;
		c->counted = false;
		if ((rc = grow_entries(*c->w, need))) return rc;
		set_err("work records did not fit (%u): repeat with mc33hip_count", need);
		return MC33HIP_ECAPACITY;
	}
	read_timing(c, false, 1);
	if ((rc = finish_counts(c, out))) return rc;
	if (c->w->h_ctr->emit_skipped) { set_err("output buffers too small: need %llu vertices, %llu triangles", c->counts.nV, c->counts.nT); return MC33HIP_ECAPACITY; }
	return MC33HIP_OK;
}

extern "C" int mc33hip_emit_download(mc33hip_ctx *c, void *dV, void *dN, void *dT, unsigned long long capV, unsigned long long capT,
                                     void *hV, void *hN, void *hT) {
	if (!c || !c->counted) { set_err("mc33hip_emit_download needs a successful mc33hip_count first"); return MC33HIP_EINVAL; }
	if (capV < c->counts.nV || capT < c->counts.nT) { set_err("output buffers too small"); return MC33HIP_ECAPACITY; }
	if ((c->counts.nV && (!dV || !dN || !hV || !hN)) || (c->counts.nT && (!dT || !hT))) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	const DownloadPlan dl{hV, hN, hT, (size_t)c->counts.nV * 3 * sizeof(real_t), (size_t)c->counts.nV * 12, (size_t)c->counts.nT * 12};
	return enqueue_emit(c, dV, dN, dT, capV, capT, &dl);
}

extern "C" int mc33hip_download_wait(mc33hip_ctx *c) {
	if (!c) return MC33HIP_EINVAL;
	if (hipSetDevice(c->device) != hipSuccess) return MC33HIP_ERUNTIME;
	HIP_TRY(hipStreamSynchronize(c->copy));    // (ordered behind the passes by the events: the arrays are complete and on the host)
	HIP_TRY(hipStreamSynchronize(c->stream));  // (... and nothing of the emit is left running when the caller gets its surface)
	return MC33HIP_OK;
}

extern "C" int mc33hip_extract(mc33hip_ctx *c, double iso, const mc33hip_range *range, void *dV, void *dN, void *dT,
                               unsigned long long capV, unsigned long long capT, mc33hip_counts *out) {
	if (!c) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if ((rc = check_range(c, range))) return rc;
	c->counted = false;
	fill_params(c, iso, range);
	if ((rc = ensure_workspaces(c))) return rc;
	unsigned launches = 0;
	for (;;) {
		if ((rc = enqueue_count(c, launches > 0))) return rc;
		launches++;
		if ((rc = enqueue_emit(c, dV, dN, dT, capV, capT))) return rc;  // checks capacities on the device
		if ((rc = fetch_counters(c))) return rc;
		if (c->w->h_ctr->entry_cursor <= c->w->entry_cap) break;
		if ((rc = grow_entries(*c->w, c->w->h_ctr->entry_cursor))) return rc;
	}
	read_timing(c, true, launches);
	if ((rc = finish_counts(c, out))) return rc;
	c->counted = true;
	if (c->w->h_ctr->emit_skipped) { set_err("output buffers too small: need %llu vertices, %llu triangles", c->counts.nV, c->counts.nT); return MC33HIP_ECAPACITY; }
	return MC33HIP_OK;
}

extern "C" int mc33hip_last_timing(mc33hip_ctx *c, mc33hip_timing *t) {
	if (!c || !t) return MC33HIP_EINVAL;
	if (c->timing_level == 0) {  // no events were recorded for the last call: zeros, not an older call's numbers
		c->emit_pending = false;
		c->timing.sweep_ms = c->timing.scan_ms = c->timing.emit_ms = c->timing.total_ms = 0.f;
	}
	if (c->emit_pending) {  // a separate mc33hip_emit: wait for it and add its time
		if (hipEventSynchronize(c->ev[3]) == hipSuccess) {
			(void)hipEventElapsedTime(&c->timing.emit_ms, c->ev[2], c->ev[3]);
			(void)hipEventElapsedTime(&c->timing.total_ms, c->ev[0], c->ev[3]);
		}
		c->emit_pending = false;
	}
	*t = c->timing;
	return MC33HIP_OK;
}

// ---------------------------------------------------------------------------------------------------
// mc33hip_probe_read: what a plain read of the resident grid reaches on this device, in this process, on this buffer -
// the ceiling the sweep's `roofline.frac` is set beside ("of peak" and "of a measured read ceiling").
// Every 16-byte chunk once, nontemporal, nothing written.  The launch shape is the best of tools/read_ceiling_probe.hip
// (profiles/r04_read_ceiling_probe.txt): every block a CONTIGUOUS piece of the buffer, 16 loads in flight per lane - 7.1 - 7.2
// TB/s at any occupancy on a 4 GiB buffer, where a grid-stride loop over 8 blocks per CU (the first form of this probe) reaches
// 6.2 - 6.4 and flattered the sweep.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_read(const u32x4_t *p, uint64_t n16, uint32_t *sink) {
	constexpr int U = 16;
	const uint64_t per = ((n16 + gridDim.x - 1) / gridDim.x + 255u) & ~(uint64_t)255, lo = (uint64_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
	u32x4_t acc = {0u, 0u, 0u, 0u};
	uint64_t i = lo + threadIdx.x;
	for (; i + (U - 1) * 256u < hi; i += U * 256u) {
		u32x4_t v[U];
#pragma unroll
		for (int k = 0; k < U; k++) v[k] = __builtin_nontemporal_load(p + i + k * 256u);
#pragma unroll
		for (int k = 0; k < U; k++) acc ^= v[k];
	}
	for (; i < hi; i += 256u) acc ^= __builtin_nontemporal_load(p + i);
	if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && sink) atomicAdd(sink, 1u);  // (keeps the loads; the word is as likely as any other)
}

extern "C" int mc33hip_probe_read(mc33hip_ctx *c, int reps, float *ms_best, float *ms_median, unsigned long long *bytes) {
	if (!c || reps < 1 || reps > 64 || !c->d_grid) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	// (a grid the context does not own may be a strided view into a larger allocation: its last plane ends with its last row)
	const uint64_t nsamples = c->owns_grid ? (uint64_t)c->slice * c->desc.npz_resident
	                                       : (uint64_t)c->slice * (c->desc.npz_resident - 1u) + (uint64_t)c->pitch * (c->desc.npy - 1u) + c->desc.npx;
	const uint64_t nbytes = (nsamples * sizeof(sample_t)) & ~(uint64_t)15;
	if (nbytes < 64) return MC33HIP_EINVAL;
	const u32x4_t *p = (const u32x4_t *)(((uintptr_t)c->d_grid + 15u) & ~(uintptr_t)15);
	const uint64_t n16 = (nbytes - ((uintptr_t)p - (uintptr_t)c->d_grid)) / 16;
	if (!c->cus) HIP_TRY(hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, c->device));
	hipEvent_t e0, e1;
	HIP_TRY(hipEventCreate(&e0));
	if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_err("hipEventCreate failed"); return MC33HIP_ERUNTIME; }
	std::vector<float> t;
	for (int k = 0; k < reps + 1; k++) {  // (the first launch is a warm-up)
		(void)hipEventRecord(e0, c->stream);
		hipLaunchKernelGGL(k_probe_read, dim3((uint32_t)std::max(1, c->cus) * 4u), dim3(256), 0, c->stream, p, n16, (uint32_t *)nullptr);
		(void)hipEventRecord(e1, c->stream);
		if (hipEventSynchronize(e1) != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); set_err("k_probe_read failed"); return MC33HIP_ERUNTIME; }
		float ms = 0.f;
		(void)hipEventElapsedTime(&ms, e0, e1);
		if (k) t.push_back(ms);
	}
	(void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
	std::sort(t.begin(), t.end());
	if (ms_best) *ms_best = t.front();
	if (ms_median) *ms_median = t[t.size() / 2];
	if (bytes) *bytes = n16 * 16ull;
	return MC33HIP_OK;
}

#ifdef MC33_DEV
extern "C" int mc33hip_debug_words(mc33hip_ctx *c, unsigned int *out /*[8]*/) {  // what a guarded kernel found wrong (developer builds)
	if (!c || !out) return MC33HIP_EINVAL;
	HIP_TRY(hipStreamSynchronize(c->stream));
	HIP_TRY(hipMemcpy(out, (const char *)c->w->d_ctr + offsetof(Counters, debug), 8 * sizeof(uint32_t), hipMemcpyDeviceToHost));
	return MC33HIP_OK;
}
#endif

extern "C" int mc33hip_synchronize(mc33hip_ctx *c) {
	if (!c) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MC33HIP_OK;
}

extern "C" int mc33hip_download_many(mc33hip_ctx *c, int n, void *const *dst, const void *const *src, const size_t *bytes, int concurrent) {
	if (!c || n < 0 || (n && (!dst || !src || !bytes))) return MC33HIP_EINVAL;
	if (hipSetDevice(c->device) != hipSuccess) return MC33HIP_ERUNTIME;  // (the concurrent form may come from another thread)
	hipStream_t st = concurrent ? c->copy : c->stream;
	for (int k = 0; k < n; k++) {
		if (!bytes[k]) continue;
		if (!dst[k] || !src[k]) return MC33HIP_EINVAL;
		if (hipMemcpyAsync(dst[k], src[k], bytes[k], hipMemcpyDeviceToHost, st) != hipSuccess) { (void)hipStreamSynchronize(st); return MC33HIP_ERUNTIME; }
	}
	return hipStreamSynchronize(st) == hipSuccess ? MC33HIP_OK : MC33HIP_ERUNTIME;
}

extern "C" int mc33hip_download_concurrent(mc33hip_ctx *c, void *dst, const void *src, size_t bytes) {
	if (!c || (bytes && (!dst || !src))) return MC33HIP_EINVAL;
	if (!bytes) return MC33HIP_OK;
	if (hipSetDevice(c->device) != hipSuccess) return MC33HIP_ERUNTIME;  // (may be another thread than the context's)
	if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->copy) != hipSuccess) return MC33HIP_ERUNTIME;
	return hipStreamSynchronize(c->copy) == hipSuccess ? MC33HIP_OK : MC33HIP_ERUNTIME;
}

extern "C" int mc33hip_download(mc33hip_ctx *c, void *dst, const void *src, size_t bytes) {
	if (!c || (bytes && (!dst || !src))) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	if (!bytes) return MC33HIP_OK;
	HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MC33HIP_OK;
}

extern "C" int mc33hip_device_alloc(mc33hip_ctx *c, void **dptr, size_t bytes) {
	if (!c || !dptr) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	*dptr = nullptr;
	HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
	return MC33HIP_OK;
}

extern "C" int mc33hip_device_free(mc33hip_ctx *c, void *dptr) {
	if (!c) return MC33HIP_EINVAL;
	int rc = use_device(c);
	if (rc) return rc;
	HIP_TRY(hipFree(dptr));
	return MC33HIP_OK;
}
