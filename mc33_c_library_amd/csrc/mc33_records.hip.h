// mc33_records.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// device-side records of the passes: counters, slice headers and slots, the sweep's per-isovalue buffers, bit-row layouts, plane records, the sweep's log.

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

// The same for the passes over several isovalues (MC33_SWEEP_DEFER_N, round 5): one log per wave for all its isovalue lanes, so an
// entry carries where it goes.  Small - 12 compact planes, 12 headers, 3.8 KB per wave - because the 4-isovalue forms keep the bit
// rows of the plane below in LDS already (32 KB per block) and the log costs them the fourth block per CU.
constexpr uint32_t LOGN_PLANES = 12, LOGN_SLICES = 12;
struct SweepLogN {  // per wave
	uint32_t plane[LOGN_PLANES][64];  // compact records: dword r = row r
	uint64_t plane_dst[LOGN_PLANES];  // the record's place in its lane's slice_compact ...
	uint64_t fmt_dst[LOGN_PLANES];    // ... and its byte in plane_fmt
	uint32_t hdr[LOGN_SLICES][10];    // the ten words of a SliceHeader
	uint64_t hdr_dst[LOGN_SLICES];
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
