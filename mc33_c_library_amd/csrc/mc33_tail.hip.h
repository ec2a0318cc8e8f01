// mc33_tail.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// the tail of an extraction: k_slots, k_cells, the slow-record kernels, the prefix sums over the row segments.

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
