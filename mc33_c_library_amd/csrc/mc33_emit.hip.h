// mc33_emit.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// the emit passes: k_emit_vertices, k_emit_fast_triangles, k_emit_slow_slots / k_emit_slow, and the small kernels of the device-side count exchange.

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
constexpr uint32_t TS = MC33_EV_RING ? 1u : 0u;  // (the ring's tags exist only with the ring: every index into them goes through this)
struct EmitVLds {                        // per wave
	uint32_t rowA[64], rowB[64];         // cell rows 0..62 of the tile: first / last record of the batch in that row, lane << 8 | x in the segment
	uint32_t rowvb[64];                  // cell rows: id of the first vertex of the row segment (seg_base)
	uint32_t rowinfo[EV_ROWS + 1];       // sample rows: staged << 31 | chunks - 1 << 16 | first chunk - chunk of the segment's first sample
	uint32_t vlist[256];                 // vertices of the batch: record (lane) | kind << 8, by kind
	uint32_t tag[MC33_EV_RING ? 3 : 1][MC33_EV_RING ? EV_ROWS + 1 : 1];  // (ring only) what slot s holds of sample row r: valid << 31 | chunks - 1 << 16 | first chunk (as rowinfo)
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
					if (MC33_EV_RING) { have0[g] = L.tag[(s0) * TS][(rr) * TS]; have1[g] = L.tag[(s1) * TS][(rr) * TS]; have2[g] = L.tag[(s2) * TS][(rr) * TS]; }
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
					if (got & 1u) { L.data[s0 * NITEM + it] = q0; if (MC33_EV_RING && ck == 0u) L.tag[(s0) * TS][(r) * TS] = info[g]; }
					if (got & 2u) { L.data[s1 * NITEM + it] = q1; if (MC33_EV_RING && ck == 0u) L.tag[(s1) * TS][(r) * TS] = info[g]; }
					if (got & 4u) { L.data[s2 * NITEM + it] = q2; if (MC33_EV_RING && ck == 0u) L.tag[(s2) * TS][(r) * TS] = info[g]; }
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
							return sl * (EV_ROWS * EV_W * 16u) + (rw * EV_W - (L.tag[(sl) * TS][(rw) * TS] & 0xFFFFu)) * 16u + xb0;
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
