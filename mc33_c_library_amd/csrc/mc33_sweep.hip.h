// mc33_sweep.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// k_sweep (the one pass over the volume) and k_boundary (the slices between its tiles).

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
#ifndef MC33_SWEEP_DEFER_N
#define MC33_SWEEP_DEFER_N 0  // (1: the hand-over of the passes over several isovalues through a log in LDS too - SweepLogN; round 5, measured, off)
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
	constexpr bool DEFER_N = NI >= 2 && MC33_SWEEP_DEFER_N;
	__shared__ typename std::conditional<DEFER_N, SweepLogN, uint32_t>::type s_logn[DEFER_N ? 4 : 1];
	uint32_t logn_np = 0, logn_ns = 0;
#if defined(MC33_DEV) && defined(MC33_SWEEP_LDS_PAD)  // (developer A/B: the LDS of a log without the log - what the lost block per CU costs by itself)
	__shared__ uint32_t s_pad[NI >= 4 ? MC33_SWEEP_LDS_PAD / 4 : 1];
	if (a.ntiles == 0xFFFFFFFFu) s_pad[threadIdx.x & 0u] = 1u;
#endif
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
	// ---- the same for the passes over several isovalues (DEFER_N; see SweepLogN) ----
	auto logn_flush = [&](bool at_end = false) __attribute__((always_inline)) {
		if constexpr (DEFER_N) {
#if defined(MC33_DEV) && defined(MC33_LOGN_DROP)  // (developer timing experiment: a log that is full is dropped, only the tile's end writes - results wrong)
			if (!at_end) { logn_np = 0; logn_ns = 0; return; }
#endif
			SweepLogN &G = s_logn[wv];
			const uint32_t ln = fresh_lane();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			for (uint32_t k = 0; k < logn_np; k++) {  // wave-uniform
				const uint64_t dst = readlane64(G.plane_dst[k], 0), fdst = readlane64(G.fmt_dst[k], 0);
				__builtin_amdgcn_raw_buffer_store_b32(G.plane[k][ln], record_rsrc((const void *)dst, 256u), ln * 4u, 0u, 0);
				if (ln == 0) *(uint8_t *)fdst = (uint8_t)PLANE_COMPACT;
			}
			if (ln < logn_ns) {
				const uint32_t *w = G.hdr[ln];
				uint32_t *h = (uint32_t *)G.hdr_dst[ln];
				*(uint4 *)h = uint4{w[0], w[1], w[2], w[3]};
				*(uint4 *)(h + 4) = uint4{w[4], w[5], w[6], w[7]};
				*(uint2 *)(h + 8) = uint2{w[8], w[9]};
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the log is written again)
			logn_np = 0; logn_ns = 0;
		}
	};
	auto logn_plane = [&](const SweepLane &L, uint64_t slot, const uint64_t (&w)[4], uint32_t ln) __attribute__((always_inline)) {
		if constexpr (DEFER_N) {
			uint32_t desc;
			if (encode_plane<S>(w, desc)) {  // (wave-uniform)
				if (logn_np == LOGN_PLANES) logn_flush();
				SweepLogN &G = s_logn[wv];
				G.plane[logn_np][ln] = desc;
				if (ln == 0) { G.plane_dst[logn_np] = (uint64_t)(L.slice_compact + slot * 64u); G.fmt_dst[logn_np] = (uint64_t)(L.plane_fmt + slot); }
				logn_np++;
			} else {
				store_plane_raw<S>(L.slice_bits + slot * 128u, w, ln);
				if (ln == 0) L.plane_fmt[slot] = (uint8_t)PLANE_RAW;
			}
		}
	};
	// (the partial sums per chunk of slots as in hand_over_slice: one atomic when the wave's slices leave a chunk)
	auto logn_header = [&](const SweepLane &L, uint64_t slot, uint64_t bp, uint64_t bc, uint64_t zrows, uint64_t zcols, const uint64_t (&act)[4], uint32_t ln,
	                       uint32_t &pchunk, unsigned long long &psum) __attribute__((always_inline)) {
		if constexpr (DEFER_N) {
			uint32_t ncell = __popcll(act[0]) + __popcll(act[1]) + __popcll(act[2]) + __popcll(act[3]);
#pragma unroll
			for (int dlt = 32; dlt; dlt >>= 1) ncell += __shfl_xor(ncell, dlt);
			if (logn_ns == LOGN_SLICES) logn_flush();
			if (ln == 0) {
				SweepLogN &G = s_logn[wv];
				uint32_t *w = G.hdr[logn_ns];
				w[0] = L.epoch << 2 | SLICE_VALID | (zrows ? SLICE_HAS_ISO : 0u);
				w[1] = (uint32_t)bp; w[2] = (uint32_t)(bp >> 32); w[3] = (uint32_t)bc; w[4] = (uint32_t)(bc >> 32); w[5] = ncell;
				w[6] = (uint32_t)zrows; w[7] = (uint32_t)(zrows >> 32); w[8] = (uint32_t)zcols; w[9] = (uint32_t)(zcols >> 32);
				G.hdr_dst[logn_ns] = (uint64_t)(L.slice_hdr + slot);
			}
			logn_ns++;
			const uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(slot / SLOT_CHUNK));
			const uint32_t nc = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncell);
			if (pchunk != chunk) {
				if (psum && ln == 0) atomicAdd(L.slot_part + pchunk, psum);
				pchunk = chunk; psum = 0ull;
			}
			psum += (unsigned long long)((nc + 63u) >> 6) << 32 | nc;
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
						} else if (DEFER_N && !MC33_DEBUG_BITS(a)) {
							const uint64_t slot = slice_slot(p - 1 - P.zs, yt, seg, a.sd), slot_up = slice_slot(p - P.zs, yt, seg, a.sd);
							if (!prev_written[q]) logn_plane(L, slot, pq, lp);
							logn_plane(L, slot_up, cur[q], lp);
							logn_header(L, slot, __ballot(ph != 0), __ballot(cur_h[q] != 0), pz | cur_z[q], pzc | cur_zc[q], act, lp, pend_chunk[q], pend_sum[q]);
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
	logn_flush(true);
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
