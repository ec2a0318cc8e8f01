// mc33_extract.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// host side, part 2: planning and enqueueing the passes, the C ABI entry points of include/mc33_hip.h.

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
	c->count_unused = false;  // (the count has served an emit: the next count of this isovalue is made anew)
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
	// The same isovalue over the same range of the library's OWN copy of the grid, counted by the call before and not emitted yet
	// (size_of_isosurface and then calculate_isosurface of the value it asked about: what a viewer does to show the memory a
	// surface will take, reference MC:1892-1940 / 1816-1889): everything the count made is still there - nothing is streamed
	// again.  Only a COUNT is reused, once: an extraction repeated with the same isovalue does all its work again, like the
	// reference's.  Not for a caller's device buffer (mc33hip_adopt_device): its samples may have been rewritten without this
	// library knowing.
	if (c->counted && c->count_unused && !c->async_count && c->owns_grid && c->timing_level == 0 && same_bits((real_t)iso, c->P.iso) && same_range(*range, c->range)) {
		c->range.id_base = range->id_base;
		return finish_counts(c, out);
	}
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
	c->count_unused = true;
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
