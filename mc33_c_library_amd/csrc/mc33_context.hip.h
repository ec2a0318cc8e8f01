// mc33_context.hip.h -- part of the ONE translation unit mc33_kernels.hip (included there, in order; not a header to include elsewhere):
// host side, part 1: errors, the context (mc33hip_ctx), environment switches, create / destroy, grid upload.

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
	bool count_unused;        // the last mc33hip_count has not served an emit yet: a count of the same isovalue and range may reuse it
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

// Rows packed into the pitched device layout through pinned staging buffers, one group of planes at a time.  row(k, j): host
// address of row j of resident plane k.  What it costs is the packing - a memcpy per row, a million scattered 4 KiB rows for a
// 1024^3 float grid made by alloc_F (every row an allocation of its own: what generate_grid_from_fn and the reference's file
// readers make, MC33_util_grd.c:147-169): 240 ms on one core, whether or not the copies over the link (77 ms) run beside it - so
// large grids are packed by up to four threads, each with two buffers of its own in turn (a group is packed while the thread's
// group before it goes over the link) and every fourth group of planes: profiles/r05_upload_rows.txt.
template <typename RowFn>
static int upload_staged(mc33hip_ctx *c, RowFn row) {
	const uint32_t npy = c->desc.npy, npz = c->desc.npz_resident;
	const size_t rowb = (size_t)c->desc.npx * sizeof(sample_t);
	const size_t planeb = c->slice * sizeof(sample_t);
	size_t planes_per = (16u << 20) / planeb;
	if (planes_per < 1) planes_per = 1;
	if (planes_per > npz) planes_per = npz;
	const uint32_t ngroups = (uint32_t)((npz + planes_per - 1) / planes_per);
	const unsigned hw = std::thread::hardware_concurrency();
	const unsigned nthreads = (uint64_t)planeb * npz < (256ull << 20) ? 1u : std::min(std::min(4u, std::max(1u, hw / 2u)), ngroups);
	HIP_TRY(hipStreamSynchronize(c->stream));  // (nothing enqueued earlier may still be reading the grid: the copies run on a stream of their own)
	std::vector<hipError_t> err(nthreads, hipSuccess);
	auto worker = [&](unsigned t) {
		hipError_t e = hipSetDevice(c->device);
		char *stage[2] = {nullptr, nullptr};
		hipEvent_t done[2] = {nullptr, nullptr};
		const bool two = ngroups > nthreads;  // (a thread with one group needs one buffer)
		if (e == hipSuccess) e = hipHostMalloc(&stage[0], planes_per * planeb, hipHostMallocDefault);
		if (e == hipSuccess && two) e = hipHostMalloc(&stage[1], planes_per * planeb, hipHostMallocDefault);
		for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
		uint32_t turn = 0;
		for (uint32_t g = t; g < ngroups && e == hipSuccess; g += nthreads, turn++) {
			const uint32_t b = two ? (turn & 1u) : 0u, k0 = g * (uint32_t)planes_per;
			const uint32_t kn = (uint32_t)((k0 + planes_per <= npz) ? planes_per : npz - k0);
			if (turn >= (two ? 2u : 1u) && (e = hipEventSynchronize(done[b])) != hipSuccess) break;  // (the buffer's last copy has left it)
			for (uint32_t k = 0; k < kn; k++)
				for (uint32_t j = 0; j < npy; j++)
					memcpy(stage[b] + k * planeb + (size_t)j * c->pitch * sizeof(sample_t), row(k0 + k, j), rowb);
			if ((e = hipMemcpyAsync((char *)c->d_grid + (size_t)k0 * planeb, stage[b], (size_t)kn * planeb, hipMemcpyHostToDevice, c->copy)) == hipSuccess)
				e = hipEventRecord(done[b], c->copy);
		}
		for (int b = 0; b < 2; b++) {  // (its own copies done before its buffers go)
			if (done[b]) { if (e == hipSuccess && turn > (uint32_t)b) e = hipEventSynchronize(done[b]); (void)hipEventDestroy(done[b]); }
		}
		if (e != hipSuccess) (void)hipStreamSynchronize(c->copy);
		for (int b = 0; b < 2; b++)
			if (stage[b]) (void)hipHostFree(stage[b]);
		err[t] = e;
	};
	std::vector<std::thread> th;
	std::vector<unsigned> inline_share;  // (a thread that could not be started: its groups on this thread)
	for (unsigned t = 1; t < nthreads; t++) {
		try { th.emplace_back(worker, t); } catch (...) { inline_share.push_back(t); }
	}
	worker(0u);
	for (unsigned t : inline_share) worker(t);
	for (auto &x : th) x.join();
	(void)hipSetDevice(c->device);
	HIP_TRY(hipStreamSynchronize(c->copy));
	for (unsigned t = 0; t < nthreads; t++)
		if (err[t] != hipSuccess) { set_err("grid upload failed: %s", hipGetErrorString(err[t])); return err[t] == hipErrorOutOfMemory ? MC33HIP_ENOMEM : MC33HIP_ERUNTIME; }
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
