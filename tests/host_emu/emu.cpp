// tests/host_emu/emu.cpp -- TEST INFRASTRUCTURE ONLY.
// Runs the MI355X path's per-cell formulation (mc33_c_library_amd/csrc/mc33_cell.h: plans, owner lookups,
// entry records, segment scans) serially on the CPU, so that the logic the HIP kernels execute can be
// checked against the oracle in a container without a GPU.  Never linked into the product library.
#include <cstdlib>
#include <cstring>
#include <vector>

// resident-plane window of the emulated rank; reads outside it are counted (and served from the full
// array, so the run completes and the test can report the violation instead of crashing)
static unsigned g_plane_lo = 0, g_plane_hi = 0xFFFFFFFFu, g_px = 0xFFFFFFFFu, g_py = 0xFFFFFFFFu;
static unsigned long long g_violations = 0;
#define MC33_BOUNDS_HOOK(x, y, z) \
	do { if ((z) < g_plane_lo || (z) > g_plane_hi || (x) >= g_px || (y) >= g_py) g_violations++; } while (0)
#include "../../mc33_c_library_amd/csrc/mc33_cell.h"
#include "../../mc33_c_library_amd/csrc/mc33_lut_data.h"
#include "../../mc33_c_library_amd/csrc/mc33_rules_data.h"

using namespace mc33;

struct emu_surface {
	uint32_t nV, nT;
	float *V, *N;
	uint32_t *T;
};

struct emu_slab {
	uint32_t z_begin, z_end, ghost, id_base;  // as mc33hip_range
	uint32_t plane_lo, plane_hi;              // planes the emulated rank holds
};

template <typename T>
static int run(const T *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0, const double *d, float iso,
               emu_surface *out, const emu_slab *slab = nullptr) {
	memset(out, 0, sizeof *out);
	Params P{};
	P.nx = npx - 1; P.ny = npy - 1; P.nz = npz - 1;
	P.nseg = (P.nx + SEG_CELLS - 1) / SEG_CELLS;
	P.iso = iso; P.zs = 0;
	P.negzero_iso = (iso == 0 && sign_of(P.iso)) ? 1 : 0;
	uint32_t z_emit = 0, ze = P.nz, id_base = 0;
	g_plane_lo = 0; g_plane_hi = npz - 1; g_px = npx; g_py = npy;
	if (slab) {
		z_emit = slab->z_begin; ze = slab->z_end; id_base = slab->id_base;
		P.zs = slab->z_begin - (slab->ghost ? 1 : 0);
		g_plane_lo = slab->plane_lo; g_plane_hi = slab->plane_hi;
	}
	if (d[0] != d[1] || d[1] != d[2]) { P.store_mode = 2; P.ca = (float)(d[2] / d[0]); P.cb = (float)(d[2] / d[1]); }
	else { P.store_mode = (d[0] == 1 && r0[0] == 0 && r0[1] == 0 && r0[2] == 0) ? 0 : 1; P.ca = P.cb = 1; }
	for (int k = 0; k < 3; k++) { P.O[k] = (float)r0[k]; P.D[k] = (float)d[k]; }
	GridView<T> G{data, npx, 0, (uint64_t)npx * npy};
	Tables tab{mc33_lut, mc33_rule_words, &mc33_rule_index[0][0]};

	const uint64_t nsegs = (uint64_t)(ze - P.zs) * P.ny * P.nseg;
	std::vector<uint32_t> seg_cnt(nsegs, 0);
	std::vector<SegDir> seg_dir(nsegs, SegDir{});
	std::vector<uint32_t> seg_first(nsegs, 0), seg_nent(nsegs, 0);
	std::vector<SegBase> seg_base(nsegs + 1);  // (+ 1: the triangle pass reads bases in pairs)
	std::vector<uint64_t> seg_mask(4 * nsegs, 0ull);
	uint32_t fast[256];
	build_fast_table(mc33_lut, fast);
	constexpr uint32_t lut_n = sizeof mc33_lut / sizeof mc33_lut[0];
	static uint32_t pat_info[lut_n];
	build_pattern_info(mc33_lut, lut_n, pat_info);
	const char *force = getenv("MC33_EMU_FORCE_SLOW");  // "all": no fast path; "odd": cells with odd x go slow
	const int force_mode = !force ? 0 : (force[0] == 'a' ? 1 : 2);
	// The records as the kernels keep them: half A for all, half B for tested and slow records (fast ones: from the table),
	// part C (the stored plan) for slow ones.
	std::vector<EntryA> ea;
	std::vector<EntryB> eb;
	std::vector<EntryC> ec;
	std::vector<uint32_t> entry_seg;
	EntryB fast_b[256];
	fast_b_table(fast, fast_b);
	float vbuf[8], wbuf[8];
	uint32_t idbuf[13];
	VRef v{vbuf, 1}, w{wbuf, 1};
	URef ids{idbuf, 1};
	// pass 1 (k_cells + k_slow_plan): one record per cut cell, in sweep order; slow cells are planned, their plan stored
	for (uint32_t z = P.zs; z < ze; z++)
		for (uint32_t y = 0; y < P.ny; y++)
			for (uint32_t x = 0; x < P.nx; x++) {
				const uint32_t i = load_cell(G, iso, x, y, z, v);
				if (i == 0 || i == 0xFF) continue;
				const uint64_t s = segment_index(P, x, y, z);
				if (seg_nent[s] == 0) seg_first[s] = (uint32_t)ea.size();
				seg_mask[4ull * s + ((x % SEG_CELLS) >> 6)] |= 1ull << (x & 63u);
				bool zero = false;
				for (int k = 0; k < 8; k++) zero |= v[k] == 0;
				bool is_fast = x && y && z && fast[i] != FAST_NONE && !zero;
				if (force_mode == 1 || (force_mode == 2 && (x & 1))) is_fast = false;
				Entry en;
				EntryC pc{{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu}, 0xDEADBEEFu};
				if (is_fast) en = make_fast_entry(x % SEG_CELLS, i, fast[i], 0, 0);  // as k_cells: everything from the sign index
				else {
					CellPlan p;
					plan_cell(p, tab, P, G, x, y, z, i, v);
					en = make_entry(x % SEG_CELLS, i, p, p.ntri, 0, 0, true);
					if (force_mode != 1 && cell_is_tested(p, x, y, z)) {  // as k_cells / k_slow_plan: the fast emit writes it
						en.w3 ^= ENTRY_SLOW | ENTRY_TESTED;
						// k_cells makes the same record from the pattern offset and the pattern-info table alone
						const Entry t = make_tested_entry(x % SEG_CELLS, i, p.poff, pat_info[p.poff], 0, 0);
						if (t.w0 != en.w0 || t.w1 != en.w1 || t.w2 != en.w2 || t.w3 != en.w3) return -8;
						Corner8 c8;
						for (int k = 0; k < 8; k++) c8.a[k] = v[k];
						uint32_t m8, n8;
						if (pattern_offset(tab.lut, i, c8, m8, n8) != p.poff) return -9;
					} else {
						if (p.zmask && z >= z_emit) en.w3 |= ENTRY_COUNT;
						pc = entry_c(p);
						// the stored plan gives the plan back
						CellPlan q;
						plan_restore(q, tab.lut, en, pc);
						plan_restore_points(q, v);
						if (q.rank != p.rank || q.visited != p.visited || q.created != p.created || q.onpoint != p.onpoint || q.onb != p.onb ||
						    q.tgt[0] != p.tgt[0] || q.tgt[1] != p.tgt[1] || q.tgt[2] != p.tgt[2] || q.poff != p.poff || q.m != p.m || q.n != p.n ||
						    q.nnew != p.nnew || q.zmask != p.zmask)
							return -10;
					}
				}
				ea.push_back(entry_a(en));
				eb.push_back((en.w3 & (ENTRY_SLOW | ENTRY_TESTED)) ? entry_b(en) : EntryB{0xDEADBEEFu, 0xDEADBEEFu});
				ec.push_back(pc);
				const Entry back = load_entry(ea.data(), eb.data(), fast_b, (uint32_t)(ea.size() - 1));
				if (back.w0 != en.w0 || back.w1 != en.w1 || back.w2 != en.w2 || back.w3 != en.w3) return -7;  // split / join must be lossless
				entry_seg.push_back((uint32_t)s);
				seg_nent[s]++;
			}
	for (uint64_t s = 0; s < nsegs; s++)
		if (seg_nent[s]) seg_dir[s] = make_segdir(seg_first[s], seg_nent[s], &seg_mask[4ull * s]);
	EmitCtx<T> c;
	c.tab = tab; c.P = P; c.G = G;
	c.seg_base = seg_base.data(); c.seg_dir = seg_dir.data();
	c.entries_a = ea.data(); c.entries_b = eb.data(); c.entries_c = ec.data(); c.fast_b = fast_b; c.fast_b_in_lds = false; c.entry_seg = entry_seg.data();
	c.z_emit = z_emit;
	// pass 2 (k_slow_count): triangles of the slow cells with a corner equal to the isovalue, by vertex identity on the stored
	// plans - checked against the round-1 formulation that plans every owner on the way (count_triangles)
	uint64_t keys[13];
	for (size_t k = 0; k < ea.size(); k++) {
		if (!(ea[k].a0 & ENTRYA_COUNT)) continue;
		const Entry en = entry_join(ea[k], eb[k]);
		CellPlan p;
		plan_restore(p, tab.lut, en, ec[k]);
		const SegCoord sc = segment_coord(P, entry_seg[k]);
		const uint32_t x = sc.xbase + (ea[k].a0 & 0xFFu);
		RootMemo memo{keys, 1, 0u};
		const uint32_t nt = count_triangles_stored(c, p, x, sc.y, sc.z, w, memo, (uint64_t)entry_seg[k], (uint32_t)k);  // (the slots resolved together: roots_together)
		if (nt != count_triangles(p, tab, P, G, x, sc.y, sc.z, w)) return -11;
		ea[k].a0 = (ea[k].a0 & ~(15u << 20) & ~ENTRYA_COUNT) | nt << 20;
	}
	// pass 3 (the wave scan of k_cells / k_seg_fix): offsets of the records inside their row segment, segment totals
	for (uint64_t s = 0; s < nsegs; s++) {
		uint32_t nv = 0, nt = 0;
		for (uint32_t k = 0; k < seg_nent[s]; k++) {
			EntryA &e = ea[seg_first[s] + k];
			e.a1 = nv | nt << 16;
			nv += entrya_nnew(e); nt += entrya_ntri(e);
		}
		seg_cnt[s] = seg_pack(nv, nt);
	}
	// scan in sweep order over records stored in [z][segment][y] order
	uint64_t nV = 0, nT = 0;
	for (uint64_t q = 0; q < nsegs; q++) {
		const uint64_t s = segment_sweep_to_store(P, q);
		seg_base[s].vbase = (uint32_t)nV; seg_base[s].tbase = (uint32_t)nT;
		nV += seg_cnt[s] & 0xFFFF; nT += seg_cnt[s] >> 16;
	}
	const uint64_t gseg = (uint64_t)(z_emit - P.zs) * P.ny * P.nseg;  // first segment of the emitted range (same index in both orders)
	const uint32_t gV = gseg ? seg_base[gseg].vbase : 0, gT = gseg ? seg_base[gseg].tbase : 0;
	nV -= gV; nT -= gT;
	out->nV = (uint32_t)nV; out->nT = (uint32_t)nT;
	out->V = (float *)malloc(nV * 12 + 16); out->N = (float *)malloc(nV * 12 + 16); out->T = (uint32_t *)malloc(nT * 12 + 16);
	memset(out->T, 0xFF, nT * 12);
	c.V = out->V; c.N = out->N; c.Tri = out->T;
	c.v_skip = gV; c.t_skip = gT; c.id_delta = id_base - gV;
	const size_t nrec = ea.size();
	ea.push_back(EntryA{0u, 0u}); ea.push_back(EntryA{0u, 0u});  // (the triangle pass reads records in pairs: as the device's array, two more)
	c.entries_a = ea.data();
	for (size_t k = 0; k < nrec; k++) {
		if (ea[k].a0 & ENTRYA_SLOW) emit_cell(c, (uint32_t)k, v, w, ids);
		else {
			const Entry en = load_entry(ea.data(), eb.data(), fast_b, (uint32_t)k);
			emit_fast_vertices(c, en, entry_seg[k]);
			emit_fast_triangles(c, en, ea[k ? k - 1 : 0], entry_seg[k], (uint32_t)k, ids);
		}
	}
	return 0;
}

extern "C" int emu_isosurface_f32(const float *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0,
                                  const double *d, float iso, emu_surface *out) {
	return run<float>(data, npx, npy, npz, r0, d, iso, out);
}
extern "C" int emu_isosurface_u16(const uint16_t *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0,
                                  const double *d, float iso, emu_surface *out) {
	return run<uint16_t>(data, npx, npy, npz, r0, d, iso, out);
}
extern "C" void emu_free(emu_surface *s) { free(s->V); free(s->N); free(s->T); memset(s, 0, sizeof *s); }

extern "C" int emu_slab_f32(const float *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0, const double *d,
                            float iso, const emu_slab *slab, emu_surface *out, unsigned long long *violations) {
	g_violations = 0;
	int rc = run<float>(data, npx, npy, npz, r0, d, iso, out, slab);
	*violations = g_violations;
	return rc;
}
extern "C" int emu_slab_u16(const uint16_t *data, uint32_t npx, uint32_t npy, uint32_t npz, const double *r0, const double *d,
                            float iso, const emu_slab *slab, emu_surface *out, unsigned long long *violations) {
	g_violations = 0;
	int rc = run<uint16_t>(data, npx, npy, npz, r0, d, iso, out, slab);
	*violations = g_violations;
	return rc;
}
extern "C" unsigned long long emu_last_violations(void) { return g_violations; }

// Every sign index with `per_index` random sets of corner values (no value 0): the record k_cells makes for an
// interior cell from the pattern offset + the pattern-info table must be the one the generic plan makes, the tests on
// register-held values (Corner8) must choose the pattern the tests on a VRef choose, and the stored plan must give the
// plan back.  Returns the number of (index, values) pairs whose pattern needed tests, or a negative error code;
// hist[g] counts the patterns by table group (c >> 12).
extern "C" long emu_check_tested_records(uint32_t seed, uint32_t per_index, unsigned long long *hist /*[8]*/) {
	Tables tab{mc33_lut, mc33_rule_words, &mc33_rule_index[0][0]};
	constexpr uint32_t lut_n = sizeof mc33_lut / sizeof mc33_lut[0];
	static uint32_t pat_info[lut_n];
	build_pattern_info(mc33_lut, lut_n, pat_info);
	uint32_t fast[256];
	build_fast_table(mc33_lut, fast);
	Params P{};
	P.nx = P.ny = P.nz = 1u << 20;
	GridView<float> G{nullptr, 0, 0, 0};
	uint64_t st = seed * 0x9E3779B97F4A7C15ull + 1;
	auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 32); };
	long tested = 0;
	for (uint32_t i = 1; i < 255; i++)
		for (uint32_t r = 0; r < per_index; r++) {
			float vb[8];
			Corner8 c8;
			for (int k = 0; k < 8; k++) {
				// magnitudes over several decades, now and then equal ones (ties in the face tests)
				const uint32_t q = rnd();
				float mag = (q & 7u) == 0 ? 1.0f : (float)((q >> 8) % 1000u + 1u) * ((q & 8u) ? 0.001f : 1.0f);
				vb[k] = ((i >> (7 - k)) & 1) ? -mag : mag;
				c8.a[k] = vb[k];
			}
			CellPlan p;
			plan_cell(p, tab, P, G, 5, 6, 7, i, VRef{vb, 1});
			if (!cell_is_tested(p, 5, 6, 7)) return -1;
			uint32_t m, n;
			if (pattern_offset(mc33_lut, i, c8, m, n) != p.poff || m != p.m || n != p.n) return -2;
			const Entry want = make_entry(37, i, p, p.ntri, 11, 13, false), got = make_tested_entry(37, i, p.poff, pat_info[p.poff], 11, 13);
			if (got.w0 != want.w0 || got.w1 != want.w1 || got.w2 != want.w2 || got.w3 != (want.w3 | ENTRY_TESTED)) return -3;
			if (fast[i] != FAST_NONE) {
				const Entry f = make_fast_entry(37, i, fast[i], 11, 13);
				if (f.w0 != want.w0 || f.w1 != want.w1 || f.w2 != want.w2 || f.w3 != want.w3) return -4;
			} else
				tested++;
			CellPlan q;
			const Entry slow = make_entry(37, i, p, p.ntri, 11, 13, true);
			plan_restore(q, mc33_lut, slow, entry_c(p));
			plan_restore_points(q, VRef{vb, 1});
			if (q.rank != p.rank || q.visited != p.visited || q.created != p.created || q.onpoint != p.onpoint || q.onb != p.onb || q.tgt[0] != p.tgt[0] ||
			    q.tgt[1] != p.tgt[1] || q.tgt[2] != p.tgt[2] || q.poff != p.poff || q.m != p.m || q.n != p.n || q.nnew != p.nnew || q.zmask != p.zmask)
				return -5;
			hist[mc33_lut[(i & 0x80) ? (i ^ 0xFF) : i] >> 12]++;
		}
	return tested;
}
