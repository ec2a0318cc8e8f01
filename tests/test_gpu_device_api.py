"""GPU tests of the device-level C ABI (include/mc33_hip.h) through mc33_c_library_amd.DeviceGrid:
torch-owned device memory in, device arrays out; z-slab decomposition with a ghost slice must
reproduce the whole-volume result exactly (this is the N>1 path of bench.py without the collective)."""
import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu


def whole(data, iso, r0=(0, 0, 0), d=(1, 1, 1)):
    import torch
    from mc33_c_library_amd import DeviceGrid
    t = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    if t.dtype == torch.uint16:
        t = t.view(torch.int16)
    g = DeviceGrid(t, r0=r0, d=d)
    V, N, T, cnt = g.extract(iso)
    return V.cpu().numpy(), N.cpu().numpy(), T.cpu().numpy().view(np.uint32), cnt


def slabbed(data, iso, cuts, r0=(0, 0, 0), d=(1, 1, 1)):
    """Emulates ranks = len(cuts)+1 z-slabs on one GPU: every slab has its own context holding only the
    planes it needs, counts are 'exchanged' on the host, outputs concatenated."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    nz_total = data.shape[0] - 1
    bounds = [0] + list(cuts) + [nz_total]
    grids, counts = [], []
    for r in range(len(bounds) - 1):
        zb, ze = bounds[r], bounds[r + 1]
        ghost = 1 if zb else 0
        p_lo = max(zb - ghost - 1, 0)
        p_hi = min(ze + 1, nz_total)
        t = torch.from_numpy(np.ascontiguousarray(data[p_lo:p_hi + 1])).cuda()
        if t.dtype == torch.uint16:
            t = t.view(torch.int16)
        g = DeviceGrid(t, nz_total=nz_total, plane0=p_lo, r0=r0, d=d)
        c = g.count(iso, Range(zb, ze, ghost, 0))
        grids.append(g)
        counts.append(c)
    Vs, Ns, Ts = [], [], []
    base = 0
    for g, c in zip(grids, counts):
        V = torch.empty((max(c.nV, 1), 3), dtype=torch.float32, device="cuda")
        N = torch.empty_like(V)
        T = torch.empty((max(c.nT, 1), 3), dtype=torch.int32, device="cuda")
        g.emit_into(V, N, T, base)
        torch.cuda.synchronize()
        Vs.append(V[:c.nV].cpu().numpy()); Ns.append(N[:c.nV].cpu().numpy()); Ts.append(T[:c.nT].cpu().numpy().view(np.uint32))
        base += c.nV
    return np.concatenate(Vs), np.concatenate(Ns), np.concatenate(Ts), counts


def beq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_device_api_matches_reference(reflibs):
    data, r0, d = fx.cos_field(96)
    V, N, T, cnt = whole(data, 0.0, r0, d)
    ref = reflibs["f32"].isosurface(data, 0.0, r0, d)
    assert cnt.nV == ref.nV and cnt.nT == ref.nT and np.array_equal(T, ref.T) and beq(V, ref.V) and beq(N, ref.N)


@pytest.mark.parametrize("case", ["cos", "noise", "quant", "u16"])
@pytest.mark.parametrize("cuts", [(20,), (7, 8, 30), (1, 46)])
def test_z_slabs_reproduce_whole_volume(case, cuts):
    if case == "cos":
        data, iso = fx.cos_field(48)[0], 0.0
    elif case == "noise":
        data, iso = fx.noise_f32(0, 3, shape=(48, 20, 70)), 0.0
    elif case == "quant":
        data, iso = fx.noise_quant(0, 5, shape=(48, 24, 40)), 0.0   # aliases chase across the slab interface
    else:
        data, iso = fx.noise_u16(0, 2, 7, shape=(48, 20, 30)), 3.0
    V0, N0, T0, c0 = whole(data, iso)
    V1, N1, T1, cs = slabbed(data, iso, cuts)
    assert sum(c.nV for c in cs) == c0.nV and sum(c.nT for c in cs) == c0.nT
    assert np.array_equal(T0, T1) and beq(V0, V1) and beq(N0, N1)


@pytest.mark.parametrize("case", ["cos", "quant", "u16"])
@pytest.mark.parametrize("concatenated", [False, True])
def test_count_exchange_on_the_device(case, concatenated):
    """The slab flow with the counts never on the host between count and emit (mc33hip_count_async / counts_to_device /
    bases_from_table / emit_at_device_bases - slabs.extract_slab_on_device without the collective): every slab's kernel writes its
    {vertices, triangles} into ITS row of one device table - what all_gather_into_tensor leaves on every rank - a one-thread kernel
    makes the slab's id base (and, `concatenated`, its rows in arrays shared by all slabs) of it, the emit passes read that.  Enqueued
    back to back for all slabs; the host looks only at the end.  Must equal the whole-volume arrays."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    if case == "cos":
        data, iso = fx.cos_field(48)[0], 0.0
    elif case == "quant":
        data, iso = fx.noise_quant(0, 5, shape=(48, 24, 40)), 0.0   # aliases chase across the slab interfaces
    else:
        data, iso = fx.noise_u16(0, 2, 7, shape=(48, 20, 30)), 3.0
    V0, N0, T0, c0 = whole(data, iso)
    nz_total, cuts = data.shape[0] - 1, (7, 8, 30)
    bounds = [0] + list(cuts) + [nz_total]
    world = len(bounds) - 1
    grids, ranges = [], []
    for r in range(world):
        zb, ze = bounds[r], bounds[r + 1]
        ghost = 1 if zb else 0
        p_lo, p_hi = max(zb - ghost - 1, 0), min(ze + 1, nz_total)
        t = torch.from_numpy(np.ascontiguousarray(data[p_lo:p_hi + 1])).cuda()
        if t.dtype == torch.uint16:
            t = t.view(torch.int16)
        grids.append(DeviceGrid(t, nz_total=nz_total, plane0=p_lo))
        ranges.append(Range(zb, ze, ghost, 0))
    for attempt in range(2):  # (the first pass sizes every slab's record buffers through the synchronous path, as bench.py's capacity pass does)
        for g, rg in zip(grids, ranges):
            g.count(iso, rg)
    table = torch.zeros(world * 2, dtype=torch.int64, device="cuda")
    capV, capT = c0.nV + 64, c0.nT + 64
    if concatenated:
        V = torch.zeros((capV, 3), dtype=torch.float32, device="cuda"); N = torch.zeros_like(V)
        T = torch.zeros((capT, 3), dtype=torch.int32, device="cuda")
        outs = [(V, N, T)] * world
    else:
        outs = [(torch.zeros((capV, 3), dtype=torch.float32, device="cuda"), torch.zeros((capV, 3), dtype=torch.float32, device="cuda"),
                 torch.zeros((capT, 3), dtype=torch.int32, device="cuda")) for _ in range(world)]
    for r, (g, rg) in enumerate(zip(grids, ranges)):       # every "rank": count, its pair into the table - nothing waited for
        g.count_async(iso, rg)
        g.counts_to_device(table[2 * r:2 * r + 2])
    for r, g in enumerate(grids):                          # (the collective would sit here) bases from the table, emit
        g.bases_from_table(table, 2, r, concatenated)
        g.emit_at_device_bases(*outs[r])
    fin = [g.count_finish() for g in grids]                # only now does the host look
    assert all(ok for _, ok in fin)
    counts = [(int(a), int(b)) for a, b in table.view(world, 2).tolist()]
    assert counts == [(c.nV, c.nT) for c, _ in fin] and sum(c[0] for c in counts) == c0.nV and sum(c[1] for c in counts) == c0.nT
    if concatenated:
        V1, N1, T1 = V[:c0.nV].cpu().numpy(), N[:c0.nV].cpu().numpy(), T[:c0.nT].cpu().numpy().view(np.uint32)
    else:
        V1 = np.concatenate([o[0][:c[0]].cpu().numpy() for o, c in zip(outs, counts)])
        N1 = np.concatenate([o[1][:c[0]].cpu().numpy() for o, c in zip(outs, counts)])
        T1 = np.concatenate([o[2][:c[1]].cpu().numpy().view(np.uint32) for o, c in zip(outs, counts)])
    assert np.array_equal(T0, T1) and beq(V0, V1) and beq(N0, N1)
    # output arrays that are too small: nothing is written past them, the step reports it
    g = grids[1]
    g.count_async(iso, ranges[1])
    g.counts_to_device(table[2:4])
    g.bases_from_table(table, 2, 1, concatenated)
    small = (torch.zeros((8, 3), dtype=torch.float32, device="cuda"), torch.zeros((8, 3), dtype=torch.float32, device="cuda"), torch.zeros((8, 3), dtype=torch.int32, device="cuda"))
    g.emit_at_device_bases(*small)
    c, ok = g.count_finish()
    assert not ok and c.nV == counts[1][0]
    for g in grids:
        g.close()


def test_capacity_error_reports_sizes():
    import torch
    from mc33_c_library_amd import DeviceGrid
    data = fx.noise_f32(24, 1)
    g = DeviceGrid(torch.from_numpy(data).cuda())
    V = torch.empty((8, 3), dtype=torch.float32, device="cuda")
    T = torch.empty((8, 3), dtype=torch.int32, device="cuda")
    cnt, ok = g.extract_into(0.0, V, V.clone(), T)
    assert not ok and cnt.nV > 8 and cnt.nT > 8
    V2, N2, T2, c2 = g.extract(0.0)
    assert c2.nV == cnt.nV and c2.nT == cnt.nT


def test_one_context_many_extractions_of_different_extent(reflibs):
    """Nothing is cleared between extractions of one context (epoch-stamped slice headers, alternating partial
    sums, counts rewritten in place): a long run of calls with changing isovalues and z ranges - whole volume,
    thin slabs, empty results, whole volume again - must each equal a fresh computation."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    data, r0, d = fx.cos_field(150)
    nz = data.shape[0] - 1
    t = torch.from_numpy(data).cuda()
    g = DeviceGrid(t, r0=r0, d=d)
    ref_cache = {}

    def reference(iso):
        if iso not in ref_cache:
            ref_cache[iso] = reflibs["f32"].isosurface(data, iso, r0, d)
        return ref_cache[iso]

    plan = [(0.0, None), (2.5, None), (0.0, (10, 14)), (5.0, None), (0.0, None), (1.0, (100, nz)), (-1.0, (0, 3)), (0.0, None),
            (2.5, (40, 41)), (2.5, None)]
    for step, (iso, zr) in enumerate(plan * 2):
        if zr is None:
            V, N, T, cnt = g.extract(iso)
            ref = reference(iso)
            assert (cnt.nV, cnt.nT) == (ref.nV, ref.nT), (step, iso)
            assert np.array_equal(T.cpu().numpy().view(np.uint32), ref.T) and beq(V.cpu().numpy(), ref.V) and beq(N.cpu().numpy(), ref.N)
        else:  # a slab of the same resident grid: compare with a fresh context doing the same
            rng = Range(zr[0], zr[1], 1 if zr[0] else 0, 0)
            V, N, T, cnt = g.extract(iso, rng)
            g2 = DeviceGrid(t, r0=r0, d=d)
            V2, N2, T2, cnt2 = g2.extract(iso, rng)
            assert (cnt.nV, cnt.nT) == (cnt2.nV, cnt2.nT), (step, iso, zr)
            assert torch.equal(T, T2) and torch.equal(V.view(torch.int32), V2.view(torch.int32)) and torch.equal(N.view(torch.int32), N2.view(torch.int32))
            g2.close()


def test_slow_emit_with_a_stale_count(reflibs):
    """`extract_into` does not stop to read the counters before the emit passes: which slow-emit kernel runs, and on how many
    blocks, follows the count of the PREVIOUS extraction of the context.  A half-integer isovalue on an unsigned char grid
    (a few slow records on the grid's faces: 16 lanes per record) followed by an integer one (every cut cell has a corner equal
    to it: tens of thousands of slow records, into a launch sized for the few) and back again - each must be the reference's."""
    import torch
    from mc33_c_library_amd import DeviceGrid
    f, _, _ = fx.cos_field(176)
    data = np.round(128.0 + 40.0 * f).astype(np.uint8)
    t = torch.from_numpy(data).cuda()
    g = DeviceGrid(t)
    refs = {iso: reflibs["u8"].isosurface(data, iso) for iso in (100.5, 100.0, 128.0)}
    cap_v, cap_t = max(r.nV for r in refs.values()) + 64, max(r.nT for r in refs.values()) + 64
    V = torch.empty((cap_v, 3), dtype=torch.float32, device="cuda")
    N = torch.empty_like(V)
    T = torch.empty((cap_t, 3), dtype=torch.int32, device="cuda")
    for step, iso in enumerate((100.5, 100.0, 100.0, 100.5, 128.0, 100.5)):
        V.fill_(float("nan")); N.fill_(float("nan")); T.fill_(-1)
        cnt, ok = g.extract_into(iso, V, N, T)
        ref = refs[iso]
        assert ok and (cnt.nV, cnt.nT) == (ref.nV, ref.nT), (step, iso)
        assert np.array_equal(T[:cnt.nT].cpu().numpy().view(np.uint32), ref.T), (step, iso)
        assert beq(V[:cnt.nV].cpu().numpy(), ref.V) and beq(N[:cnt.nV].cpu().numpy(), ref.N), (step, iso)
    g.close()


def test_segment_count_tags_start_over(reflibs):
    """The counts of the row segments carry the number of the tail that wrote them (1 .. 255 in turn; a word with another
    number counts as zero, so the rows of slices without cut cells are never written and nothing is cleared between
    extractions); when the numbers start over the host clears the array.  600 extractions of one context, isovalues and
    z ranges changing so that most row segments hold something from SOME earlier call: every one must equal the first
    computation of its kind."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    data = fx.noise_quant(40, 11)  # samples in {-2 .. 2}: slow cells and dirty row segments too
    r0, d = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    nz = data.shape[0] - 1
    t = torch.from_numpy(data).cuda()
    g = DeviceGrid(t, r0=r0, d=d)
    kinds = [(0.0, None), (0.5, None), (-1.0, (5, 20)), (1.5, (0, 3)), (9.0, None), (0.0, (30, nz))]
    first = {}
    for step in range(600):
        iso, zr = kinds[step % len(kinds)]
        rng = None if zr is None else Range(zr[0], zr[1], 1 if zr[0] else 0, 0)
        V, N, T, cnt = g.extract(iso) if rng is None else g.extract(iso, rng)
        got = (cnt.nV, cnt.nT, T.cpu().numpy().tobytes(), V.cpu().numpy().tobytes(), N.cpu().numpy().tobytes())
        if step < len(kinds):
            first[step] = got
            if zr is None:
                ref = reflibs["f32"].isosurface(data, iso, r0, d)
                assert (cnt.nV, cnt.nT) == (ref.nV, ref.nT) and np.array_equal(T.cpu().numpy().view(np.uint32), ref.T) and beq(V.cpu().numpy(), ref.V)
        else:
            assert got == first[step % len(kinds)], (step, iso, zr)
    g.close()


@pytest.mark.parametrize("dtype", ["f32", "u16", "u16mix", "u8"])
def test_sweep_many_equals_single_sweeps(reflibs, dtype):
    """mc33hip_sweep_many classifies up to 8 isovalues per pass over the grid (4 per kernel launch); the extractions
    that follow must be exactly what they are without it - and the reference's.  Odd counts (4 + 2 + 1 lanes), a slab with
    a ghost slice, an isovalue that is not among the swept ones, a sweep that is never used, one used after a regrow."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    if dtype == "f32":
        data, isos = fx.noise_quant(0, 3, shape=(40, 70, 300)), [0.0, 1.0, -1.0, 0.5, 1.5, -0.5, 2.0]
    elif dtype == "u16":   # half-integer isovalues: no sample can equal one (k_sweep classifies with one compare per sample)
        data, isos = fx.cos_field_u16(300, 130, 50), [15268.5 + 5000.0 * k for k in range(8)]
    elif dtype == "u16mix":  # integer isovalues (samples equal to them), half-integer ones, one above and one below the type's range
        data, isos, dtype = fx.cos_field_u16(300, 130, 50), [25268.0, 15268.5, 32768.0, 70000.0, -5.0, 40000.0], "u16"
    else:
        data, isos = fx.noise_u8(0, 4, 5, shape=(30, 66, 258)), [2.0, 1.5, 3.0]
    t = torch.from_numpy(data).cuda()
    if t.dtype == torch.uint16:
        t = t.view(torch.int16)
    g = DeviceGrid(t)
    g.sweep_many(isos)
    for iso in isos + [isos[0]]:  # (the last one: its sweep was used up - the call sweeps for itself)
        V, N, T, cnt = g.extract(iso)
        ref = reflibs[dtype].isosurface(data, iso)
        assert (cnt.nV, cnt.nT) == (ref.nV, ref.nT), (dtype, iso)
        assert np.array_equal(T.cpu().numpy().view(np.uint32), ref.T) and beq(V.cpu().numpy(), ref.V), (dtype, iso)
        nan = np.isnan(ref.N)
        assert np.array_equal(N.cpu().numpy()[~nan].view(np.uint32), ref.N[~nan].view(np.uint32)), (dtype, iso)
    # a slab with ghost slice; sweeps that are dropped unused; another range in between
    nz = data.shape[0] - 1
    rng = Range(nz // 2, nz, 1, 0)
    g.sweep_many(isos[:3], rng)
    V0, N0, T0, c0 = g.extract(isos[1])          # whole volume: no matching sweep
    V1, N1, T1, c1 = g.extract(isos[1], rng)     # the slab: uses lane 1
    g2 = DeviceGrid(t)
    V2, N2, T2, c2 = g2.extract(isos[1], rng)
    assert (c1.nV, c1.nT) == (c2.nV, c2.nT) and torch.equal(T1, T2) and torch.equal(V1.view(torch.int32), V2.view(torch.int32))
    # the whole-volume call reused lane 0, whose sweep over the slab was never consumed: the arrays, not only the count
    ref = reflibs[dtype].isosurface(data, isos[1])
    assert (c0.nV, c0.nT) == (ref.nV, ref.nT)
    assert np.array_equal(T0.cpu().numpy().view(np.uint32), ref.T) and beq(V0.cpu().numpy(), ref.V)
    nan = np.isnan(ref.N)
    assert np.array_equal(N0.cpu().numpy()[~nan].view(np.uint32), ref.N[~nan].view(np.uint32))


@pytest.mark.parametrize("dtype", ["f32", "u16mix", "u8"])
def test_prepare_many_counts_first_then_emits_in_any_order(reflibs, dtype):
    """mc33hip_prepare_many makes the sweeps AND the tails of up to 8 isovalues ahead, one launch of every tail kernel per pass
    for its 4 isovalues, each isovalue into buffers of its own: all counts can be read before the first emit (z-slabs: one
    exchange of counts per step), the emits come in any order, any number of times, also at another id base - and equal the
    reference's arrays.  7 isovalues = passes of 4 + 2 + 1; a slab with a ghost slice; a capacity that is too small."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    if dtype == "f32":
        data, isos = fx.noise_quant(0, 3, shape=(40, 70, 300)), [0.0, 1.0, -1.0, 0.5, 1.5, -0.5, 2.0]
    elif dtype == "u16mix":
        data, isos, dtype = fx.cos_field_u16(300, 130, 50), [25268.0, 15268.5, 32768.0, 70000.0, -5.0, 40000.0, 30268.5], "u16"
    else:
        data, isos = fx.noise_u8(0, 4, 5, shape=(30, 66, 258)), [2.0, 1.5, 3.0, 2.5, 1.0]
    t = torch.from_numpy(data).cuda()
    if t.dtype == torch.uint16:
        t = t.view(torch.int16)
    g = DeviceGrid(t)
    refs = [reflibs[dtype].isosurface(data, iso) for iso in isos]
    for rep in range(2):
        g.prepare_many(isos)
        cnts = [g.count(iso) for iso in isos]                       # all counts before the first emit
        for c, r in zip(cnts, refs):
            assert (c.nV, c.nT) == (r.nV, r.nT)
        order = list(range(len(isos)))[::-1] + [1, 1, 0]              # backwards, then some of them again
        for k in order:
            c = g.count(isos[k])                                    # selects the isovalue's buffers; nothing is computed
            assert (c.nV, c.nT) == (refs[k].nV, refs[k].nT)
            V = torch.empty((max(c.nV, 1), 3), dtype=torch.float32, device="cuda"); N = torch.empty_like(V)
            T = torch.empty((max(c.nT, 1), 3), dtype=torch.int32, device="cuda")
            base = 1000 * rep
            g.emit_into(V, N, T, base)
            torch.cuda.synchronize()
            r = refs[k]
            assert np.array_equal(T.cpu().numpy().view(np.uint32)[:r.nT], r.T + np.uint32(base)), (dtype, isos[k])
            assert beq(V.cpu().numpy()[:r.nV], r.V)
            nan = np.isnan(r.N)
            assert np.array_equal(N.cpu().numpy()[:r.nV][~nan].view(np.uint32), r.N[~nan].view(np.uint32))
    # extract (count + emit in one call) on a prepared isovalue, with too little room first
    g.prepare_many(isos[:3])
    small = torch.empty((4, 3), dtype=torch.float32, device="cuda")
    cnt, ok = g.extract_into(isos[1], small, torch.empty_like(small), torch.empty((4, 3), dtype=torch.int32, device="cuda"))
    assert (not ok or refs[1].nV <= 4) and (cnt.nV, cnt.nT) == (refs[1].nV, refs[1].nT)
    _same_as_reference(g.extract(isos[1]), refs[1], "prepared isovalue after a refused emit")
    _same_as_reference(g.extract(isos[2]), refs[2], "prepared isovalue through extract")
    # a slab with a ghost slice; an isovalue that was not prepared sweeps for itself
    nz = data.shape[0] - 1
    rng = Range(nz // 2, nz, 1, 0)
    g.prepare_many(isos[:5], rng)
    g2 = DeviceGrid(t)
    for k in (4, 0, 2):
        V1, N1, T1, c1 = g.extract(isos[k], rng)
        V2, N2, T2, c2 = g2.extract(isos[k], rng)
        assert (c1.nV, c1.nT) == (c2.nV, c2.nT) and torch.equal(T1, T2) and torch.equal(V1.view(torch.int32), V2.view(torch.int32))
    _same_as_reference(g.extract(isos[-1]), refs[-1], "whole volume after prepared slabs")
    g.close(); g2.close()


def test_probe_read_reports_the_resident_bytes():
    """mc33hip_probe_read (bench.py's roofline.read_ceiling): a plain read of the resident grid - every byte once, a positive time,
    and nothing of an extraction is disturbed by it."""
    import torch
    from mc33_c_library_amd import DeviceGrid
    data, r0, d = fx.cos_field(96)
    g = DeviceGrid(torch.from_numpy(data).cuda(), r0=r0, d=d)
    before = g.count(0.0)
    best, med, nbytes = g.probe_read(3)
    assert nbytes == data.nbytes and 0.0 < best <= med
    after = g.count(0.0)
    assert (before.nV, before.nT) == (after.nV, after.nT)
    g.close()


def _same_as_reference(got, ref, what):
    V, N, T, cnt = got
    assert (cnt.nV, cnt.nT) == (ref.nV, ref.nT), what
    assert np.array_equal(T.cpu().numpy().view(np.uint32), ref.T) and beq(V.cpu().numpy(), ref.V), what
    nan = np.isnan(ref.N)
    assert np.array_equal(N.cpu().numpy()[~nan].view(np.uint32), ref.N[~nan].view(np.uint32)), what


def test_sweeps_that_are_never_consumed_leave_nothing_behind(reflibs):
    """A lane swept ahead by mc33hip_sweep_many has added its slices to the partial sums k_slots scans; when its count /
    extract call never comes (another isovalue is asked for, the grid is adopted again, the range changes) the next sweep
    into that lane must start from clean sums in BOTH halves (they alternate by extraction): several such sequences, every
    result compared with the reference bit for bit."""
    import torch
    from mc33_c_library_amd import DeviceGrid, Range
    data = fx.noise_quant(0, 3, shape=(40, 70, 300))
    isos = [0.0, 1.0, -1.0, 0.5]
    t = torch.from_numpy(data).cuda()
    g = DeviceGrid(t)
    refs = {iso: reflibs["f32"].isosurface(data, iso) for iso in isos + [0.25, 1.5]}
    g.extract(0.25)                       # the lane has a history: both halves of the sums have been used
    g.extract(1.5)
    g.sweep_many(isos)                    # never consumed ...
    g.sweep_many(isos[::-1])              # ... swept again, never consumed either
    for k, iso in enumerate([0.25, 1.5, 0.25, 0.0]):   # lane 0, single sweeps, odd and even epochs
        _same_as_reference(g.extract(iso), refs[iso], ("after unused sweeps", k, iso))
    g.sweep_many(isos)
    g.lib.mc33hip_adopt_device(g.ctx, t.data_ptr(), t.stride(1), t.stride(0))   # "the grid changed": sweeps are forgotten
    for k, iso in enumerate([1.0, 0.0, 1.0]):
        _same_as_reference(g.extract(iso), refs[iso], ("after re-adopt", k, iso))
    nz = data.shape[0] - 1
    g.sweep_many(isos, Range(nz // 2, nz, 1, 0))   # another range, unused
    g.sweep_many(isos[:2])
    _same_as_reference(g.extract(isos[1]), refs[isos[1]], "lane 1 of the second sweep")
    _same_as_reference(g.extract(isos[0]), refs[isos[0]], "lane 0 of the second sweep")
    _same_as_reference(g.extract(0.25), refs[0.25], "and a plain call after them")


@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_sweep_made_for_one_zero_is_not_used_for_the_other(reflibs, dtype):
    """+0.0 and -0.0 compare equal, but a sample equal to the isovalue gives v = iso - F = -0 for -0.0: zero AND negative.
    A lane swept for one zero must not serve a call with the other (the lanes are matched by the bit pattern of the
    isovalue).  The grid holds many exact zeros.  For -0.0 the reference itself is not a function of its input:
    that result is compared with a fresh context's."""
    import torch
    from mc33_c_library_amd import DeviceGrid
    data = fx.noise_quant(0, 4, shape=(24, 40, 70)) if dtype == "f32" else fx.noise_u8(0, 4, 3, shape=(24, 40, 70))
    assert (data == 0).sum() > 100
    t = torch.from_numpy(data).cuda()
    ref = reflibs[dtype].isosurface(data, 0.0)
    fresh = DeviceGrid(t)
    Vn, Nn, Tn, cn = fresh.extract(-0.0)
    for order in ([-0.0, 0.0], [0.0, -0.0], [-0.0, 1.0], [0.0, 1.0]):
        g = DeviceGrid(t)
        g.sweep_many(order)
        _same_as_reference(g.extract(0.0), ref, (dtype, order, "+0.0"))
        V, N, T, c = g.extract(-0.0)
        assert (c.nV, c.nT) == (cn.nV, cn.nT) and torch.equal(T, Tn) and torch.equal(V.view(torch.int32), Vn.view(torch.int32)), (dtype, order, "-0.0")
        g.close()
