"""Python view of the C ABI (include/mc33_hip.h, include/marching_cubes_33.h).

Plumbing only: device memory and streams come from PyTorch-ROCm, the work is done by the HIP kernels
inside libMC33_{f32,f64,u8,u16,u32}.so (one library per grid sample type).  There is no CPU fallback: loading fails loudly when the shared object is
missing, and every call into it fails loudly when no GPU is present.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))

OK, EINVAL, ENOGPU, ENOMEM, ECAPACITY, ERUNTIME, EOVERFLOW = 0, -1, -2, -3, -4, -5, -6


class GridDesc(C.Structure):
    _fields_ = [("npx", C.c_uint), ("npy", C.c_uint), ("npz_resident", C.c_uint), ("plane0", C.c_uint),
                ("nz_total", C.c_uint), ("r0", C.c_double * 3), ("d", C.c_double * 3),
                ("sample_bytes", C.c_int), ("device", C.c_int)]


class Range(C.Structure):
    _fields_ = [("z_begin", C.c_uint), ("z_end", C.c_uint), ("ghost_below", C.c_uint), ("id_base", C.c_uint)]


class Counts(C.Structure):
    _fields_ = [("nV", C.c_ulonglong), ("nT", C.c_ulonglong), ("nV_ghost", C.c_ulonglong),
                ("nT_ghost", C.c_ulonglong), ("active_cells", C.c_ulonglong)]


class Timing(C.Structure):
    _fields_ = [("sweep_ms", C.c_float), ("scan_ms", C.c_float), ("emit_ms", C.c_float), ("total_ms", C.c_float),
                ("sweep_launches", C.c_uint)]


HIP_API = ["mc33hip_set_id_base", "mc33hip_create", "mc33hip_destroy", "mc33hip_last_error", "mc33hip_upload_rows",
           "mc33hip_upload_contiguous", "mc33hip_adopt_device", "mc33hip_set_stream", "mc33hip_count",
           "mc33hip_emit", "mc33hip_extract", "mc33hip_last_timing", "mc33hip_download",
           "mc33hip_device_alloc", "mc33hip_device_free", "mc33hip_set_inclined", "mc33hip_download_concurrent", "mc33hip_synchronize", "mc33hip_download_many", "mc33hip_set_normal_neg", "mc33hip_sweep_many", "mc33hip_set_timing", "mc33hip_probe_read", "mc33hip_prepare_many",
           "mc33hip_emit_download", "mc33hip_download_wait", "mc33hip_own_stream", "mc33hip_device_count", "mc33hip_count_async",
           "mc33hip_counts_to_device", "mc33hip_bases_from_table", "mc33hip_emit_at_device_bases", "mc33hip_count_finish"]
REFERENCE_API = ["create_MC33", "calculate_isosurface", "size_of_isosurface", "free_MC33", "free_surface_memory",
                 "adjustvectorlenght_s", "DefaultColorMC", "free_memory_grd", "alloc_F", "grid_from_data_pointer",
                 "generate_grid_from_fn", "_multTSA_bf", "_multA_bf", "mult_Abf",
                 "write_bin_s", "read_bin_s", "write_txt_s", "write_obj_s", "write_ply_s",
                 "read_grd", "read_grd_binary", "read_scanfiles", "read_raw_file", "read_dat_file", "calculate_isosurfaces", "MC33_grid_changed"]


class MC33Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("mc33hip error %d: %s" % (code, msg))
        self.code = code


def library_path(dtype="f32"):
    # MC33_LIB_DIR: developer switch (tools/): load a -DMC33_DEV build from another directory
    return os.path.join(os.environ.get("MC33_LIB_DIR") or PKG, "libMC33_%s.so" % dtype)


_libs = {}


def load_library(dtype="f32"):
    """dlopen the product library; raises if it was not built (python -m mc33_c_library_amd.build)."""
    if dtype in _libs:
        return _libs[dtype]
    path = library_path(dtype)
    if not os.path.exists(path):
        raise FileNotFoundError("%s not built - run `python -m mc33_c_library_amd.build` (hipcc, gfx950); "
                                "there is no CPU fallback" % path)
    try:  # torch's bundled HIP runtime must be the one the process loads first (same SONAME)
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    P, V = C.POINTER, C.c_void_p
    lib.mc33hip_create.argtypes = [P(V), P(GridDesc)]
    lib.mc33hip_destroy.argtypes = [V]
    lib.mc33hip_destroy.restype = None
    lib.mc33hip_last_error.restype = C.c_char_p
    lib.mc33hip_upload_rows.argtypes = [V, V]
    lib.mc33hip_upload_contiguous.argtypes = [V, V]
    lib.mc33hip_adopt_device.argtypes = [V, V, C.c_size_t, C.c_size_t]
    lib.mc33hip_set_stream.argtypes = [V, V]
    lib.mc33hip_count.argtypes = [V, C.c_double, P(Range), P(Counts)]
    lib.mc33hip_set_id_base.argtypes = [V, C.c_uint]
    lib.mc33hip_emit.argtypes = [V, V, V, V, C.c_ulonglong, C.c_ulonglong]
    lib.mc33hip_extract.argtypes = [V, C.c_double, P(Range), V, V, V, C.c_ulonglong, C.c_ulonglong, P(Counts)]
    lib.mc33hip_last_timing.argtypes = [V, P(Timing)]
    lib.mc33hip_download.argtypes = [V, V, V, C.c_size_t]
    lib.mc33hip_device_alloc.argtypes = [V, P(V), C.c_size_t]
    lib.mc33hip_device_free.argtypes = [V, V]
    lib.mc33hip_set_inclined.argtypes = [V, V, V, C.c_int]
    lib.mc33hip_set_normal_neg.argtypes = [V, C.c_int]
    lib.mc33hip_sweep_many.argtypes = [V, P(C.c_double), C.c_int, P(Range)]
    lib.mc33hip_set_timing.argtypes = [V, C.c_int]
    lib.mc33hip_prepare_many.argtypes = [V, P(C.c_double), C.c_int, P(Range)]
    lib.mc33hip_probe_read.argtypes = [V, C.c_int, P(C.c_float), P(C.c_float), P(C.c_ulonglong)]
    lib.mc33hip_count_async.argtypes = [V, C.c_double, P(Range)]
    lib.mc33hip_counts_to_device.argtypes = [V, V]
    lib.mc33hip_bases_from_table.argtypes = [V, V, C.c_int, C.c_int, C.c_int]
    lib.mc33hip_emit_at_device_bases.argtypes = [V, V, V, V, C.c_ulonglong, C.c_ulonglong]
    lib.mc33hip_count_finish.argtypes = [V, P(Counts)]
    _libs[dtype] = lib
    return lib


def _check(lib, rc, allow=()):
    if rc != OK and rc not in allow:
        raise MC33Error(rc, lib.mc33hip_last_error().decode(errors="replace"))
    return rc


class DeviceGrid:
    """A grid (or a z-slab of one) resident in HBM as a torch tensor [planes, rows, pitch], plus the
    extraction context working on it.  dtype float32 or uint16 (carried as torch.int16 bit patterns)."""

    def __init__(self, tensor, nz_total=None, plane0=0, r0=(0.0, 0.0, 0.0), d=(1.0, 1.0, 1.0), npx=None):
        import torch
        assert tensor.is_cuda and tensor.dim() == 3 and tensor.stride(2) == 1, "need a device tensor [z, y, x]"
        if tensor.dtype == torch.float32:
            self.dtype, sb = "f32", 4
        elif tensor.dtype in (torch.int16, torch.uint16):
            self.dtype, sb = "u16", 2
        elif tensor.dtype == torch.uint8:
            self.dtype, sb = "u8", 1
        elif tensor.dtype in (torch.int32, torch.uint32):
            self.dtype, sb = "u32", 4
        elif tensor.dtype == torch.float64:
            self.dtype, sb = "f64", 8
        else:
            raise TypeError("grid samples must be float32, float64, uint8, (u)int16 or (u)int32 bit patterns")
        self.lib = load_library(self.dtype)
        self.tensor = tensor  # keeps the memory alive
        npz, npy, pitch = tensor.shape[0], tensor.shape[1], tensor.stride(1)
        npx = npx if npx is not None else tensor.shape[2]
        desc = GridDesc(npx, npy, npz, plane0, (nz_total if nz_total is not None else npz - 1),
                        (C.c_double * 3)(*r0), (C.c_double * 3)(*d), sb, tensor.device.index)
        self.desc = desc
        self.ctx = C.c_void_p()
        _check(self.lib, self.lib.mc33hip_create(C.byref(self.ctx), C.byref(desc)))
        # (per-pass hipEvent timing is off, as in the library: a caller that wants timing() says set_timing(2) first -
        # the event records cost ~20 us per call; MC33_HIP_TIMING in the environment sets the level a context starts with)
        _check(self.lib, self.lib.mc33hip_adopt_device(self.ctx, C.c_void_p(tensor.data_ptr()), pitch, tensor.stride(0)))
        self.device = tensor.device
        self.use_stream(torch.cuda.current_stream(self.device))

    def set_timing(self, level):
        """0: no events (the production path), 1: whole call, 2: per pass (timing() then reports sweep / scan / emit)."""
        _check(self.lib, self.lib.mc33hip_set_timing(self.ctx, int(level)))

    def use_stream(self, stream):
        self.stream = stream
        _check(self.lib, self.lib.mc33hip_set_stream(self.ctx, C.c_void_p(stream.cuda_stream)))

    def set_inclined(self, A=None, Ai=None, triangular=False):
        """Non-orthogonal grid: _GRD._A / _GRD.A_ (3x3, row major); None switches back."""
        if A is None:
            _check(self.lib, self.lib.mc33hip_set_inclined(self.ctx, None, None, 0))
            return
        a = (C.c_double * 9)(*[float(x) for row in A for x in row])
        ai = (C.c_double * 9)(*[float(x) for row in Ai for x in row])
        _check(self.lib, self.lib.mc33hip_set_inclined(self.ctx, a, ai, int(bool(triangular))))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mc33hip_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    def full_range(self):
        return Range(0, self.desc.nz_total, 0, 0)

    def count(self, iso, rng=None):
        rng = rng or self.full_range()
        cnt = Counts()
        _check(self.lib, self.lib.mc33hip_count(self.ctx, C.c_double(iso), C.byref(rng), C.byref(cnt)))
        return cnt

    def sweep_many(self, isos, rng=None):
        """Classify up to 8 isovalues in one go (the grid is streamed once per 4 of them); the count / extract calls
        for these isovalues over the same range then skip their sweep."""
        rng = rng or self.full_range()
        arr = (C.c_double * len(isos))(*[float(x) for x in isos])
        _check(self.lib, self.lib.mc33hip_sweep_many(self.ctx, arr, len(isos), C.byref(rng)))

    def prepare_many(self, isos, rng=None):
        """sweep_many plus everything else a count needs, every isovalue into buffers of its own: count / emit_into for these
        isovalues then run in any order, any number of times (slabs: all counts first, ONE exchange, then the emits)."""
        rng = rng or self.full_range()
        arr = (C.c_double * len(isos))(*[float(x) for x in isos])
        _check(self.lib, self.lib.mc33hip_prepare_many(self.ctx, arr, len(isos), C.byref(rng)))

    def extract_into(self, iso, V, N, T, rng=None):
        """One extraction into caller-owned device tensors V, N [capV,3] float32 and T [capT,3] int32.
        Returns (Counts, enough_capacity)."""
        rng = rng or self.full_range()
        cnt = Counts()
        rc = self.lib.mc33hip_extract(self.ctx, C.c_double(iso), C.byref(rng), C.c_void_p(V.data_ptr()),
                                      C.c_void_p(N.data_ptr()), C.c_void_p(T.data_ptr()), V.shape[0], T.shape[0],
                                      C.byref(cnt))
        _check(self.lib, rc, allow=(ECAPACITY,))
        return cnt, rc == OK

    def emit_into(self, V, N, T, id_base=None):
        """Emit pass for the range last counted (asynchronous on the stream)."""
        if id_base is not None:
            _check(self.lib, self.lib.mc33hip_set_id_base(self.ctx, id_base))
        _check(self.lib, self.lib.mc33hip_emit(self.ctx, C.c_void_p(V.data_ptr()), C.c_void_p(N.data_ptr()),
                                               C.c_void_p(T.data_ptr()), V.shape[0], T.shape[0]))

    # -- a z-slab's count, count exchange and emit without a host round trip in between (slabs.py: extract_slab) --------------
    def count_async(self, iso, rng=None):
        """What count() does, enqueued only: the counters stay on the device until count_finish()."""
        rng = rng or self.full_range()
        _check(self.lib, self.lib.mc33hip_count_async(self.ctx, C.c_double(iso), C.byref(rng)))

    def counts_to_device(self, dst):
        """{vertices, triangles} of the range last counted into dst, an int64 device tensor of 2 elements (stream-ordered)."""
        import torch
        assert dst.is_cuda and dst.dtype == torch.int64 and dst.numel() >= 2 and dst.is_contiguous()
        _check(self.lib, self.lib.mc33hip_counts_to_device(self.ctx, C.c_void_p(dst.data_ptr())))

    def bases_from_table(self, table, stride, rank, concatenated):
        """table: int64 device tensor, rank r's {vertices, triangles} at table[r * stride]: this rank's vertex id base and output rows."""
        import torch
        assert table.is_cuda and table.dtype == torch.int64 and table.is_contiguous() and table.numel() >= (rank + 1) * stride
        _check(self.lib, self.lib.mc33hip_bases_from_table(self.ctx, C.c_void_p(table.data_ptr()), int(stride), int(rank), int(bool(concatenated))))

    def emit_at_device_bases(self, V, N, T):
        """emit_into with the id base / output rows that bases_from_table left on the device."""
        _check(self.lib, self.lib.mc33hip_emit_at_device_bases(self.ctx, C.c_void_p(V.data_ptr()), C.c_void_p(N.data_ptr()),
                                                              C.c_void_p(T.data_ptr()), V.shape[0], T.shape[0]))

    def count_finish(self):
        """Waits for what count_async (and the emit behind it) enqueued: (Counts, True), or (Counts, False) when the work
        records or the output buffers were too small - room for the records has been made, repeat the step with count()."""
        cnt = Counts()
        rc = self.lib.mc33hip_count_finish(self.ctx, C.byref(cnt))
        _check(self.lib, rc, allow=(ECAPACITY,))
        return cnt, rc == OK

    def extract(self, iso, rng=None):
        """Count, allocate exact-size outputs with torch, emit.  Returns (V, N, T, Counts)."""
        import torch
        rng = rng or self.full_range()
        cnt = self.count(iso, rng)
        V = torch.empty((max(cnt.nV, 1), 3), dtype=torch.float64 if self.dtype == "f64" else torch.float32, device=self.device)
        N = torch.empty((max(cnt.nV, 1), 3), dtype=torch.float32, device=self.device)
        T = torch.empty((max(cnt.nT, 1), 3), dtype=torch.int32, device=self.device)
        _check(self.lib, self.lib.mc33hip_emit(self.ctx, C.c_void_p(V.data_ptr()), C.c_void_p(N.data_ptr()),
                                               C.c_void_p(T.data_ptr()), V.shape[0], T.shape[0]))
        self.stream.synchronize()
        return V[:cnt.nV], N[:cnt.nV], T[:cnt.nT], cnt

    def probe_read(self, reps=10):
        """A plain read of the resident grid (nothing to do with an extraction): (best ms, median ms, bytes)."""
        best, med, nbytes = C.c_float(), C.c_float(), C.c_ulonglong()
        _check(self.lib, self.lib.mc33hip_probe_read(self.ctx, int(reps), C.byref(best), C.byref(med), C.byref(nbytes)))
        return best.value, med.value, nbytes.value

    def timing(self):
        t = Timing()
        _check(self.lib, self.lib.mc33hip_last_timing(self.ctx, C.byref(t)))
        return t
