"""mc33_c_library_amd - MI355X-native Marching Cubes 33 isosurface extraction.

The product is the set of shared objects libMC33_<type>.so, <type> = f32, f64, u8, u16, u32 (+ the _ortho and _nneg
flavours): HIP kernels for gfx950 behind the C API of the reference's marching_cubes_33.h and the device-level C ABI
of include/mc33_hip.h.  This package only holds the build recipe, thin ctypes/torch plumbing around them (api.py)
and the host-side z-slab orchestration over several GPUs (slabs.py).
"""
from .api import (Counts, DeviceGrid, GridDesc, MC33Error, Range, Timing, HIP_API, REFERENCE_API, library_path,  # noqa: F401
                  load_library)
