"""mc33_c_library_amd - MI355X-native Marching Cubes 33 isosurface extraction.

The product is the pair of shared objects libMC33_f32.so / libMC33_u16.so (HIP kernels for gfx950 behind
the C API of the reference's marching_cubes_33.h and the device-level C ABI of include/mc33_hip.h).
This package only holds the build recipe and thin ctypes/torch plumbing around them.
"""
from .api import (Counts, DeviceGrid, GridDesc, MC33Error, Range, Timing, HIP_API, REFERENCE_API, library_path,  # noqa: F401
                  load_library)
