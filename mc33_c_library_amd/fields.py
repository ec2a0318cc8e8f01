"""Synthetic scalar fields of the benchmark configs, built directly in HBM.

The 1-D cosine tables are computed on the host exactly like generate_grid_from_fn walks its axes
(coordinates advanced by repeated addition in double, reference MC33_util_grd.c:660-672); the separable
sum (cos x + cos y) + cos z and the cast to the sample type run on the GPU, which gives the same bits as
the host generator in tests/fixtures.py.
"""
import numpy as np


def axis_accum(lo, step, n):
    a = np.full(n, step, dtype=np.float64)
    a[0] = lo
    return np.add.accumulate(a)


def cos_field_slab(n_xy, z_points, h, lo, device, z_first=0):
    """float32 tensor [len(z range), n_xy, n_xy] of cos x + cos y + cos z; plane k has global index
    z_first + k.  z_points = number of planes wanted."""
    import torch
    cx = torch.from_numpy(np.cos(axis_accum(lo, h, n_xy))).to(device)
    zall = axis_accum(lo, h, z_first + z_points)[z_first:]
    cz = torch.from_numpy(np.cos(zall)).to(device)
    out = torch.empty((z_points, n_xy, n_xy), dtype=torch.float32, device=device)
    xy = cx[None, :] + cx[:, None]  # cos x + cos y, [y, x]
    step = 64
    for k in range(0, z_points, step):  # bounded float64 temporaries
        out[k:k + step] = (xy[None, :, :] + cz[k:k + step, None, None]).to(torch.float32)
    return out


def cos_field_cube(n, device, lo=-4.0, hi=4.0):
    """BASELINE.json configs[0..2]: n^3 points on [lo,hi]^3.  Returns (tensor, r0, d)."""
    h = (hi - lo) / (n - 1)
    return cos_field_slab(n, n, h, lo, device), (lo, lo, lo), (h, h, h)


def cos_field_u16(nx, ny, nz, device, z_first=0, nz_total=None):
    """BASELINE.json configs[4] field: ushort 32768 + 10000 (cos x + cos y + cos z), x,y in [-8,8],
    z in [-4,4]; returned as int16 bit patterns [nz, ny, nx]."""
    import torch
    nz_total = nz_total or nz
    cx = torch.from_numpy(np.cos(np.linspace(-8.0, 8.0, nx))).to(device)
    cy = torch.from_numpy(np.cos(np.linspace(-8.0, 8.0, ny))).to(device)
    cz = torch.from_numpy(np.cos(np.linspace(-4.0, 4.0, nz_total))[z_first:z_first + nz]).to(device)
    out = torch.empty((nz, ny, nx), dtype=torch.int16, device=device)
    xy = cx[None, :] + cy[:, None]
    step = 32
    for k in range(0, nz, step):
        f = torch.round(32768.0 + 10000.0 * (xy[None, :, :] + cz[k:k + step, None, None])).to(torch.int32)
        out[k:k + step] = torch.where(f >= 32768, f - 65536, f).to(torch.int16)
    return out
