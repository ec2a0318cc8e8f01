"""Rank process of tests/test_gpu_multirank.py: the real HIP slab path with several ranks on ONE GPU (cuda:0), the
collectives over gloo on host copies.  Each rank uploads only the planes of its slab, runs count -> exchange of
counts -> emit at the global id base -> surface exchange through mc33_c_library_amd.slabs (the code bench.py's N>1 path
runs); rank 0 compares the concatenated arrays with the UNMODIFIED REFERENCE (oracle/_ref) on the whole grid."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import fixtures as fx  # noqa: E402
from mc33_capi import MC33Lib, ref_path  # noqa: E402
from mc33_c_library_amd import DeviceGrid  # noqa: E402
from mc33_c_library_amd.slabs import MODES, Slab, SurfaceExchange, extract_slab  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cases = [
        # samples equal to the isovalue everywhere: aliases are chased across the slab interfaces
        ("f32", fx.noise_quant(0, 5, shape=(41, 24, 300)), 0.0, None, None),
        ("f32", fx.noise_quant(0, 9, L=3, shape=(26, 70, 40)), 0.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0)),
        ("u16", fx.noise_u16(0, 2, 7, shape=(37, 20, 30)), 3.0, None, None),
        ("f32", fx.noise_f32(0, 3, shape=(33, 66, 70)), 0.0, None, None),       # all 8 table groups
        ("f32", fx.cos_field(72)[0], 0.0, (-4.0, -4.0, -4.0), (8 / 71,) * 3),
    ]
    done = 0
    for dtype, data, iso, r0, d in cases:
        nzt = data.shape[0] - 1
        slab = Slab(rank, world, nzt)
        t = torch.from_numpy(np.ascontiguousarray(data[slab.p_lo:slab.p_hi + 1]))
        if t.dtype == torch.uint16:
            t = t.view(torch.int16)
        grid = DeviceGrid(t.to(dev), nz_total=nzt, plane0=slab.p_lo, r0=r0 or (0, 0, 0), d=d or (1, 1, 1))
        ref = MC33Lib(ref_path(dtype), dtype).isosurface(data, iso, r0, d) if rank == 0 else None
        for mode in MODES:
            c = grid.count(iso, slab.range())
            print("[rank %d] %s %s iso %g mode %s: slab [%d, %d) nV %d nT %d" % (rank, dtype, data.shape, iso, mode, slab.z_begin, slab.z_end, c.nV, c.nT), file=sys.stderr)
            caps = torch.tensor([c.nV + 16, c.nT + 16])
            dist.all_reduce(caps, op=dist.ReduceOp.MAX)
            ex = SurfaceExchange(world, rank, dev, int(caps[0]), int(caps[1]), mode=mode, nbuf=2, host_collectives=True)
            for b in (0, 1, 0):
                counts, c = extract_slab(grid, slab, ex, iso, b)
            ex.drain()
            torch.cuda.synchronize()
            if rank == 0:
                V, N, T = (x.cpu().numpy() for x in ex.concatenated(0, counts))
                assert (V.shape[0], T.shape[0]) == (ref.nV, ref.nT), (dtype, mode, V.shape, T.shape, ref.nV, ref.nT)
                assert np.array_equal(T.view(np.uint32), ref.T), (dtype, mode)
                assert np.array_equal(V.view(np.uint32), ref.V.view(np.uint32)), (dtype, mode)
                nan = np.isnan(ref.N)
                assert np.array_equal(np.isnan(N), nan) and np.array_equal(N[~nan].view(np.uint32), ref.N[~nan].view(np.uint32)), (dtype, mode)
                assert sum(1 for cnt in counts if cnt[0]) == world, "every rank should hold a piece of the surface: %s %s %s %s" % (counts, dtype, data.shape, mode)
                done += 1
            dist.barrier()
        grid.close()
    if rank == 0:
        print("GPU_SLABS_OK %d" % done)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
