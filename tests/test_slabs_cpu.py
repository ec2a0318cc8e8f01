"""World-size-2 / 3 gloo tests of mc33_c_library_amd.slabs - the orchestration bench.py's N>1 path and the GPU slab
tests use: Slab plan, count exchange, emit at the global id base into the exchange buffers, surface exchange in all
three modes (padded all-gather, all-pairs send/recv, gather-to-root).  The extraction itself runs on the host
emulator here (no GPU in this container); everything else is the code the ranks run on the GPU box.  Rank 0 compares
the concatenated arrays with the unmodified reference (oracle/_ref) when it is built, else with the oracle restatement."""
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import fixtures as fx
    from mc33_emu import Emu
    from mc33_c_library_amd.slabs import Slab, SurfaceExchange, extract_slab, extract_slab_many, extract_slab_on_device, MODES

    class EmuGrid:
        """DeviceGrid's count / emit_into on the host emulator (tensors on the CPU)"""
        def __init__(self, data, slab):
            self.em, self.data, self.slab = Emu("f32"), data, slab
        def _run(self, iso, base):
            s = self.slab
            out = self.em.isosurface(self.data, iso, slab=(s.z_begin, s.z_end, s.ghost, base, s.p_lo, s.p_hi))
            assert self.em.violations == 0
            return out
        def count(self, iso, rng):
            self.iso = iso
            return self._run(iso, 0)
        def prepare_many(self, isos, rng):
            self.prepared = list(isos)
        def emit_into(self, V, N, T, id_base):
            s = self._run(self.iso, id_base)
            V[:s.nV] = torch.from_numpy(s.V); N[:s.nV] = torch.from_numpy(s.N); T[:s.nT] = torch.from_numpy(s.T.view(np.int32))
        # the flow with the counts "on the device" (slabs.extract_slab_on_device): the emulator stands in for the kernels, the
        # tensors live on the CPU and the collective is gloo - what is checked is the orchestration every rank runs over RCCL
        def count_async(self, iso, rng):
            self.iso, self.pending = iso, self._run(iso, 0)
        def counts_to_device(self, dst):
            dst[0], dst[1] = self.pending.nV, self.pending.nT
        def bases_from_table(self, table, stride, rank, concatenated):
            assert stride == 2 and not concatenated
            self.base = int(table.view(-1, 2)[:rank, 0].sum())
        def emit_at_device_bases(self, V, N, T):
            self.emit_into(V, N, T, self.base)
        def count_finish(self):
            return self.pending, True

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    data = fx.noise_quant(0, 5, shape=(33, 16, 24))   # samples equal to iso: aliases cross the slab interfaces
    iso, nzt = 0.0, data.shape[0] - 1
    slab = Slab(rank, world, nzt)
    grid = EmuGrid(data, slab)
    whole = None
    if rank == 0:
        from mc33_capi import MC33Lib, ref_path
        from mc33_oracle import Oracle
        whole = (MC33Lib(ref_path("f32"), "f32") if os.path.exists(ref_path("f32")) else Oracle("f32")).isosurface(data, iso)
    for mode in MODES:
        c0 = grid.count(iso, None)
        caps = torch.tensor([c0.nV + 7, c0.nT + 5])
        dist.all_reduce(caps, op=dist.ReduceOp.MAX)   # one capacity for all ranks, like bench.py
        ex = SurfaceExchange(world, rank, torch.device("cpu"), int(caps[0]), int(caps[1]), mode=mode, nbuf=2, host_collectives=True)
        for b in (0, 1, 0):
            counts, c = extract_slab(grid, slab, ex, iso, b)
            assert counts[rank] == (c.nV, c.nT)
        ex.drain()
        if mode != "root" or rank == 0:
            V, N, T = ex.concatenated(0, counts)
            if rank == 0:
                assert np.array_equal(T.numpy().view(np.uint32), whole.T), mode
                assert np.array_equal(V.numpy().view(np.uint32), whole.V.view(np.uint32)), mode
                nan = np.isnan(whole.N)
                assert np.array_equal(np.isnan(N.numpy()), nan) and np.array_equal(N.numpy()[~nan].view(np.uint32), whole.N[~nan].view(np.uint32)), mode
            else:  # every rank holds the same arrays
                assert T.shape[0] == sum(c[1] for c in counts)
        if mode == "allgather":
            assert ex.bytes_received == (world - 1) * (max(c[0] for c in counts) * 24 + max(c[1] for c in counts) * 12)
        elif mode == "pairs":
            assert ex.bytes_received == sum(c[0] * 24 + c[1] * 12 for r, c in enumerate(counts) if r != rank)
        else:
            assert ex.bytes_received == (sum(c[0] * 24 + c[1] * 12 for c in counts[1:]) if rank == 0 else 0)
        dist.barrier()
    # count, count exchange and emit with no host look at the counts in between; the surfaces gathered by capacity
    c0 = grid.count(iso, None)
    caps = torch.tensor([c0.nV + 7, c0.nT + 5])
    dist.all_reduce(caps, op=dist.ReduceOp.MAX)
    ex = SurfaceExchange(world, rank, torch.device("cpu"), int(caps[0]), int(caps[1]), mode="allgather", nbuf=2, host_collectives=False)
    for b in (0, 1, 0):
        counts, c = extract_slab_on_device(grid, slab, ex, iso, b)
        assert counts[rank] == (c.nV, c.nT)
    ex.drain()
    assert ex.rows[0] == (int(caps[0]), int(caps[1])) and ex.bytes_received == (world - 1) * (int(caps[0]) * 24 + int(caps[1]) * 12)
    V, N, T = ex.concatenated(0, counts)
    if rank == 0:
        assert np.array_equal(T.numpy().view(np.uint32), whole.T) and np.array_equal(V.numpy().view(np.uint32), whole.V.view(np.uint32))
    dist.barrier()
    # an iso sweep over the slabs: ALL counts in one collective, then the emits (extract_slab_many)
    isos = [0.0, 2.0, -3.5]
    c0 = [grid.count(v, None) for v in isos]
    caps = torch.tensor([max(c.nV for c in c0) + 7, max(c.nT for c in c0) + 5])
    dist.all_reduce(caps, op=dist.ReduceOp.MAX)
    ex = SurfaceExchange(world, rank, torch.device("cpu"), int(caps[0]), int(caps[1]), mode="pairs", nbuf=3, host_collectives=True)
    res = extract_slab_many(grid, slab, ex, isos)
    ex.drain()
    assert grid.prepared == isos and len(res) == 3
    for i, v in enumerate(isos):
        counts, c = res[i]
        assert counts[rank] == (c.nV, c.nT)
        V, N, T = ex.concatenated(i, counts)
        if rank == 0:
            from mc33_capi import MC33Lib, ref_path
            from mc33_oracle import Oracle
            w = (MC33Lib(ref_path("f32"), "f32") if os.path.exists(ref_path("f32")) else Oracle("f32")).isosurface(data, v)
            assert np.array_equal(T.numpy().view(np.uint32), w.T) and np.array_equal(V.numpy().view(np.uint32), w.V.view(np.uint32)), v
    dist.barrier()
    if rank == 0:
        print("SLABS_OK", whole.nV, whole.nT)
    dist.destroy_process_group()
''') % (HERE, ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_modes_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                          "--master-addr", "127.0.0.1", "--master-port", str(29540 + world), str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "SLABS_OK" in out.stdout


def test_slab_plan_covers_the_volume():
    from mc33_c_library_amd.slabs import Slab
    for nz, world in ((1023, 8), (33, 2), (9, 8), (2047, 2), (100, 3)):
        slabs = [Slab(r, world, nz) for r in range(world)]
        assert slabs[0].z_begin == 0 and slabs[-1].z_end == nz
        for a, b in zip(slabs, slabs[1:]):
            assert a.z_end == b.z_begin
        for s in slabs:
            assert s.p_lo == max(s.z_begin - (2 if s.rank else 0), 0) and s.p_hi == min(s.z_end + 1, nz)
            assert s.range().ghost_below == (1 if s.rank else 0)
    with pytest.raises(ValueError):
        Slab(0, 8, 7)  # fewer slices than ranks


def test_bench_parent_launches_ranks_without_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two rank processes itself (the driver's form) and
    fail with the ranks' exit code.  Here there is no GPU, so the ranks fail - what is checked is that the parent
    spawned them with the right environment, did not import torch, and returned non-zero."""
    probe = tmp_path / "sitecustomize.py"
    probe.write_text("import os\nopen(os.path.join(%r, 'rank_%%s_of_%%s' %% (os.environ.get('RANK', 'parent'), os.environ.get('WORLD_SIZE', '-'))), 'w').close()\n" % str(tmp_path))
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--points", "32"],
                         capture_output=True, text=True, timeout=300, env=env)
    names = sorted(os.listdir(tmp_path))
    assert "rank_0_of_2" in names and "rank_1_of_2" in names and "rank_parent_of_-" in names, names
    assert out.returncode != 0 and "exited with code" in out.stderr
