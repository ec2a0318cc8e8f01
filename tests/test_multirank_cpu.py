"""World-size-2 gloo test of the N>1 orchestration of bench.py: two processes each own a z-slab (with ghost
slice), exchange their counts with torch.distributed, rebase the ids and all-gather the padded surface
arrays; rank 0 checks that the concatenation equals the single-rank result.  The extraction itself runs on
the host emulator here (no GPU in this container); the collective pattern is the one bench.py uses."""
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    import fixtures as fx
    from mc33_emu import Emu
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    data = fx.noise_quant(0, 5, shape=(33, 16, 24))   # samples equal to iso: aliases cross the slab interface
    iso, nzt = 0.0, data.shape[0] - 1
    per = (nzt + world - 1) // world
    zb, ze = rank * per, min((rank + 1) * per, nzt)
    ghost = 1 if rank else 0
    em = Emu("f32")
    # 1. count pass (emulated with id_base 0), 2. exchange counts, 3. emit with the global base
    s0 = em.isosurface(data, iso, slab=(zb, ze, ghost, 0, max(zb - ghost - 1, 0), min(ze + 1, nzt)))
    counts = torch.zeros(world * 2, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, torch.tensor([s0.nV, s0.nT], dtype=torch.int64))
    counts = counts.view(world, 2)
    base = int(counts[:rank, 0].sum())
    s = em.isosurface(data, iso, slab=(zb, ze, ghost, base, max(zb - ghost - 1, 0), min(ze + 1, nzt)))
    assert em.violations == 0
    capV, capT = int(counts[:, 0].max()), int(counts[:, 1].max())
    def pad(a, n, dt):
        out = torch.zeros((n, 3), dtype=dt)
        out[:a.shape[0]] = torch.from_numpy(a.view(np.int32) if dt == torch.int32 else a)
        return out
    gV = torch.zeros(world * capV * 3); gT = torch.zeros(world * capT * 3, dtype=torch.int32)
    dist.all_gather_into_tensor(gV, pad(s.V, capV, torch.float32).view(-1))
    dist.all_gather_into_tensor(gT, pad(s.T, capT, torch.int32).view(-1))
    gV = gV.view(world, capV, 3); gT = gT.view(world, capT, 3)
    if rank == 0:
        whole = em.isosurface(data, iso)
        V = np.concatenate([gV[r, :int(counts[r, 0])].numpy() for r in range(world)])
        T = np.concatenate([gT[r, :int(counts[r, 1])].numpy() for r in range(world)]).view(np.uint32)
        assert np.array_equal(T, whole.T) and np.array_equal(V.view(np.uint32), whole.V.view(np.uint32))
        print("MULTIRANK_OK", whole.nV, whole.nT)
    dist.destroy_process_group()
''') % HERE


def test_two_rank_slab_exchange_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "MULTIRANK_OK" in out.stdout
