"""Multi-PROCESS GPU tests of the z-slab path: rank processes sharing cuda:0 of the one-GPU box,
collectives over gloo.  The processes are started by the helper of tests/launcher.py (started before this pytest
process touched the GPU)."""
import json
import os
import sys

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PORT = [29600]


def torchrun(launcher, world, script, env=None, timeout=900):
    PORT[0] += 1
    e = {"MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "2", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    e.update(env or {})
    return launcher.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                         "--master-port", str(PORT[0]), script], env=e, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_rank_processes_on_one_gpu_equal_reference(launcher, world):
    """Degenerate-rich float and ushort grids, noise that reaches every table group, a cos field; all three
    exchange modes; concatenation compared bit for bit with oracle/_ref inside rank 0."""
    out = torchrun(launcher, world, os.path.join(HERE, "gpu_slab_worker.py"))
    assert out["rc"] == 0, out["stdout"][-3000:] + out["stderr"][-5000:]
    assert "GPU_SLABS_OK 15" in out["stdout"], out["stdout"][-2000:]


def bench(launcher, args, env, timeout=900):
    e = {"MC33_BENCH_REHEARSAL": "1", "MC33_BENCH_VERIFY": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    e.update(env)
    out = launcher.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, unset=["WORLD_SIZE", "RANK", "LOCAL_RANK"], timeout=timeout)
    assert out["rc"] == 0, out["stdout"][-3000:] + out["stderr"][-5000:]
    line = [l for l in out["stdout"].splitlines() if l.startswith("{")]
    assert len(line) == 1, out["stdout"][-2000:]
    return json.loads(line[0]), out


@pytest.mark.parametrize("gather", ["allgather", "pairs", "root"])
def test_bench_self_launch_c3_rehearsal(launcher, reflibs, tmp_path, gather):
    """`python bench.py --gpus 2` exactly as the driver starts it (no WORLD_SIZE): the parent spawns the ranks; the
    ranks share cuda:0 (rehearsal), verify the concatenation against a whole-volume extraction (MC33_BENCH_VERIFY) and
    dump it; here it is compared with oracle/_ref on the same field."""
    dump = str(tmp_path / "surf.npz")
    n = 80
    res, out = bench(launcher, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--points", str(n), "--gather", gather], {"MC33_BENCH_DUMP": dump})
    assert res["n_gpus"] == 2 and res["config"]["name"] == "c3" and res["gather"]["mode"] == gather and res["scaling"] == "weak"
    assert set(res["gather"]["alone"]) == {"allgather", "pairs", "root"} and res["value"] > 0 and res["step_ms_min"] <= res["step_ms_max"]
    assert "equals whole-volume result: True" in out["stderr"]
    h = 8.0 / (n - 1)
    x = np.cos(fx.axis_accum(-4.0, h, n))
    z = np.cos(fx.axis_accum(-4.0, h, 2 * n))
    data = ((x[None, None, :] + x[None, :, None]) + z[:, None, None]).astype(np.float32)
    ref = reflibs["f32"].isosurface(data, 0.0, (-4.0, -4.0, -4.0), (h, h, h))
    got = np.load(dump)
    assert res["config"]["vertices"] == ref.nV and res["config"]["triangles"] == ref.nT
    assert np.array_equal(got["T"], ref.T) and np.array_equal(got["V"].view(np.uint32), ref.V.view(np.uint32))
    assert np.array_equal(got["N"].view(np.uint32), ref.N.view(np.uint32))


def test_bench_self_launch_c3_strong_scaling(launcher, reflibs, tmp_path):
    """`--strong`: BASELINE configs[3] read literally - ONE n^3 grid cut into z-slabs over the ranks (here 3 ranks, 26 / 27 / 26
    slices) - against oracle/_ref on the whole grid."""
    dump = str(tmp_path / "surf.npz")
    n = 80
    res, out = bench(launcher, ["--gpus", "3", "--steps", "2", "--warmup", "1", "--points", str(n), "--gather", "root", "--strong"], {"MC33_BENCH_DUMP": dump})
    assert res["n_gpus"] == 3 and res["scaling"] == "strong" and "equals whole-volume result: True" in out["stderr"]
    data, r0, d = fx.cos_field(n)
    ref = reflibs["f32"].isosurface(data, 0.0, r0, d)
    got = np.load(dump)
    assert res["config"]["vertices"] == ref.nV and res["config"]["triangles"] == ref.nT
    assert np.array_equal(got["T"], ref.T) and np.array_equal(got["V"].view(np.uint32), ref.V.view(np.uint32))
    assert np.array_equal(got["N"].view(np.uint32), ref.N.view(np.uint32))


@pytest.mark.parametrize("gather", ["allgather", "pairs", "root"])
def test_bench_strong_scaling_five_rank_processes(launcher, reflibs, tmp_path, gather):
    """configs[3]'s geometry with as many rank PROCESSES as a one-GPU box allows (6 processes may have the card open, this
    pytest process is one of them; the config names 8): `bench.py --gpus 5 --strong --points 256` started exactly as the
    driver starts it, the 256^3 grid in five z-slabs of 51 slices, every exchange mode; the concatenated surface against
    oracle/_ref on the whole grid."""
    dump = str(tmp_path / "surf.npz")
    n = 256
    res, out = bench(launcher, ["--gpus", "5", "--steps", "2", "--warmup", "1", "--points", str(n), "--gather", gather, "--strong", "--rank-timeout", "600"],
                     {"MC33_BENCH_DUMP": dump, "OMP_NUM_THREADS": "2"}, timeout=900)
    assert res["n_gpus"] == 5 and res["scaling"] == "strong" and res["gather"]["mode"] == gather
    assert "equals whole-volume result: True" in out["stderr"]
    assert res["rank_sweep_ms"]["min"] > 0 and res["rank_sweep_ms"]["max"] >= res["rank_sweep_ms"]["min"]
    data, r0, d = fx.cos_field(n)
    ref = reflibs["f32"].isosurface(data, 0.0, r0, d)
    got = np.load(dump)
    assert res["config"]["vertices"] == ref.nV and res["config"]["triangles"] == ref.nT
    assert np.array_equal(got["T"], ref.T) and np.array_equal(got["V"].view(np.uint32), ref.V.view(np.uint32))
    assert np.array_equal(got["N"].view(np.uint32), ref.N.view(np.uint32))


def test_bench_parent_kills_stuck_ranks(launcher):
    """--rank-timeout: a self-launched run whose ranks do not finish in time is killed by the parent (process groups of
    fresh children, never a re-exec), which exits non-zero and says so - instead of holding the GPU until somebody else's
    limit.  One second is not enough for anything."""
    out = launcher.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--points", "96", "--rank-timeout", "1"],
                       env={"MC33_BENCH_REHEARSAL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, unset=["WORLD_SIZE", "RANK", "LOCAL_RANK"], timeout=300)
    assert out["rc"] != 0 and "--rank-timeout" in out["stderr"] and not any(l.startswith("{") for l in out["stdout"].splitlines())


def test_bench_self_launch_c5_rehearsal(launcher, reflibs, tmp_path):
    """The ushort / 8-isovalue config on 3 rank processes (strong scaling: one grid cut into z-slabs)."""
    dump = str(tmp_path / "surf.npz")
    res, out = bench(launcher, ["--gpus", "3", "--steps", "2", "--warmup", "1", "--config", "c5", "--points", "40", "--gather", "pairs"], {"MC33_BENCH_DUMP": dump})
    assert res["n_gpus"] == 3 and res["dtype"] == "u16" and res["config"]["isovalues_per_step"] == 8 and res["scaling"] == "strong"
    data = fx.cos_field_u16(80, 80, 40)
    total_v = total_t = 0
    for k in range(8):
        nV, nT, _ = reflibs["u16"].sizes(data, 15268.5 + 5000.0 * k)
        total_v += nV; total_t += nT
    assert res["config"]["vertices"] == total_v and res["config"]["triangles"] == total_t
    ref = reflibs["u16"].isosurface(data, 15268.5 + 5000.0 * 7)
    got = np.load(dump)
    assert np.array_equal(got["T"], ref.T) and np.array_equal(got["V"].view(np.uint32), ref.V.view(np.uint32))


def test_bench_single_gpu_lines(launcher):
    """The default line and the C5 line at reduced size: contract keys, roofline and cpu_baseline objects."""
    for args, name in ((["--points", "128", "--with-c5", "on", "--c5-points", "48", "--c5-steps", "2"], "c3"), (["--config", "c5", "--points", "64"], "c5")):
        res, _ = bench(launcher, ["--steps", "3", "--warmup", "1"] + args, {"MC33_BENCH_REHEARSAL": "0", "MC33_BENCH_VERIFY": "0"})
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline", "cpu_baseline", "step_ms_min", "step_ms_median", "step_ms_max"):
            assert key in res, key
        assert res["config"]["name"] == name and res["n_gpus"] == 1 and res["vs_baseline"] is None
        assert res["roofline"]["bound"] == "hbm" and res["roofline"]["frac"] > 0 and res["roofline"]["traffic"] is None
        assert res["cpu_baseline"]["kind"] == "reference" and res["cpu_baseline"]["cores"] == 1 and res["cpu_baseline"]["value"] > 0
        assert res["with_event_records"]["ms_per_step"] > 0 and res["roofline"]["step_bytes_moved"] > res["roofline"]["algorithmic_bytes_per_launch"]
        if name == "c3":  # the configs[4] workload rides on the default line as a compact object
            c5 = res["c5"]
            assert c5["dtype"] == "u16" and c5["value"] > 0 and c5["roofline"]["isovalues_per_launch"] == 4 and c5["roofline"]["frac"] > 0
            assert c5["roofline"]["step_frac"] > 0 and c5["cpu_baseline"]["value"] > 0 and "all 8 isovalues" in c5["cpu_baseline"]["sample"]
        else:
            assert "c5" not in res and res["roofline"]["whole_call_reference_equivalent"]["frac"] > res["roofline"]["step_frac"]


def test_programs_built_against_the_reference_header(launcher):
    """Drop-in at the binary level: tests/callers/caller.c was compiled in the build container against the REFERENCE's
    marching_cubes_33.h (oracle/_ref/callers/, built by __graft_entry__.build()), once linked with the product library
    and once with the reference library.  Run as child processes here: same struct layout, same counts, same digests of
    T, V and N - for every sample type, with and without GRD_ORTHOGONAL."""
    d = os.path.join(ROOT, "oracle", "_ref", "callers")
    if not os.path.isdir(d):
        pytest.skip("oracle/_ref/callers not built (needs the reference header in the build container)")
    names = sorted(n[len("refhdr_"):] for n in os.listdir(d) if n.startswith("refhdr_"))
    assert len(names) == 10
    compared = 0
    for n in names:
        got = launcher.run([os.path.join(d, "refhdr_" + n)], timeout=300)
        assert got["rc"] == 0 and "digest T" in got["stdout"], (n, got)
        nv = int(got["stdout"].split("surface nV ")[1].split()[0])
        assert nv > 5000, got["stdout"]
        want_path = os.path.join(d, "reflib_" + n)
        if os.path.exists(want_path):
            want = launcher.run([want_path], timeout=300)
            assert want["rc"] == 0
            # (capacities after the reference's doubling growth differ from exact-fit ones before adjustvectorlenght_s; the
            # lines compared are layout, surface counts, digests and the state after the adjust)
            assert got["stdout"] == want["stdout"], (n, got["stdout"], want["stdout"])
            compared += 1
    assert compared >= 7


@pytest.mark.parametrize("gather", ["allgather", "pairs", "root"])
def test_bench_rccl_code_path_with_one_rank(launcher, gather):
    """The N > 1 branch of bench.py with the REAL backend (RCCL through torch.distributed's "nccl"), as far as a one-GPU
    box allows: MC33_BENCH_FORCE_DIST=1 runs one rank through init_process_group("nccl"), the second communicator of
    the count exchange, the device-tensor collectives and the overlapped (async) exchange of two buffer sets."""
    for args in (["--points", "160"], ["--config", "c5", "--points", "64"]):
        res, out = bench(launcher, ["--steps", "4", "--warmup", "2", "--gather", gather, "--no-cpu-baseline"] + args,
                         {"MC33_BENCH_REHEARSAL": "0", "MC33_BENCH_FORCE_DIST": "1", "MC33_BENCH_VERIFY": "1", "MASTER_PORT": str(PORT[0] + 500)})
        PORT[0] += 1
        assert res["n_gpus"] == 1 and res["gather"]["mode"] == gather and res["gather"]["overlapped_with_next_extraction"] is True
        assert "equals whole-volume result: True" in out["stderr"]
