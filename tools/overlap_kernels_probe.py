#!/usr/bin/env python3
"""Developer probe: would the tail + emit of one z-chunk hide beside the sweep of the next?

Two contexts on ONE resident grid, a stream each.  `sweep_many([iso])` enqueues nothing but k_sweep; an extract call for an
isovalue that has been swept ahead runs only k_boundary ... k_emit_*.  So, with no new code in the library:

    sweep alone        : A.sweep_many                                (k_sweep of the whole grid)
    tail + emit alone  : B.extract of an isovalue swept ahead        (k_boundary ... k_scan_apply, the emit kernels)
    both               : A.sweep_many on stream A, B's tail + emit on stream B, issued together

which is the steady state a z-chunk pipeline would run in (chunk k's tail + emit beside chunk k + 1's sweep), at full size.
The pipeline pays if `both` is shorter than the sum by more than its own overheads (~0.1 ms at 1024^3).

    python tools/overlap_kernels_probe.py [f32|u16] [reps] [priority]
        priority: 0 = two plain streams, 1 = the tail + emit stream has high priority
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prio = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
if kind == "f32":
    grid, r0, d = fields.cos_field_cube(1024, dev)
    iso = 0.0
else:
    grid = fields.cos_field_u16(2048, 2048, 1024, dev)
    r0, d = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
    iso = 30268.5
sa = torch.cuda.Stream(dev)
sb = torch.cuda.Stream(dev, priority=-1) if prio else torch.cuda.Stream(dev)
with torch.cuda.stream(sa):
    A = api.DeviceGrid(grid, r0=r0, d=d)
    A.use_stream(sa)
with torch.cuda.stream(sb):
    B = api.DeviceGrid(grid, r0=r0, d=d)
    B.use_stream(sb)
A.set_timing(0)
B.set_timing(0)
cnt = B.count(iso)
V = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev)
N = torch.empty_like(V)
T = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)
A.count(iso)  # (workspaces of A)


def sync():
    torch.cuda.synchronize()


def t_sweep():
    sync()
    t0 = time.perf_counter()
    A.sweep_many([iso])
    sa.synchronize()
    return (time.perf_counter() - t0) * 1e3


def t_tail():
    B.sweep_many([iso])
    sync()
    t0 = time.perf_counter()
    B.extract_into(iso, V, N, T)  # finds its sweep made: tail + emit only; returns after its stream's synchronisation
    return (time.perf_counter() - t0) * 1e3


def t_both():
    B.sweep_many([iso])
    sync()
    t0 = time.perf_counter()
    A.sweep_many([iso])           # k_sweep on stream A (asynchronous)
    B.extract_into(iso, V, N, T)  # tail + emit on stream B
    t_b = (time.perf_counter() - t0) * 1e3
    sa.synchronize()
    return (time.perf_counter() - t0) * 1e3, t_b


def t_whole():
    sync()
    t0 = time.perf_counter()
    B.extract_into(iso, V, N, T)
    return (time.perf_counter() - t0) * 1e3


def med(a):
    a = sorted(a)
    return a[len(a) // 2]


for _ in range(3):
    t_sweep(); t_tail(); t_both(); t_whole()
sw = med([t_sweep() for _ in range(reps)])
tl = med([t_tail() for _ in range(reps)])
bo = [t_both() for _ in range(reps)]
both, tail_in_both = med([x[0] for x in bo]), med([x[1] for x in bo])
wh = med([t_whole() for _ in range(reps)])
print("%s grid, %d reps, medians, host clock around one synchronised call each (priority stream for the tail: %s)" % (kind, reps, "yes" if prio else "no"))
print("  whole extraction, one call          : %.3f ms" % wh)
print("  k_sweep alone                       : %.3f ms" % sw)
print("  tail + emit alone (swept ahead)     : %.3f ms" % tl)
print("  sum                                 : %.3f ms" % (sw + tl))
print("  both, two streams                   : %.3f ms   (the tail + emit inside it: %.3f ms)" % (both, tail_in_both))
print("  gain of the pair over the sum       : %.3f ms" % (sw + tl - both))
