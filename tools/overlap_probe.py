#!/usr/bin/env python3
"""Developer probe: do two extractions on two streams overlap (the sweep of one with the cells/scan/emit tail of the
other)?  Two z-halves of the 1024^3 bench field, each its own context and stream, driven from two host threads;
compared with the same two extractions one after the other and with the whole volume in one call."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402
from mc33_c_library_amd.api import Range  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
grid, r0, d = fields.cos_field_cube(n, dev)
nz = n - 1
whole = api.DeviceGrid(grid, r0=r0, d=d)
cnt = whole.count(0.0)
V = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev); N = torch.empty_like(V)
T = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("whole volume, one call: %.3f ms" % timeit(lambda: whole.extract_into(0.0, V, N, T)))
cuts = [nz * k // parts for k in range(parts + 1)]
grids, streams, bufs = [], [], []
all_streams = [torch.cuda.Stream(dev) for _ in range(parts)]  # created back to back: consecutive hardware queues
for st in all_streams:
    with torch.cuda.stream(st):
        torch.zeros(1, device=dev)
for k in range(parts):
    s = all_streams[k]
    with torch.cuda.stream(s):
        g = api.DeviceGrid(grid, r0=r0, d=d)
        g.use_stream(s)
    rng = Range(cuts[k], cuts[k + 1], 1 if k else 0, 0)
    c = g.count(0.0, rng)
    bufs.append((torch.empty((c.nV + 1024, 3), dtype=torch.float32, device=dev), torch.empty((c.nV + 1024, 3), dtype=torch.float32, device=dev),
                 torch.empty((c.nT + 1024, 3), dtype=torch.int32, device=dev), rng))
    grids.append(g); streams.append(s)


def one(k):
    Vk, Nk, Tk, rng = bufs[k]
    grids[k].extract_into(0.0, Vk, Nk, Tk, rng)


def sequential():
    for k in range(parts):
        one(k)


def concurrent():
    th = [threading.Thread(target=one, args=(k,)) for k in range(parts)]
    for t in th:
        t.start()
    for t in th:
        t.join()


def staggered():  # part k starts when part k-1 is about half way: its sweep meets the other's tail
    th = []
    for k in range(parts):
        t = threading.Thread(target=one, args=(k,))
        t.start()
        th.append(t)
        time.sleep(0.0003 * 2 / parts)
    for t in th:
        t.join()


print("%d parts one after the other: %.3f ms" % (parts, timeit(sequential)))
print("%d parts from %d threads at once: %.3f ms" % (parts, parts, timeit(concurrent)))
print("%d parts staggered: %.3f ms" % (parts, timeit(staggered)))
