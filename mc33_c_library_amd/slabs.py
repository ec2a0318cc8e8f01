"""Host-side orchestration of the z-slab decomposition: one process per GPU, rank p owns the
cell slices [z_p, z_{p+1}) plus one ghost slice below, the ranks exchange their vertex / triangle counts and then
the surface arrays.  Plumbing only (torch.distributed over RCCL, or gloo on host copies when several ranks
rehearse on one GPU); the extraction is done by the HIP kernels behind `DeviceGrid`.

Exchange modes of the surface arrays (V, N: 12 B per vertex, T: 12 B per triangle):

  allgather  every rank receives every surface, each rank's rows padded to the longest (north_star: "RCCL
             all-gather ... to concatenate the per-rank surface arrays"); RCCL has no all-gather-v
  pairs      the same result without padding: every rank emits straight into its place of the concatenated arrays
             and the ranks exchange their pieces all-pairs with one grouped batch of send / recv - one hop over the
             direct xGMI link of each pair
  root       only rank 0 - the one that hands the `surface` to the caller of calculate_isosurface (reference
             GLUT_example/TestMC33_glut.c:421-458) - receives the pieces: 1/world of the traffic

In `pairs` and `root` the concatenated arrays ARE the single-GPU result (rank order = z order, ids rebased on the
device by the emit pass): nothing is copied or re-indexed after the exchange.
"""
from .api import Range

MODES = ("allgather", "pairs", "root")


class Slab:
    """Cell slices [z_begin, z_end) of one rank, its ghost slice, and the sample planes [p_lo, p_hi] it keeps
    resident: the cells' own planes, one above for the central differences of the normals (reference
    marching_cubes_33.c:888-890, 1036-1038), the ghost slice's lower plane and one below that for vertices on grid
    points (MC:643-647)."""

    def __init__(self, rank, world, nz_total, per=None):
        self.rank, self.world, self.nz_total = rank, world, nz_total
        if per:  # fixed number of slices per rank (the last rank takes what is left)
            self.z_begin, self.z_end = min(rank * per, nz_total), min((rank + 1) * per, nz_total)
        else:    # even split
            self.z_begin, self.z_end = rank * nz_total // world, (rank + 1) * nz_total // world
        if self.z_begin >= self.z_end:
            raise ValueError("rank %d of %d has no cell slice (nz = %d)" % (rank, world, nz_total))
        self.ghost = 1 if rank else 0
        self.p_lo = max(self.z_begin - self.ghost - 1, 0)
        self.p_hi = min(self.z_end + 1, nz_total)

    @property
    def planes(self):
        return self.p_hi - self.p_lo + 1

    def range(self, id_base=0):
        return Range(self.z_begin, self.z_end, self.ghost, id_base)


class SurfaceExchange:
    """Buffers and collectives of one rank.  nbuf = 2 lets the exchange of one surface run while the next one is
    extracted into the other set."""

    def __init__(self, world, rank, device, capV, capT, mode="allgather", nbuf=1, host_collectives=False, count_group=None):
        import torch
        assert mode in MODES
        self.world, self.rank, self.dev, self.mode, self.nbuf = world, rank, device, mode, nbuf
        self.host = host_collectives          # rehearsal: gloo on host copies (NCCL refuses two ranks on one device)
        self.count_group = count_group        # own communicator: the tiny count exchange must not queue behind surfaces
        self.capV, self.capT = capV, capT
        f32, i32 = torch.float32, torch.int32
        # flat [world * cap, 3]: read as [world, rows, 3] (allgather) or as the concatenated arrays (pairs / root)
        self.gV = [torch.empty((world * capV, 3), dtype=f32, device=device) for _ in range(nbuf)]
        self.gN = [torch.empty((world * capV, 3), dtype=f32, device=device) for _ in range(nbuf)]
        self.gT = [torch.empty((world * capT, 3), dtype=i32, device=device) for _ in range(nbuf)]
        if mode == "allgather":  # the rank's own rows, gathered from here
            self.V = [torch.empty((capV, 3), dtype=f32, device=device) for _ in range(nbuf)]
            self.N = [torch.empty((capV, 3), dtype=f32, device=device) for _ in range(nbuf)]
            self.T = [torch.empty((capT, 3), dtype=i32, device=device) for _ in range(nbuf)]
        self.counts_dev = torch.zeros(world * 2, dtype=torch.int64, device="cpu" if host_collectives else device)
        self.mine_dev = torch.zeros(2, dtype=torch.int64, device="cpu" if host_collectives else device)  # this rank's pair, written by a kernel (extract_slab_on_device)
        self.pending = [[] for _ in range(nbuf)]
        self.rows = [(0, 0)] * nbuf
        self.bytes_received = 0

    # -- counts ------------------------------------------------------------------------------------------
    def exchange_counts(self, nV, nT):
        """[(nV, nT)] of all ranks."""
        import torch
        import torch.distributed as dist
        mine = torch.tensor([nV, nT], dtype=torch.int64, device=self.counts_dev.device)
        dist.all_gather_into_tensor(self.counts_dev, mine, group=None if self.host else self.count_group)
        c = self.counts_dev.view(self.world, 2).tolist()
        return [(int(a), int(b)) for a, b in c]

    def exchange_counts_many(self, mine):
        """mine: [(nV, nT)] of this rank for k isovalues -> per isovalue [(nV, nT)] of all ranks, in ONE collective (an iso sweep
        over z-slabs then has one host-synchronised exchange per step instead of one per isovalue)."""
        import torch
        import torch.distributed as dist
        k = len(mine)
        dev = self.counts_dev.device
        if getattr(self, "_many", None) is None or self._many.numel() != self.world * 2 * k:
            self._many = torch.zeros(self.world * 2 * k, dtype=torch.int64, device=dev)
        flat = torch.tensor([x for c in mine for x in c], dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(self._many, flat, group=None if self.host else self.count_group)
        rows = self._many.view(self.world, k, 2).tolist()
        return [[(int(rows[r][i][0]), int(rows[r][i][1])) for r in range(self.world)] for i in range(k)]

    @staticmethod
    def bases(counts, rank):
        return sum(c[0] for c in counts[:rank]), sum(c[1] for c in counts[:rank])

    # -- where the emit pass of this rank writes -----------------------------------------------------------
    def targets(self, b, counts):
        """(V, N, T) device tensors the rank's emit pass fills for buffer set b."""
        if self.mode == "allgather":
            return self.V[b], self.N[b], self.T[b]
        vb, tb = self.bases(counts, self.rank)
        nV, nT = counts[self.rank]
        if sum(c[0] for c in counts) > self.world * self.capV or sum(c[1] for c in counts) > self.world * self.capT:
            raise RuntimeError("surface larger than the exchange buffers")
        return self.gV[b][vb:vb + nV], self.gN[b][vb:vb + nV], self.gT[b][tb:tb + nT]

    # -- the exchange --------------------------------------------------------------------------------------
    def wait(self, b):
        for w in self.pending[b]:
            w.wait()
        self.pending[b] = []

    def drain(self):
        for b in range(self.nbuf):
            self.wait(b)

    def start(self, b, counts, async_op=False):
        """Exchange the surface arrays of buffer set b (after the emit pass was enqueued on the current stream)."""
        import torch
        import torch.distributed as dist
        world, rank = self.world, self.rank
        if self.mode == "allgather":
            rv, rt = max(c[0] for c in counts), max(c[1] for c in counts)  # rows per rank: the longest, not the capacity
            if rv > self.capV or rt > self.capT:
                raise RuntimeError("surface larger than the exchange buffers (the capacity must be the same on all ranks)")
            self.rows[b] = (rv, rt)
            self.bytes_received = (world - 1) * (rv * 24 + rt * 12)
            pairs = [(self.gV[b][:world * rv], self.V[b][:rv]), (self.gN[b][:world * rv], self.N[b][:rv]), (self.gT[b][:world * rt], self.T[b][:rt])]
            for out, inp in pairs:
                if inp.numel() == 0:
                    continue
                if self.host:
                    o = torch.empty(out.shape, dtype=out.dtype)
                    dist.all_gather_into_tensor(o.view(-1), inp.reshape(-1).cpu())
                    out.copy_(o)
                else:
                    w = dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), async_op=async_op)
                    if async_op:
                        self.pending[b].append(w)
            return
        # pairs / root: variable-length pieces straight into their place of the concatenated arrays
        vbase = [sum(c[0] for c in counts[:r]) for r in range(world)]
        tbase = [sum(c[1] for c in counts[:r]) for r in range(world)]

        def piece(r):
            nV, nT = counts[r]
            return [t for t in (self.gV[b][vbase[r]:vbase[r] + nV], self.gN[b][vbase[r]:vbase[r] + nV], self.gT[b][tbase[r]:tbase[r] + nT]) if t.numel()]

        receivers = range(world) if self.mode == "pairs" else (0,)
        sends = [(dst, t) for dst in receivers if dst != rank for t in piece(rank)]
        recvs = [(src, t) for src in range(world) if src != rank and rank in receivers for t in piece(src)]
        self.bytes_received = sum(t.numel() * 4 for _, t in recvs)
        if self.host:
            host_r = [(src, torch.empty(t.shape, dtype=t.dtype)) for src, t in recvs]
            ops = [dist.P2POp(dist.isend, t.cpu().contiguous(), dst) for dst, t in sends] + [dist.P2POp(dist.irecv, h, src) for src, h in host_r]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for (_, t), (_, h) in zip(recvs, host_r):
                t.copy_(h)
            return
        ops = [dist.P2POp(dist.isend, t, dst) for dst, t in sends] + [dist.P2POp(dist.irecv, t, src) for src, t in recvs]
        if not ops:
            return
        works = dist.batch_isend_irecv(ops)  # one grouped ncclSend / ncclRecv batch
        if async_op:
            self.pending[b].extend(works)
        else:
            for w in works:
                w.wait()

    def start_by_capacity(self, b, async_op=False):
        """`allgather` exchange of buffer set b with every rank's rows = the CAPACITY of the buffers (the same on every rank):
        needs no count on the host, moves the few per cent of slack with it (extract_slab_on_device)."""
        import torch.distributed as dist
        assert self.mode == "allgather" and not self.host
        self.rows[b] = (self.capV, self.capT)
        self.bytes_received = (self.world - 1) * (self.capV * 24 + self.capT * 12)
        for out, inp in ((self.gV[b], self.V[b]), (self.gN[b], self.N[b]), (self.gT[b], self.T[b])):
            w = dist.all_gather_into_tensor(out.view(-1), inp.reshape(-1), async_op=async_op)
            if async_op:
                self.pending[b].append(w)

    # -- the result ----------------------------------------------------------------------------------------
    def concatenated(self, b, counts):
        """(V, N, T) of the whole volume on this rank (every rank; in `root` mode only rank 0 holds them)."""
        import torch
        nV, nT = sum(c[0] for c in counts), sum(c[1] for c in counts)
        if self.mode != "allgather":
            return self.gV[b][:nV], self.gN[b][:nV], self.gT[b][:nT]
        rv, rt = self.rows[b]
        gv, gn, gt = (self.gV[b][:self.world * rv].view(self.world, rv, 3), self.gN[b][:self.world * rv].view(self.world, rv, 3),
                      self.gT[b][:self.world * rt].view(self.world, rt, 3))
        return (torch.cat([gv[r, :counts[r][0]] for r in range(self.world)]), torch.cat([gn[r, :counts[r][0]] for r in range(self.world)]),
                torch.cat([gt[r, :counts[r][1]] for r in range(self.world)]))


def extract_slab_on_device(grid, slab, exchange, iso, b=0, async_op=False):
    """extract_slab with the counts kept on the device between count and emit (`allgather` mode over RCCL): the slab's
    {vertices, triangles} are written into a device tensor by a kernel behind the count, gathered by all_gather_into_tensor,
    turned into this rank's vertex id base by a one-thread kernel and read by the emit passes from there; the surface arrays
    are gathered by capacity (the same on every rank, a few per cent above the counts) instead of by the longest rank's rows.
    Count, count exchange, emit and surface exchange are ENQUEUED back to back; only then does the host wait - for the gathered
    table and this rank's counters.  (The host-side flow waits for its counters, runs a collective, reads the result back and
    only then enqueues the emit: two host round trips and a collective inside every extraction.)
    Returns (counts of all ranks, this rank's Counts), or None when this rank's work records did not fit - room has been made,
    the caller repeats the step through extract_slab (every rank sees the same table, but only this rank its own overflow: the
    caller decides collectively, see bench.py)."""
    import torch.distributed as dist
    assert exchange.mode == "allgather" and not exchange.host
    exchange.wait(b)
    grid.count_async(iso, slab.range())
    grid.counts_to_device(exchange.mine_dev)
    dist.all_gather_into_tensor(exchange.counts_dev, exchange.mine_dev, group=exchange.count_group)
    grid.bases_from_table(exchange.counts_dev, 2, slab.rank, False)
    grid.emit_at_device_bases(exchange.V[b], exchange.N[b], exchange.T[b])
    exchange.start_by_capacity(b, async_op=async_op)
    table = exchange.counts_dev.view(exchange.world, 2).tolist()   # (the one wait of the step: everything above is enqueued)
    c, ok = grid.count_finish()
    counts = [(int(a), int(t)) for a, t in table]
    return (counts, c) if ok else None


def extract_slab(grid, slab, exchange, iso, b=0, async_op=False):
    """One extraction of a z-slabbed volume on this rank: count -> exchange counts -> emit with the global vertex
    base -> exchange the surface arrays.  Returns (counts of all ranks, this rank's Counts)."""
    exchange.wait(b)
    c = grid.count(iso, slab.range())
    counts = exchange.exchange_counts(c.nV, c.nT)
    id_base, _ = exchange.bases(counts, slab.rank)
    V, N, T = exchange.targets(b, counts)
    grid.emit_into(V, N, T, id_base)
    exchange.start(b, counts, async_op=async_op)
    return counts, c


def extract_slab_many(grid, slab, exchange, isos, b0=0, async_op=False, on_emitted=None):
    """An iso sweep over a z-slabbed volume on this rank (BASELINE configs[4] on N GPUs): the slab is streamed once per 4
    isovalues and everything a count needs is made behind each pass (DeviceGrid.prepare_many), the counts of ALL isovalues
    are exchanged in ONE collective, then every isovalue is emitted at its global vertex base and its surface arrays are
    exchanged (buffer sets b0, b0 + 1, ... in turn; on_emitted(i) is called behind each isovalue's emit).  Returns [(counts of all ranks, this rank's Counts)] per isovalue."""
    if len(isos) > 8:  # the library prepares at most 8 isovalues at a time (two passes over the slab, 4 isovalues each): in groups
        out = []
        for k in range(0, len(isos), 8):
            out += extract_slab_many(grid, slab, exchange, isos[k:k + 8], b0=b0 + k, async_op=async_op,
                                     on_emitted=(lambda i, k=k: on_emitted(k + i)) if on_emitted is not None else None)
        return out
    grid.prepare_many(isos, slab.range())
    mine = [grid.count(iso, slab.range()) for iso in isos]           # (made already: these only fetch the counters)
    allc = exchange.exchange_counts_many([(c.nV, c.nT) for c in mine])
    out = []
    for i, iso in enumerate(isos):
        b = (b0 + i) % exchange.nbuf
        exchange.wait(b)
        counts = allc[i]
        grid.count(iso, slab.range())                                # selects the isovalue's buffers for the emit (nothing is computed)
        id_base, _ = exchange.bases(counts, slab.rank)
        V, N, T = exchange.targets(b, counts)
        grid.emit_into(V, N, T, id_base)
        exchange.start(b, counts, async_op=async_op)
        if on_emitted is not None:
            on_emitted(i)
        out.append((counts, mine[i]))
    return out
