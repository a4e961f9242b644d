"""Block-partitioned stepping across ranks: one process per GPU, torch.distributed for the exchange.

The reference's only distributed design is root-centric MPI: every step rank 0 broadcasts nine
n-float arrays and gathers three acceleration slices back (ver5_all/GSimulation.cpp:170-214,
cpu/Compute.cpp:47-58, 95-97); its OpenCL back end splits the i range across devices the same
way (opencl/Compute.cpp:241-255, 273).  Here the same i-block partition is kept but each rank
owns its bodies for good: it integrates them itself, velocities never travel, and the only
exchange per time step is ONE in-place all-gather of the freshly integrated {x,y,z,G*m} blocks
(RCCL over xGMI when the backend is "nccl").  Kinetic energy is a scalar all-reduce, done only
when a caller asks for it.

PyTorch is plumbing here: rendezvous, the collective, barriers.  The compute is libnbx.
"""
import ctypes


BLOCK_ALIGN = 256  # = the kernels' j tile: keeps every rank's block a whole number of tiles


def block_partition(n, world, rank, align=BLOCK_ALIGN):
    """Balanced block partition with equal, tile-aligned blocks.

    Returns (block, i_begin, i_count, n_alloc).  All ranks hold n_alloc = world*block records;
    records >= n are zero-mass padding (their pair terms are exactly 0).  Unlike the reference's
    slices (cpu/Compute.cpp:50-51, wrong unless n % size == 0) any n works; trailing ranks may
    own fewer real bodies.  A world so large that trailing ranks would own NOTHING is the caller's
    to refuse (ShardedSimulation does, identically on every rank: see check_world).
    """
    if n <= 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad partition arguments n=%r world=%r rank=%r" % (n, world, rank))
    block = -(-n // world)
    block = -(-block // align) * align
    i_begin = min(rank * block, n)
    i_count = max(0, min(n, (rank + 1) * block) - i_begin)
    return block, i_begin, i_count, world * block


def check_world(n, world, align=BLOCK_ALIGN):
    """Raise the SAME ValueError on every rank when `world` ranks cannot all own bodies.

    The partition is a pure function of (n, world), so each rank can decide alone, before any
    engine or collective exists: nobody is left blocking in an all-gather while another rank has
    already raised (n = 1500 on 8 ranks: block = 256, ranks 6 and 7 would be empty).  The native
    nbx_group_create drops empty ranks instead; with one process per rank the launcher fixed the
    world size, so the only clean answer is to refuse it.
    """
    block = block_partition(n, world, 0, align)[0]
    used = -(-n // block)
    if used < world:
        raise ValueError("n=%d bodies fill only %d blocks of %d: a world of %d ranks would leave %d rank(s) without bodies; "
                         "launch at most %d ranks" % (n, used, block, world, world - used, used))


def _stat3(xs):
    xs = [float(x) for x in xs]
    return {"min": min(xs), "mean": sum(xs) / len(xs), "max": max(xs)} if xs else None


def gather_rank_reports(dist, report):
    """Every rank contributes one dict; every rank gets the list in rank order (all_gather_object: works over RCCL and gloo)."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, report)
    return out


def summarise_rank_reports(reports, block_bytes):
    """bench.py's N > 1 breakdown from the per-rank reports (dicts with rank, device, force_ms_mean, allgather_ms (list),
    elapsed_s, bodies_owned): min / mean / max over ranks of the force kernel and of the all-gather, and the skew.

    allgather_ms on a rank = the collective + the wait for the slowest rank (see ShardedSimulation.profile_exchange), so
    its minimum over ranks estimates the collective alone and `skew_ms` (slowest minus fastest force kernel) says how much
    of the rest is waiting.  What the reference does at this point -- nine n-float broadcasts and three gathers through
    rank 0 per step, ver5_all/GSimulation.cpp:170-214 -- has no breakdown at all."""
    world = len(reports)
    reports = sorted(reports, key=lambda r: r["rank"])
    if [r["rank"] for r in reports] != list(range(world)):
        raise ValueError("rank reports do not cover 0..%d exactly once: %r" % (world - 1, [r["rank"] for r in reports]))
    force = [r["force_ms_mean"] for r in reports]
    ag_mean = [sum(r["allgather_ms"]) / len(r["allgather_ms"]) if r["allgather_ms"] else 0.0 for r in reports]
    ag_max = [max(r["allgather_ms"]) if r["allgather_ms"] else 0.0 for r in reports]
    return {
        "world_seen": world,
        "force_kernel_ms": _stat3(force),
        "allgather_ms_per_step": _stat3(ag_mean),
        "allgather_ms_worst_step": max(ag_max) if ag_max else None,
        "skew_ms": max(force) - min(force),
        "slowest_rank": max(range(world), key=lambda r: force[r]),
        "elapsed_s": _stat3([r["elapsed_s"] for r in reports]),
        "bytes_gathered_per_step": block_bytes * (world - 1),  # received by every rank: the other ranks' blocks
        "bytes_sent_per_step": block_bytes,
        "per_rank": [{"rank": r["rank"], "device": r.get("device"), "host": r.get("host"), "bodies_owned": r.get("bodies_owned"),
                      "force_ms": r["force_ms_mean"], "allgather_ms": a} for r, a in zip(reports, ag_mean)],
    }


class _DeviceBuffer:
    """Zero-copy view of a raw device pointer for torch.as_tensor (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2, "strides": None,
        }


def _nbx():
    """Import nbx.py by path (the package directory name contains '-')."""
    import importlib.util
    import os
    import sys
    if "nbx" in sys.modules:
        return sys.modules["nbx"]
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("nbx", os.path.join(here, "nbx.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nbx"] = mod
    spec.loader.exec_module(mod)
    return mod


class NbxEngine:
    """libnbx context of one rank (the product path): owns [i_begin, i_begin+i_count)."""

    def __init__(self, n, precision, i_begin, i_count, n_alloc, **opts):
        import torch
        nbx = _nbx()
        self.torch = torch
        self.precision = precision
        # the context enqueues on torch's CURRENT stream (handle 0 = the default stream), so the
        # collectives torch issues on that stream are ordered with the kernels
        self.stream = torch.cuda.current_stream()
        if i_count <= 0:
            raise ValueError("a rank with no bodies cannot build a context (n too small for this world size)")
        self.ctx = nbx.Context(n, precision, i_begin=i_begin, i_count=i_count, n_alloc=n_alloc,
                               stream=ctypes.c_void_p(self.stream.cuda_stream), external_stream=1, **opts)
        self._views = {}

    def upload(self, state):
        self.ctx.upload(state)

    def step_local(self, dt):
        self.ctx.step_local(dt)

    def exchange_tensor(self):
        """uint8 tensor aliasing the whole NEXT position buffer (n_alloc records)."""
        ptr, total, _, _ = self.ctx.exchange_buffer()
        t = self._views.get(ptr)
        if t is None:
            t = self.torch.as_tensor(_DeviceBuffer(ptr, total), device="cuda")
            self._views[ptr] = t
        return t

    def commit(self):
        self.ctx.commit()

    def kenergy_partial(self):
        return self.ctx.kenergy_partial()

    def download(self):
        return self.ctx.download()

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.ctx.close()


class ShardedSimulation:
    """The reference's time-step loop (ver7/GSimulation.cpp:138-200) over `world` ranks.

    engine_factory(n, precision, i_begin, i_count, n_alloc) builds the rank's compute engine
    (default: NbxEngine = libnbx on the rank's GPU).  `dist` is torch.distributed, already
    initialised (nccl == RCCL on ROCm; gloo for CPU rehearsals), or None for a single rank.
    """

    def __init__(self, n, precision=32, dist=None, engine_factory=None, force_collective=False, weights=None, **opts):
        """weights (one positive number per rank, the same list on every rank): unequal shares in whole 256-record tiles
        (nbx_partition_weighted: the reference's `cpu+gpu <ratio>` split with GPUs as the devices, include/nbx.h); the per-step
        exchange is then one in-place broadcast per owner instead of the equal-block all-gather."""
        self.n = int(n)
        self.precision = precision
        # a 1-rank group skips the collectives unless force_collective (used to rehearse the RCCL path on one GPU)
        use = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force_collective)
        self.dist = dist if use else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.shares = None  # weighted form: [(i_begin, i_count)] of every rank
        if weights is not None and self.world > 1:
            if len(weights) != self.world:
                raise ValueError("weights needs one number per rank (%d), got %d" % (self.world, len(weights)))
            parts = [_nbx().partition_weighted(self.n, self.world, list(weights), r) for r in range(self.world)]  # host arithmetic of libnbx
            if parts[0][0] < self.world:  # the same verdict on every rank, before any engine or collective exists
                raise ValueError("n=%d bodies are %d tiles of 256 records: a world of %d ranks would leave rank(s) without bodies"
                                 % (self.n, parts[0][0], self.world))
            self.shares = [(p[1], p[2]) for p in parts]
            self.i_begin, self.i_count = self.shares[self.rank]
            self.n_alloc = parts[0][3]
            self.block = max(c for _, c in self.shares)  # the largest share (reports); blocks differ in size
        else:
            check_world(self.n, self.world)  # collective decision: raises on all ranks or on none
            self.block, self.i_begin, self.i_count, self.n_alloc = block_partition(self.n, self.world, self.rank)
        self.rec = 16 if precision == 32 else 32
        factory = engine_factory or NbxEngine
        self.engine = factory(self.n, precision, self.i_begin, self.i_count, self.n_alloc, **opts)
        self.steps_done = 0
        self.bytes_gathered = 0
        self._staged = False      # out-of-place fallback of the all-gather (see _all_gather_in_place)
        self._stage_buf = None
        self._xprof = None        # profile_exchange(True): per-step durations of the all-gather on this rank

    def upload(self, state):
        self.engine.upload(state)

    def _all_gather_in_place(self, full):
        """In-place all-gather: rank r contributes full[r*block : (r+1)*block] (in records)."""
        import torch
        if self.shares is not None:
            # unequal shares: every owner broadcasts its block in place (P small collectives instead of one all-gather)
            staged = full.is_cuda and self.dist.get_backend() != "nccl"  # rehearsal only: several ranks sharing one GPU under gloo
            for r, (b, c) in enumerate(self.shares):
                blk = full[b * self.rec:(b + c) * self.rec]
                if staged:
                    host = blk.cpu()
                    self.dist.broadcast(host, src=r)
                    if r != self.rank:
                        blk.copy_(host)
                else:
                    self.dist.broadcast(blk, src=r)
            self.bytes_gathered += (self.n - self.i_count) * self.rec
            return
        nb = self.block * self.rec
        own = full[self.rank * nb:(self.rank + 1) * nb]
        if full.is_cuda and self.dist.get_backend() != "nccl":
            # rehearsal only (several ranks sharing one GPU under gloo): stage through the host
            host = torch.empty(full.numel(), dtype=full.dtype)
            self.dist.all_gather_into_tensor(host, own.cpu())
            full.copy_(host)
        elif not self._staged:
            try:
                self.dist.all_gather_into_tensor(full, own)
            except (RuntimeError, ValueError) as e:
                # an argument-level refusal of the aliasing output (never seen; the 1-rank RCCL rehearsal accepts it):
                # keep the run alive with an out-of-place gather + one device copy, and say so once
                import sys
                print("sharded: in-place all-gather refused (%s); staging through a second buffer" % e, file=sys.stderr)
                self._staged = True
        if self._staged and not (full.is_cuda and self.dist.get_backend() != "nccl"):
            if self._stage_buf is None:
                self._stage_buf = torch.empty_like(full)
            self.dist.all_gather_into_tensor(self._stage_buf, own.clone())
            full.copy_(self._stage_buf)
        self.bytes_gathered += nb * (self.world - 1)

    def profile_exchange(self, enable=True):
        """Time every all-gather from now on (bench.py's N > 1 breakdown).  On a device buffer over "nccl" the bracket
        is a pair of events on the stream the force kernel and the collective are ordered on: the start fires when this
        rank's force launch has finished, the end when the gathered array is complete -- the collective itself PLUS the
        wait for the slowest rank.  The smallest value over the ranks is therefore the collective's own cost, the spread
        the skew.  Anything else (CPU rehearsals, the host-staged gloo rehearsal) is timed on the host clock after
        draining this rank's own work."""
        self._xprof = {"events": [], "host_ms": []} if enable else None

    def exchange_ms(self):
        """Per-step all-gather durations recorded since profile_exchange(True), in milliseconds (synchronises)."""
        if not self._xprof:
            return []
        out = list(self._xprof["host_ms"])
        if self._xprof["events"]:
            import torch
            torch.cuda.synchronize()
            out += [a.elapsed_time(b) for a, b in self._xprof["events"]]
        return out

    def _exchange(self):
        full = self.engine.exchange_tensor()
        xp = self._xprof
        if xp is None:
            return self._all_gather_in_place(full)
        import time
        import torch
        if full.is_cuda and self.dist.get_backend() == "nccl":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._all_gather_in_place(full)
            b.record()
            xp["events"].append((a, b))
        else:
            self.engine.sync()
            t0 = time.perf_counter()
            self._all_gather_in_place(full)
            xp["host_ms"].append(1e3 * (time.perf_counter() - t0))

    def step(self, nsteps=1, dt=None):
        dt = _nbx().DT if dt is None else dt
        for _ in range(nsteps):
            self.engine.step_local(dt)
            if self.dist:
                self._exchange()
            self.engine.commit()
            self.steps_done += 1

    def kenergy(self):
        """_kenergy of ver7/GSimulation.cpp:200 after the last step (synchronises)."""
        import torch
        part = float(self.engine.kenergy_partial())
        if self.dist:
            t = torch.tensor([part], dtype=torch.float64)
            if self.dist.get_backend() == "nccl":
                t = t.cuda()
            self.dist.all_reduce(t)
            part = float(t.item())
        return 0.5 * part

    def sync(self):
        self.engine.sync()

    def close(self):
        self.engine.close()
