"""Halo layer: facet-trace exchange between mesh blocks, one process per GPU.

Replaces the implicit PyOP2/MPI halo exchange that runs inside every
``assemble`` of the reference (``seigen/elastic.py:364``, ``:404-436``;
``ParLoopHaloEnd`` in ``tests/tiling/utils.py:144``).  DG couples cells only
through facets (the ``dS`` terms, ``elastic.py:206``, ``:213-215``), so per
stage each block sends the traces of a field on its block sides to its face
neighbours (`torch.distributed` point-to-point: RCCL over xGMI on GPUs, gloo in
the CPU tests) and meanwhile computes cells that nobody is waiting for.

The exchange is pipelined across stages: the input of every stage is the output
of the stage before it (STAGE_INPUT / STAGE_OUTPUT below), so a stage first
runs SG_REGION_FIRST - the shell next to the neighbours together with half of
the interior, one large launch -, sends the fresh traces of its OUTPUT, and
runs the rest (SG_REGION_SECOND) while they travel; the next stage finds its
halo in place.  (The plainer form - send the input's traces, run the interior,
wait, run the shell - costs a separate small launch for the shell: 1.6 to 2.4
items per wave on an MI355X, 7 % of the step.)
"""
import os
import sys

from . import _lib

# input field of each fused stage (include/seigen_hip.h, enum sg_stage)
STAGE_INPUT = {
    _lib.STAGE_UH1: _lib.FIELD_S,
    _lib.STAGE_STEMP: _lib.FIELD_UH,
    _lib.STAGE_U1: _lib.FIELD_SH,
    _lib.STAGE_SH1: _lib.FIELD_U,
    _lib.STAGE_UTEMP: _lib.FIELD_SH,
    _lib.STAGE_S1: _lib.FIELD_UH,
}
# ... and its output, which is the input of the stage after it (cyclically over the step)
STAGE_OUTPUT = {
    _lib.STAGE_UH1: _lib.FIELD_UH,
    _lib.STAGE_STEMP: _lib.FIELD_SH,
    _lib.STAGE_U1: _lib.FIELD_U,
    _lib.STAGE_SH1: _lib.FIELD_SH,
    _lib.STAGE_UTEMP: _lib.FIELD_UH,
    _lib.STAGE_S1: _lib.FIELD_S,
}
assert all(STAGE_OUTPUT[k] == STAGE_INPUT[(k + 1) % 6] for k in range(6))


def _dist():
    if "torch" not in sys.modules and "WORLD_SIZE" not in os.environ:
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def world():
    """(rank, world_size) of the default process group, (0, 1) if none."""
    dist = _dist()
    if dist is None:
        return 0, 1
    return dist.get_rank(), dist.get_world_size()


class NativeExchanger(object):
    """The exchange inside the library (csrc/comm.cpp): one RCCL communicator per block grid, created from a unique
    id that rank 0 makes and the default process group broadcasts; `step` is ONE call into the C-ABI that runs every
    stage, pack, grouped ncclSend/ncclRecv and SECOND launch of all the steps (the reference's exchange is as
    implicit: elastic.py:364).  Same surface as HaloExchanger as far as ElasticLF4 and bench.py use it."""
    staged = False
    native = True

    def __init__(self, block, partition, group=None):
        """Collective over `group`.  The ranks agree on the outcome of every phase BEFORE the next collective starts, so
        that a failure on one rank (no RCCL, a bad argument, a communicator that did not come up, an exchange that
        delivers the wrong traces) raises on ALL ranks - which then fall back together (ElasticLF4.setup) - instead of
        leaving the others waiting inside a broadcast or ncclCommInitRank."""
        import torch
        import torch.distributed as dist
        from .backend import comm_unique_id
        self.block, self.part = block, partition
        self.sides = [s for s in range(2 * partition.dim) if partition.neighbour(s) is not None]
        rank, nranks = dist.get_rank(group), dist.get_world_size(group)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"

        def agree(failed, what):
            t = torch.tensor([1.0 if failed else 0.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, group=group)
            if t.item() > 0:
                raise RuntimeError("native halo exchange: %s failed on %d rank(s)%s" % (what, int(t.item()), (": %s" % failed) if failed else ""))

        peers = [partition.neighbour(s) for s in range(2 * partition.dim)]
        # phase 1: what can be refused locally (RCCL present, peers against the block's neighbour mask)
        err = None
        try:
            block.comm_check(rank, nranks, peers)
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        agree(err, "the argument check")
        # phase 2: the unique id - made by rank 0, travelling together with its error, if any
        box = [None]
        if rank == 0:
            try:
                box[0] = (comm_unique_id(), None)
            except Exception as e:      # noqa: BLE001
                box[0] = (None, repr(e))
        dist.broadcast_object_list(box, src=0, group=group)
        uid, err = box[0]
        if uid is None:
            raise RuntimeError("native halo exchange: rank 0 could not make an RCCL unique id: %s" % err)
        # phase 3: the communicator (collective inside the library)
        err = None
        try:
            block.comm_init(uid, rank, nranks, peers)
        except Exception as e:      # noqa: BLE001
            err = repr(e)
        try:
            agree(err, "ncclCommInitRank")
            # phase 4: does every side receive the trace of its neighbour's facing side?  (pattern exchange, sg_comm_selftest)
            bad, err = 0, None
            try:
                bad = block.comm_selftest()
            except Exception as e:      # noqa: BLE001
                err = repr(e)
            agree(err or (("%d values arrived from the wrong place" % bad) if bad else None), "the exchange self-test")
        except Exception:
            try:
                block.comm_finalize()
            except Exception:      # noqa: BLE001
                pass
            raise
        self._sent0 = 0
        self.timing = False
        from .backend import comm_library
        self.library, self.version = comm_library()      # which RCCL moves the traces: part of a run's record

    @property
    def bytes_sent(self):
        return self.block.comm_stats()["bytes_sent"]

    def reset_stats(self, timing=False):
        self.timing = bool(timing)
        self.block.comm_stats(reset=True)

    def stats(self):
        st = self.block.comm_stats()
        return {"exposed_wait_ms": st["exposed_wait_ms"], "exposed_wait_transport_ms": st["exposed_wait_ms"],
                "exchanges": st["exchanges"], "bytes_sent": st["bytes_sent"], "host_blocked_ms": 0.0}

    def step(self, nsteps=1):
        if int(nsteps) > 0:
            self.block.step(int(nsteps))

    def step_unpipelined(self, nsteps=1):
        raise RuntimeError("the native exchange runs the pipelined schedule only (SEIGEN_HALO_NATIVE=0 for the "
                           "host-driven exchanger and its plain schedule)")


class HaloExchanger(object):
    """Per-stage trace exchange for one block.

    `block` needs: halo_bytes(field, side), halo_pack(field, side, ptr),
    halo_attach(field, side, ptr), run_stage(stage, region), end_step().
    Buffers are torch tensors on `device` so the same object serves RCCL
    (device tensors) and gloo (CPU tensors).  A process group without
    device-to-device transport (gloo) under a block that lives on a GPU is
    served through pinned host mirrors of the buffers ("host-staged": pack on
    the device, copy out, send, copy in) - the transport for machines without
    a GPU-aware fabric and for exercising the multi-process path on one GPU."""

    def __init__(self, block, partition, device, group=None, stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.block = block
        self.part = partition
        self.group = group
        self.stream = stream      # torch.cuda.Stream the block launches on (None: CPU tensors / gloo)
        # the block's second stream (SECOND launches of a split stage run there beside FIRST, include/seigen_hip.h):
        # only to time how long a stage really waited for traces
        self.stream2 = None
        if stream is not None and hasattr(block, "stream2_ptr") and block.stream2_ptr():
            self.stream2 = torch.cuda.ExternalStream(block.stream2_ptr(), device=stream.device)
        on_gpu = torch.device(device).type == "cuda"
        self.staged = on_gpu and dist.get_backend(group) != "nccl"
        self.hsend, self.hrecv = {}, {}
        self.sides = [s for s in range(2 * partition.dim) if partition.neighbour(s) is not None]
        self.send, self.recv = {}, {}
        # halo buffers hold the block's device type (float in the FP32 mode)
        tdt, esz = (torch.float32, 4) if getattr(block, "dtype", "f64") == "f32" else (torch.float64, 8)
        self.elem_bytes = esz
        for kind, field in (("u", _lib.FIELD_U), ("s", _lib.FIELD_S)):
            for s in self.sides:
                n = block.halo_bytes(field, s) // esz
                self.send[(kind, s)] = torch.zeros(n, dtype=tdt, device=device)
                self.recv[(kind, s)] = torch.zeros(n, dtype=tdt, device=device)
                if self.staged:
                    self.hsend[(kind, s)] = torch.zeros(n, dtype=tdt).pin_memory()
                    self.hrecv[(kind, s)] = torch.zeros(n, dtype=tdt).pin_memory()
        # both fields of a kind read the same ghost buffer: a buffer is consumed by the stage
        # that follows its exchange before the next exchange of that kind starts
        for field in (_lib.FIELD_U, _lib.FIELD_UH, _lib.FIELD_S, _lib.FIELD_SH):
            kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
            for s in self.sides:
                block.halo_attach(field, s, self.recv[(kind, s)].data_ptr())
        if on_gpu:
            # the zero fills above ran on torch's current stream, the block's kernels run on its own
            # (non-blocking) stream: make sure no fill can land after the first pack
            torch.cuda.synchronize(device)
        self.bytes_sent = 0
        self.reset_stats(False)

    # ---- instrumentation: what PyOP2's ParLoopHaloEnd timer measures in the reference (tests/tiling/utils.py:144)
    def reset_stats(self, timing=False):
        """timing=True: measure how long the consumer waits for each exchange - on the launch stream
        (event pair around the wait; with a second stream: from the end of the SECOND launch, which
        runs beside the exchange, to the end of the wait) with a device-aware transport, on the host otherwise."""
        self.timing = bool(timing)
        self.exchanges = 0
        self._wait_host_s = 0.0
        self._host_blocked_s = 0.0
        self._wait_events = []
        self._wait_dev_ms = 0.0

    def _resolve_wait_events(self):
        for a, b in self._wait_events:
            a.synchronize()
            b.synchronize()
            self._wait_dev_ms += max(0.0, a.elapsed_time(b))
        self._wait_events = []

    def stats(self):
        self._resolve_wait_events()
        # exposed_wait_ms keeps the definition of the round-1/2 records: everything the consumer waited for an
        # exchange - on the stream with a device-aware transport, on the host (copy-out synchronisation included)
        # with the host-staged one.  exposed_wait_transport_ms is the part that belongs to the transport alone
        # (host-staged: the send / receive wait without the wait for this rank's own FIRST launch and copies).
        host_staged = self.staged or self.stream is None
        return {"exposed_wait_ms": self._wait_dev_ms + 1e3 * (self._host_blocked_s if host_staged else self._wait_host_s),
                "exposed_wait_transport_ms": self._wait_dev_ms + 1e3 * self._wait_host_s,
                "exchanges": self.exchanges, "bytes_sent": self.bytes_sent, "host_blocked_ms": 1e3 * self._host_blocked_s}

    def start(self, field):
        """Pack the block-side traces of `field` and post the sends / receives."""
        kind = "s" if field in (_lib.FIELD_S, _lib.FIELD_SH) else "u"
        if hasattr(self.block, "halo_pack_sides"):      # all sides in one launch
            self.block.halo_pack_sides(field, {s: self.send[(kind, s)].data_ptr() for s in self.sides})
        else:
            for s in self.sides:
                self.block.halo_pack(field, s, self.send[(kind, s)].data_ptr())
        self.bytes_sent += sum(self.send[(kind, s)].numel() * self.elem_bytes for s in self.sides)
        if not self.sides:
            return (kind, [])
        if self.staged:
            # host-staged: queue the copies out and mark their end; the host waits for that mark and
            # posts the transfers only in finish(), after the caller has queued the next launch
            ev = None
            if self.stream is not None:
                with self.torch.cuda.stream(self.stream):
                    for s in self.sides:
                        self.hsend[(kind, s)].copy_(self.send[(kind, s)], non_blocking=True)
                    ev = self.torch.cuda.Event()
                    ev.record(self.stream)
            else:
                for s in self.sides:
                    self.hsend[(kind, s)].copy_(self.send[(kind, s)])
                self.torch.cuda.synchronize()
            return (kind, ev)
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):   # RCCL orders itself against the CURRENT stream
                return (kind, self.dist.batch_isend_irecv(self._ops(kind, self.send, self.recv)))
        return (kind, self.dist.batch_isend_irecv(self._ops(kind, self.send, self.recv)))

    def _ops(self, kind, wire_out, wire_in):
        """sends in the order of my sides, receives in the order of the FACING sides: transfers between two ranks are
        paired in posting order, and what belongs on my side s is what the neighbour sent from its side s ^ 1.  It only
        matters when two sides lead to one rank (a block that is its own neighbour across an axis): side s then gets
        the facing side's trace - the periodic image - not its own (csrc/comm.cpp exchange does the same)."""
        ops = []
        for s in self.sides:
            ops.append(self.dist.P2POp(self.dist.isend, wire_out[(kind, s)], self.part.neighbour(s), self.group))
        for s in sorted(self.sides, key=lambda side: side ^ 1):
            ops.append(self.dist.P2POp(self.dist.irecv, wire_in[(kind, s)], self.part.neighbour(s), self.group))
        return ops

    def finish(self, pending):
        kind, reqs = pending
        if not self.sides:
            return
        self.exchanges += 1
        if self.timing and (self.staged or self.stream is None):
            import time
            # host-staged transport: only the send/receive wait counts as "waited for traces"; the
            # synchronisation on the copy-out event before it sits behind the FIRST launch, the pack and the
            # device-to-host copies and is this rank's own work (it is reported separately, host_blocked_ms)
            t0 = time.perf_counter()
            if self.staged and reqs is not None:
                reqs.synchronize()
                reqs = None
            t1 = time.perf_counter()
            try:
                return self._finish(kind, reqs)
            finally:
                self._wait_host_s += time.perf_counter() - t1
                self._host_blocked_s += time.perf_counter() - t0
        if self.timing:
            a, b = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            # the wait only costs what it lasts beyond the SECOND launch that was queued just before it
            a.record(self.stream2 if self.stream2 is not None else self.stream)
            self._finish(kind, reqs)
            b.record(self.stream)
            self._wait_events.append((a, b))
            if len(self._wait_events) >= 4096:
                self._resolve_wait_events()
            return
        return self._finish(kind, reqs)

    def _finish(self, kind, reqs):
        if self.staged:
            if reqs is not None:
                reqs.synchronize()          # the copies out are done (later launches keep running)
            for r in self.dist.batch_isend_irecv(self._ops(kind, self.hsend, self.hrecv)):
                r.wait()
            if self.stream is not None:
                with self.torch.cuda.stream(self.stream):
                    for s in self.sides:
                        self.recv[(kind, s)].copy_(self.hrecv[(kind, s)], non_blocking=True)
            else:
                for s in self.sides:
                    self.recv[(kind, s)].copy_(self.hrecv[(kind, s)])
            return
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                for r in reqs:
                    r.wait()
            return
        for r in reqs:
            r.wait()

    def step(self, nsteps=1):
        """`nsteps` LF4 steps.  Per stage: FIRST (needs the halo of the stage's input, which the
        stage before has sent) -> send the output's traces -> SECOND meanwhile -> receive.
        The halo of the very first input is exchanged up front (the caller may have changed the
        fields since the last call)."""
        if int(nsteps) <= 0:
            return
        self.finish(self.start(STAGE_INPUT[0]))
        for _ in range(int(nsteps)):
            for stage in range(6):
                self.block.run_stage(stage, _lib.REGION_FIRST)
                reqs = self.start(STAGE_OUTPUT[stage])
                self.block.run_stage(stage, _lib.REGION_SECOND)
                self.finish(reqs)
            self.block.end_step()

    def step_unpipelined(self, nsteps=1):
        """The plain schedule: per stage exchange the input's traces || interior, then the shell."""
        for _ in range(int(nsteps)):
            for stage in range(6):
                reqs = self.start(STAGE_INPUT[stage])
                self.block.run_stage(stage, _lib.REGION_INTERIOR)
                self.finish(reqs)
                self.block.run_stage(stage, _lib.REGION_BOUNDARY)
            self.block.end_step()
