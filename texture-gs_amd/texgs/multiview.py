"""Multi-view (frame-sharded) training path: one process per GPU, views sharded over ranks, parameters
replicated, ONE collective per step -- an all-reduce (SUM) of a single flat fp32 gradient bucket over RCCL/xGMI
(SURVEY.md section 8e).  The reference has no distributed code (train.py:138-149 is one view per step on one
GPU); this is the data-parallel layer the north star adds around the operator, nothing more.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ~143 MB bucket all-reduced once per 8-view step costs
O(1-2 ms) against O(10 ms) of rendering.  The bucket is flat so RCCL sees large messages, and it is reduced as TWO
segments on a side stream (GradBucket.all_reduce_async): the texture gradient (half the bytes) as soon as the last
view's texture-gradient reduce kernel has been issued -- it then overlaps the remaining K8 launches and the host's
end-of-step work --, the per-Gaussian gradients after the last K8; the compute stream waits for both only before the
gradients are read (GradBucket.wait).  Views are dealt to ranks by estimated cost (lpt_shard_views), not round-robin:
the step ends with the slowest rank.
"""
from typing import List, Sequence

import torch


def shard_views(num_views: int, rank: int, world: int) -> List[int]:
    """Round-robin frame sharding: rank r renders views r, r+world, ...  Every view belongs to exactly one rank."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    views = list(range(rank, num_views, world))
    if not views:
        raise ValueError(f"rank {rank} gets no view: {num_views} views over {world} ranks")
    return views


def lpt_shard_views(costs: Sequence[float], rank: int, world: int) -> List[int]:
    """Longest-processing-time-first sharding: views sorted by estimated cost (e.g. the instance count D of the view's last
    render: K6 / K7 time is proportional to it) are dealt, most expensive first, to the rank with the smallest load so far.
    Deterministic (ties by view index), every view belongs to exactly one rank, every rank gets the same NUMBER of views
    when num_views is a multiple of world (the ranks of a step must issue the same number of collectives and the driver's
    bench counts views)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    n = len(costs)
    if n < world:
        raise ValueError(f"{n} views over {world} ranks")
    order = sorted(range(n), key=lambda v: (-float(costs[v]), v))
    quota = [n // world + (1 if r < n % world else 0) for r in range(world)]
    load = [0.0] * world
    mine = [[] for _ in range(world)]
    for v in order:
        r = min((r for r in range(world) if len(mine[r]) < quota[r]), key=lambda r: (load[r], r))
        mine[r].append(v)
        load[r] += float(costs[v])
    return sorted(mine[rank])


class GradBucket:
    """One flat fp32 buffer that backs the .grad of every parameter, so that (a) autograd accumulates the views
    of a step in place, and (b) the whole gradient is all-reduced with a single collective."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = [p for p in params if p is not None and p.requires_grad]
        if not self.params:
            raise ValueError("no parameter requires grad")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.slices = []
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket needs float32 parameters on one device")
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self.slices.append((off, n))
            off += n
        # ViewPipeline state: replica buffers (one per extra stream) that kernels accumulate into instead of `flat`, the
        # replica in use, and the hook the rasterizer's backward calls before its accumulating kernel (K8)
        self.replicas = []
        self.active = 0
        self.before_accumulate = None

    def select(self, k: int):
        """Accumulate the following views into buffer k (0 = `flat`, the one .grad aliases; k > 0 = a private replica,
        zero-filled when first made and after every fold()).  Lets views on different streams accumulate with no ordering
        between them; fold() adds the replicas into `flat` before anything reads .grad."""
        while len(self.replicas) < k:
            self.replicas.append(torch.zeros_like(self.flat))
        self.active = k

    def fold(self):
        self.active = 0
        for r in self.replicas:
            self.flat.add_(r)
            r.zero_()

    def sink_for(self, tensor):
        """The bucket slice backing `tensor`'s gradient if `tensor` is one of the registered leaves AND its .grad still
        aliases that slice, else None (autograd then delivers the gradient the normal way).

        `optimizer.zero_grad(set_to_none=True)` -- the PyTorch default and what the reference's optimize_step does
        (models/texture_gaussian3d.py:442-444) -- sets p.grad = None: the slice is then zeroed and re-attached (the
        semantics of "no gradient yet"), so kernels that accumulate into it are never writing into a buffer the
        optimizer no longer sees.  A .grad replaced by some other tensor is left alone (returns None)."""
        for p, (off, n) in zip(self.params, self.slices):
            if p is tensor:
                sl = self.flat[off:off + n]
                if p.grad is None:
                    sl.zero_()
                    p.grad = sl.view_as(p)
                elif p.grad.data_ptr() != self.flat.data_ptr() + 4 * off or not p.grad.is_contiguous():
                    return None
                return sl if self.active == 0 else self.replicas[self.active - 1][off:off + n]
        return None

    def zero(self):
        self.flat.zero_()
        for p, (off, n) in zip(self.params, self.slices):   # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].view_as(p)

    def segment_of(self, tensors) -> tuple:
        """(offset, count) of the contiguous run of the flat buffer that backs `tensors` (registered parameters that are adjacent
        in the order the bucket was built with) -- e.g. the texture as one segment and everything else as the other."""
        idx = sorted(i for i, p in enumerate(self.params) if any(p is t for t in tensors))
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError("the tensors of a segment must be registered parameters that are adjacent in the bucket")
        off = self.slices[idx[0]][0]
        return off, self.slices[idx[-1]][0] + self.slices[idx[-1]][1] - off

    def all_reduce_async(self, dist, segment=None, after=(), timing=False, wire_dtype=None):
        """Issue the SUM all-reduce of `segment` = (offset, count) of the flat buffer (default: all of it) on the bucket's side
        stream, ordered after the current stream's work so far and after the events in `after`; returns immediately.  Collectives
        are issued in call order on that one stream, so every rank must call this in the same order.  wait() joins.
        Backend "nccl" (= RCCL): in place on the device buffer.  Any other backend (gloo: the CPU tests, and the single-GPU
        rehearsal where two ranks share a device) cannot run asynchronously: the segment is reduced here and now through a host copy.

        `wire_dtype=torch.bfloat16` (opt-in, the wire-size lever of DESIGN.md section 7): the segment travels -- and is SUMMED by the
        collective -- as bf16: half the bytes over xGMI, ~2^-8 relative rounding per partial sum (a segment whose consumer tolerates
        that: the 45 view-dependent SH coefficients per Gaussian are 54 of the 150 MB of a C3 bucket and feed Adam, which normalises
        by the gradient's own running magnitude).  The fp32 buffer is converted into a staging tensor on the comm stream, reduced,
        and converted back in place."""
        off, n = segment if segment is not None else (0, self.flat.numel())
        buf = self.flat[off:off + n]
        if dist.get_backend() != "nccl" or not self.flat.is_cuda:
            for ev in after:                    # (other streams' texture-gradient kernels: the host copy must see their sums)
                ev.synchronize()
            self._reduce_now(dist, buf, wire_dtype)
            return self
        cur = torch.cuda.current_stream(self.flat.device)
        if getattr(self, "_comm", None) is None:
            self._comm = torch.cuda.Stream(self.flat.device)
            self._comm_events = []
        self._comm.wait_stream(cur)
        for ev in after:
            self._comm.wait_event(ev)
        with torch.cuda.stream(self._comm):
            e0 = e1 = None
            if timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self._comm)
            if wire_dtype is None or wire_dtype == torch.float32:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                wire_bytes = n * 4
            else:
                stage = buf.to(wire_dtype)      # (allocated on the comm stream: the caching allocator keeps it there)
                dist.all_reduce(stage, op=dist.ReduceOp.SUM)
                buf.copy_(stage)
                wire_bytes = n * stage.element_size()
            if timing:
                e1.record(self._comm)
                self._comm_events.append((wire_bytes, e0, e1))
        self._pending = True
        return self

    def wait(self):
        """Make the current stream wait for every all-reduce issued with all_reduce_async (before the optimizer / zero())."""
        if getattr(self, "_pending", False):
            torch.cuda.current_stream(self.flat.device).wait_stream(self._comm)
            self._pending = False
        return self.flat

    def comm_timings(self):
        """[(bytes, milliseconds)] of the timed asynchronous all-reduces so far (synchronises their events); clears the list."""
        out = []
        for nbytes, e0, e1 in getattr(self, "_comm_events", []):
            e1.synchronize()
            out.append((nbytes, e0.elapsed_time(e1)))
        if getattr(self, "_comm_events", None):
            self._comm_events.clear()
        return out

    def _reduce_now(self, dist, buf, wire_dtype=None):
        wd = torch.float32 if wire_dtype is None else wire_dtype
        if buf.is_cuda:
            host = torch.empty(buf.shape, dtype=wd).pin_memory()
            host.copy_(buf, non_blocking=True)
            torch.cuda.current_stream(buf.device).synchronize()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host, non_blocking=True)
        elif wd != torch.float32:
            stage = buf.to(wd)
            dist.all_reduce(stage, op=dist.ReduceOp.SUM)
            buf.copy_(stage)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)

    def all_reduce(self, dist, average_over: int = 0):
        """SUM over ranks (one collective).  average_over > 0 divides by that count afterwards (mean-of-views).

        Backend "nccl" (= RCCL over xGMI) reduces the device buffer in place.  Any other backend (gloo: the CPU tests, and
        the single-GPU rehearsal where two ranks share one device -- RCCL refuses that) goes through a pinned host copy."""
        if dist.get_backend() == "nccl":
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        else:
            if self.flat.is_cuda:
                if getattr(self, "_host", None) is None:
                    self._host = torch.empty(self.flat.shape, dtype=torch.float32).pin_memory()
                self._host.copy_(self.flat, non_blocking=True)
                torch.cuda.current_stream(self.flat.device).synchronize()
                dist.all_reduce(self._host, op=dist.ReduceOp.SUM)
                self.flat.copy_(self._host, non_blocking=True)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average_over:
            self.flat.div_(float(average_over))
        return self.flat


class ViewPipeline:
    """The views of one multi-view step on `depth` HIP streams (default 2), round-robin.

    Inside one view every kernel depends on the one before it: K1..K5 are fourteen small launches that never fill the chip,
    K6 and K7 end in a tail, and the host waits once per forward for the instance count.  Views are independent of each other,
    so view i+1 is issued on another stream and fills the holes of view i (C3 on one MI355X: 656 -> 772 views/s with three
    streams; K7 itself leaves no room on a CU it occupies, so the gain is the small kernels and the tails).  Ordering that
    remains:

    * K8 of view i+1 waits for the whole backward of view i (`order="accumulate"`, the default with a GradBucket sink): K8
      adds into the shared gradient sink with plain read-modify-writes; K7 and the texture-gradient reduce only use atomics
      and per-stream scratch and run unordered.  Everything autograd does downstream of the rasterizer node (AccumulateGrad
      of gradients that are not in the sink) is issued after that K8 on the same stream, hence ordered too; gradient paths
      that bypass the rasterizer are the caller's to order (`order="backward"` serialises whole backwards);
    * a stream runs its own views in order, so per-stream scratch (moment accumulators, texture bins: keyed by stream in
      texgs.rasterizer) is never shared by two views in flight;
    * the caller's stream is joined before (parameters, zeroed sinks) and after (all-reduce, optimizer).

    `forward_fn(view)` runs the forward and whatever loss is attached and returns what `backward_fn` needs;
    `backward_fn(obj)` calls autograd (which runs each node on the stream its forward ran on)."""

    def __init__(self, device, depth: int = 2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)] if depth > 1 else []

    def close(self):
        """Free the backward scratch (moment accumulators, texture-gradient bins) cached for this pipeline's streams."""
        from . import rasterizer
        for s in self.streams:
            s.synchronize()
            rasterizer.release_scratch(self.device, s.cuda_stream)
        self.streams = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, views, forward_fn, backward_fn=None, sink=None, order="accumulate", texture_ready=None, prefetch_fn=None,
            prefetch_ahead=1):
        """`prefetch_fn(view)` (optional; e.g. GaussianRasterizer.prefetch with the arguments forward_fn will use): begins the forward
        of the view this stream renders NEXT -- K1 and the instance-count readback -- right after the current view's forward and
        BEFORE its backward is queued, so that the next forward on this stream does not wait for that backward (the first view of every
        stream is begun before the loop).  The host then runs up to one view per stream ahead of the device instead of none, which is
        what keeps the queues full when a host thread is descheduled for a few milliseconds.

        `texture_ready(events)` (optional, needs a GradBucket sink and order="accumulate"): called ONCE, from inside the last
        view's backward, between its K7 + texture-gradient reduce and its K8 -- the point where every view's contribution to
        dL/dtexture has been issued; `events` = one event per stream, recorded after that stream's latest K7 + reduce.  The caller
        starts the texture segment's all-reduce there (GradBucket.all_reduce_async(..., after=events)).

        order (what of view i+1 waits for view i when both add into `sink`):
        "backward"   -- its whole backward (works with any gradient path, e.g. plain autograd accumulation);
        "accumulate" -- only its accumulating kernel K8, through sink.before_accumulate (needs a GradBucket sink);
        "none"       -- nothing: every stream accumulates into its own replica of the bucket, folded at the end."""
        results = []
        views = list(views)
        if not self.streams:                       # depth 1: the caller's stream, nothing to order
            for i, v in enumerate(views):
                obj = forward_fn(v)
                if prefetch_fn is not None and i + 1 < len(views):
                    prefetch_fn(views[i + 1])
                if backward_fn is not None:
                    if texture_ready is not None and sink is not None and i == len(views) - 1:
                        sink.before_accumulate = lambda: texture_ready([])
                    try:
                        backward_fn(obj)
                    finally:
                        if sink is not None:
                            sink.before_accumulate = None
                results.append(obj)
            return results
        if order not in ("backward", "accumulate", "none"):
            raise ValueError(order)
        if sink is None and order != "backward":
            order = "backward"
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)
        prev_bwd = None
        render_done = {}                            # stream index -> event after that stream's latest K7 + texture-gradient reduce
        depth = len(self.streams)
        try:
            ahead = depth * max(1, int(prefetch_ahead))     # (views; prefetch_ahead = forwards begun ahead PER STREAM)
            if prefetch_fn is not None:
                for i, v in enumerate(views[:ahead]):
                    with torch.cuda.stream(self.streams[i % depth]):
                        prefetch_fn(v)
            for i, v in enumerate(views):
                k = i % depth
                s = self.streams[k]
                with torch.cuda.stream(s):
                    if order == "none":
                        sink.select(k)
                    obj = forward_fn(v)
                    if prefetch_fn is not None and i + ahead < len(views):
                        prefetch_fn(views[i + ahead])
                    if backward_fn is not None:
                        if order == "backward" and prev_bwd is not None:
                            s.wait_event(prev_bwd)
                        if order == "accumulate":
                            last = i == len(views) - 1

                            def hook(ev=prev_bwd, s=s, k=k, last=last):
                                if texture_ready is not None:
                                    e = torch.cuda.Event()
                                    e.record(s)
                                    render_done[k] = e
                                    if last:
                                        texture_ready([render_done[j] for j in sorted(render_done)])
                                if ev is not None:
                                    s.wait_event(ev)
                            sink.before_accumulate = hook
                        backward_fn(obj)
                        prev_bwd = torch.cuda.Event()
                        prev_bwd.record(s)
                results.append(obj)
        finally:
            if sink is not None:
                sink.before_accumulate = None
            for s in self.streams:
                cur.wait_stream(s)
            if sink is not None and order == "none":
                sink.fold()
        return results
