"""Multi-view (frame-sharded) training path: one process per GPU, views sharded over ranks, parameters
replicated, ONE collective per step -- an all-reduce (SUM) of a single flat fp32 gradient bucket over RCCL/xGMI
(SURVEY.md section 8e).  The reference has no distributed code (train.py:138-149 is one view per step on one
GPU); this is the data-parallel layer the north star adds around the operator, nothing more.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ~143 MB bucket all-reduced once per 8-view step costs
O(1 ms) against O(10 ms) of rendering, so no overlap machinery is needed; the bucket is flat so RCCL sees one
large message instead of eight small ones.
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
                return sl
        return None

    def zero(self):
        self.flat.zero_()
        for p, (off, n) in zip(self.params, self.slices):   # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].view_as(p)

    def all_reduce(self, dist, average_over: int = 0):
        """SUM over ranks (one collective).  average_over > 0 divides by that count afterwards (mean-of-views)."""
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if average_over:
            self.flat.div_(float(average_over))
        return self.flat
