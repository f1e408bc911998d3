"""Data-parallel gradient exchange for the flat gradient arena (reference: Lightning DDPStrategy -> torch DDP bucketed
all-reduce over NCCL, train.py:123-127; here RCCL over xGMI via torch.distributed).

One process per GPU.  The gradient arena is cut into the segments the backward pass finishes one after the other
(token path -> event embedder -> prong embedder); each segment's all-reduce(AVG) is issued asynchronously as soon as the
runtime reports it final, so it overlaps with the rest of backward.  BatchNorm buffers are broadcast from rank 0 before
every forward like DDP's broadcast_buffers=True; batch statistics stay per rank (the reference does not sync BN).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


# Validation switch (tests/test_distributed_gpu.py): issue every collective even in a one-rank group, so that the RCCL code path
# -- ReduceOp.AVG on arena slices, issued from the backward's streams, broadcast of the buffer arena -- executes on a one-GPU box.
# A one-rank all-reduce(AVG) / broadcast is the identity; production never sets this.
SINGLE_RANK_COLLECTIVES = False


def _active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES)


def segment_plan(total: int, spans: Dict[str, Tuple[int, int]]) -> Dict[str, List[Tuple[int, int]]]:
    """Map each readiness tag to the [lo, hi) slices of the flat gradient it completes.  Every entry of `spans` is a tag with its
    own slice ("event", "prong", or "prong0" ... "prong4" when the prong embedder's backward is issued block by block);
    'head' owns everything no span covers (token path, position embeddings, ...)."""
    cuts = sorted(spans.values())
    head, pos = [], 0
    for lo, hi in cuts:
        if lo > pos:
            head.append((pos, lo))
        pos = max(pos, hi)
    if pos < total:
        head.append((pos, total))
    plan = {tag: [span] for tag, span in spans.items()}
    plan["head"] = head
    return plan


class GradReducer:
    def __init__(self, flat_grad: torch.Tensor, spans: Dict[str, Tuple[int, int]], group=None):
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.plan = segment_plan(flat_grad.numel(), spans)
        self.pending = []
        self.avg = dist.ReduceOp.AVG if (dist.is_initialized() and dist.get_backend(group) == "nccl") else None

    def on_ready(self, tag: str):
        if self.world == 1 and not SINGLE_RANK_COLLECTIVES:
            return
        for lo, hi in self.plan[tag]:
            seg = self.flat_grad[lo:hi]
            if self.avg is not None:
                self.pending.append((dist.all_reduce(seg, op=self.avg, group=self.group, async_op=True), None))
            else:                                   # gloo (CPU tests): SUM then scale
                self.pending.append((dist.all_reduce(seg, group=self.group, async_op=True), seg))

    def finish(self):
        for work, seg in self.pending:
            work.wait()
            if seg is not None:
                seg.div_(self.world)
        self.pending = []


def broadcast_buffers(flat_buf: torch.Tensor, group=None):
    if _active(group):
        dist.broadcast(flat_buf, 0, group=group)


def sync_state(runtime, optimizer=None, group=None):
    """Rank 0's model state to every rank -- what torch DDP's construction-time ``_sync_module_states`` does for the reference
    (train.py sets no seed and relies on it): the parameter arena, the BatchNorm buffer arena, the num_batches_tracked counters
    and, when an optimizer with flat moments exists already (resume), its moments and step.  The drop-in module hides its
    parameters from torch DDP (`_ddp_params_and_buffers_to_ignore`), so DDP's own broadcast covers the 1-element anchor only."""
    if not _active(group):
        return
    for t in (runtime.flat_param, runtime.flat_buf, getattr(runtime, "flat_nbt", None)):
        if t is not None and t.numel():
            dist.broadcast(t, 0, group=group)
    if optimizer is not None and hasattr(optimizer, "exp_avg"):
        dist.broadcast(optimizer.exp_avg, 0, group=group)
        dist.broadcast(optimizer.exp_avg_sq, 0, group=group)
        step = torch.tensor([int(optimizer._step)], dtype=torch.int64, device=optimizer.exp_avg.device)
        dist.broadcast(step, 0, group=group)
        optimizer._step = int(step.item())
