"""Data-parallel gradient exchange of the distillation step.

The reference wraps only the trainable `diffusion` sub-module in torch DDP (lakonlab/parallel/ddp_wrapper.py:9-26)
and lets its bucketed NCCL all-reduce overlap the single backward (lakonlab/models/base_diffusion.py:59-60).  Here the
trainable set lives in flat fp32 buffers, so the exchange is one large all-reduce per student step issued
asynchronously on the process group's stream (RCCL over xGMI on MI355X: few, large messages -- a 7-link
point-to-point fabric is per-link bound) while the next student step computes; the buffers are
averaged and summed when the optimizer needs them.  torch.distributed is plumbing here (backend "nccl" == RCCL on
ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch


class GradReducer:
    def __init__(self, process_group=None):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.group = process_group
        self.world = self.dist.get_world_size(process_group) if self.dist else 1
        self.rank = self.dist.get_rank(process_group) if self.dist else 0
        self._pending: List = []

    def launch(self, flat_grad: torch.Tensor) -> None:
        """Start the SUM all-reduce of one flat gradient buffer (returns immediately)."""
        if self.dist is None or self.world == 1:
            return
        self._pending.append(self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> float:
        """Wait for every outstanding exchange; returns the factor (1/world) that turns the sums into means
        (folded into the optimizer's grad_scale instead of a separate pass over the buffer)."""
        for h in self._pending:
            h.wait()
        self._pending.clear()
        return 1.0 / self.world

    def all_reduce_max(self, value: float, device) -> float:
        if self.dist is None or self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.float32, device=device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t)
