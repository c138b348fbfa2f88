"""Data-parallel gradient exchange of the distillation step.

The reference wraps only the trainable `diffusion` sub-module in torch DDP (lakonlab/parallel/ddp_wrapper.py:9-26)
and lets its bucketed NCCL all-reduce overlap the single backward (lakonlab/models/base_diffusion.py:59-60): ONE
exchange of the trainable set per iteration.  Here the trainable set lives in one flat fp32 buffer that both student
steps accumulate into, and the exchange is the same volume as the reference's: every slice of the buffer is
all-reduced exactly once per iteration, launched asynchronously the moment it is final -- block by block in reverse
while the LAST sample of the LAST student step is still back-propagating -- so only the tail (block 0 + heads) is
exposed.  Slices are whole blocks (tens of MB): few, large messages, which is what a 7-link point-to-point xGMI fabric
wants (per-link bound), not DDP's 25 MB buckets.  torch.distributed is plumbing here (backend "nccl"
== RCCL on ROCm; "gloo" in the CPU tests and in the 2-ranks-on-one-GPU equivalence test, where device buffers are staged
through the host because RCCL refuses two ranks on one device).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def init_distributed(local_rank: int):
    """Join the torchrun rendezvous as one rank and pick this rank's device.  Returns (torch.distributed, device string).

    Production: backend "nccl" (= RCCL over xGMI), rank r on GPU r.  For boxes with ONE GPU (the build's test pool) two environment
    switches let the very same launch lines -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --train`,
    `tools/train.py --launcher pytorch` -- run end to end: ARCFLOW_DIST_BACKEND=gloo (RCCL refuses two ranks on one device; the
    reducer stages device buffers through the host for gloo) and ARCFLOW_DIST_ONE_DEVICE=1 (every rank on cuda:0)."""
    import os
    import torch.distributed as dist
    backend = os.environ.get('ARCFLOW_DIST_BACKEND', 'nccl')
    index = 0 if os.environ.get('ARCFLOW_DIST_ONE_DEVICE', '0') == '1' else local_rank
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set before the process's FIRST torch.cuda call to have any effect: launchers export it, tools/train.py
    # sets it at the top of main(); here it only covers callers that have not touched the device yet)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(index)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', index))
    else:
        dist.init_process_group(backend)
    return dist, f'cuda:{index}'


def host_or_device(dist, device):
    """Where a small control tensor of a collective must live: gloo reduces host tensors, RCCL device tensors."""
    return 'cpu' if dist.get_backend() == 'gloo' else device


class GradReducer:
    def __init__(self, process_group=None):
        import torch.distributed as dist
        self.dist = dist if dist.is_available() and dist.is_initialized() else None
        self.group = process_group
        self.world = self.dist.get_world_size(process_group) if self.dist else 1
        self.rank = self.dist.get_rank(process_group) if self.dist else 0
        self.backend = self.dist.get_backend(process_group) if self.dist else None
        # ARCFLOW_DP_FORCE_COLLECTIVES=1: issue every collective even in a one-rank group -- lets a single-GPU box execute the RCCL branch
        # (async handles, stream waits, the exposed-time events) that otherwise first runs on a multi-GPU node (tests/test_distill.py)
        import os
        self._skip_single = self.world == 1 and os.environ.get('ARCFLOW_DP_FORCE_COLLECTIVES', '0') != '1'
        self._pending: List = []
        self._staged: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.bytes_launched = 0          # per iteration (reset by finish): the tests check "every byte exactly once"
        self.last_exposed_ms = 0.0       # time the compute stream waited in the last finish() (device tensors, RCCL)

    def launch(self, flat_grad: torch.Tensor) -> None:
        """Start the SUM all-reduce of one contiguous slice of the flat gradient buffer (returns immediately)."""
        if flat_grad.numel() == 0:
            return
        self.bytes_launched += flat_grad.numel() * flat_grad.element_size()
        if self.dist is None or self._skip_single:
            return
        assert flat_grad.is_contiguous()
        if flat_grad.is_cuda and self.backend == 'gloo':      # 2 ranks on one GPU (tests): stage through the host
            host = flat_grad.detach().to('cpu', copy=True)
            self._pending.append(self.dist.all_reduce(host, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._staged.append((flat_grad, host))
            return
        self._pending.append(self.dist.all_reduce(flat_grad, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> float:
        """Wait for every outstanding exchange; returns the factor (1/world) that turns the sums into means
        (folded into the optimizer's grad_scale instead of a separate pass over the buffer)."""
        ev = None
        if self._pending and not self._staged and torch.cuda.is_available() and self.backend == 'nccl':
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for h in self._pending:
            h.wait()                      # RCCL: the current stream waits for the collective, the host does not
        if ev is not None:
            ev[1].record()
            self._ev = ev
        for dst, host in self._staged:
            dst.copy_(host)
        self._pending.clear()
        self._staged.clear()
        self.bytes_launched = 0
        return 1.0 / self.world

    def exposed_ms(self) -> float:
        """Milliseconds the compute stream spent blocked on the collectives of the last finish() (syncs the events)."""
        ev = getattr(self, '_ev', None)
        if ev is None:
            return 0.0
        ev[1].synchronize()
        self.last_exposed_ms = ev[0].elapsed_time(ev[1])
        self._ev = None
        return self.last_exposed_ms

    def broadcast_(self, flat: torch.Tensor, src: int = 0) -> torch.Tensor:
        """Overwrite `flat` on every rank with rank `src`'s values -- what DDP does with the module state at construction
        (``_sync_module_states``, reached through lakonlab/parallel/ddp_wrapper.py:19-25): ranks must start from identical trainables,
        identical seeds alone do not guarantee it (a resumed rank, a different library build)."""
        if self.dist is None or self._skip_single:
            return flat
        if flat.is_cuda and self.backend == 'gloo':
            host = flat.detach().to('cpu', copy=True)
            self.dist.broadcast(host, src=src, group=self.group)
            flat.copy_(host)
        else:
            self.dist.broadcast(flat, src=src, group=self.group)
        return flat

    def check_consistent(self, flat: torch.Tensor, what: str = 'trainable parameters') -> None:
        """Raise when the ranks do not hold the same values in `flat` (three fp64 checksums over different index subsets, MIN / MAX
        over ranks; no temporary of the buffer's size: the FLUX trainable set is 650 M values).  Called after the construction
        broadcast and by tools/train.py at every checkpoint interval."""
        if self.dist is None or self._skip_single:
            return
        x = flat.detach().flatten()
        sums = torch.stack([x.sum(dtype=torch.float64), x[::2].sum(dtype=torch.float64), x[1::3].sum(dtype=torch.float64)])
        # NaN != NaN would read as "ranks disagree", and raising on one rank before the collectives would hang the others: the
        # non-finite flag travels with the checksums (4th entry) and every rank raises the same error after the reduce
        bad = (~torch.isfinite(sums)).any().to(torch.float64).reshape(1)
        sums4 = torch.cat([torch.nan_to_num(sums, nan=0.0, posinf=0.0, neginf=0.0), bad])
        dev = 'cpu' if self.backend == 'gloo' else flat.device
        lo, hi = sums4.to(dev).clone(), sums4.to(dev).clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        if float(hi[3]) > 0:
            raise RuntimeError(f'the {what} contain non-finite values on at least one rank (rank {self.rank}: checksums {sums.tolist()})')
        if not torch.equal(lo, hi):
            raise RuntimeError(f'data-parallel ranks disagree on the {what}: checksums min {lo[:3].tolist()} max {hi[:3].tolist()} '
                               f'(rank {self.rank} has {sums.tolist()})')

    def all_reduce_max(self, value: float, device) -> float:
        if self.dist is None or self._skip_single:
            return value
        dev = 'cpu' if self.backend == 'gloo' else device
        t = torch.tensor([value], dtype=torch.float32, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t)
