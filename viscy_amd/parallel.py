"""Pure data parallelism for the flat-buffer engine: one process per GPU, RCCL (torch
``"nccl"`` backend on ROCm) all-reduce of the flat fp32 gradient buffer over xGMI.

Replaces Lightning ``strategy: ddp`` → ``DistributedDataParallel``: the only
collective on the path is the per-step gradient all-reduce.  The engine lays parameters out in
reverse forward order and calls ``on_bucket_ready(i)`` as soon as bucket i's gradients are
complete (0 = head + decoder, 1 = encoder stages 3-2, 2 = stages 1-0 + stem), so each bucket's
all-reduce — ONE large contiguous message, sized for the per-link-bound xGMI ring rather than
DDP's 25 MB default — overlaps the rest of backward.  Averaging (1/world) is folded into the
fused AdamW kernel's grad_scale.  Works with ``gloo`` on CPU for the world_size-2 tests.
"""

from __future__ import annotations

import torch
import torch.distributed as dist


class FlatDataParallel:
    def __init__(self, engine, optimizer=None, process_group=None, broadcast: bool = True, force: bool = False):
        """``force``: run the collective path even in a 1-rank group (exercises RCCL init, the bucket hooks and the
        graph + collective ordering on a single GPU; numerically a no-op)."""
        self.engine = engine
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.works = []
        self._reduced = set()
        self.active = self.world > 1 or (force and dist.is_initialized())
        if self.active:
            if broadcast:  # DDP-style: rank 0's parameters win
                dist.broadcast(engine.flat, src=0, group=process_group)
            engine.on_bucket_ready = self._bucket_ready
            if optimizer is not None:
                optimizer.grad_scale = 1.0 / self.world

    def _bucket_ready(self, i: int) -> None:
        self.reduce_bucket(i)

    def reduce_bucket(self, i: int) -> None:
        """async all-reduce (SUM) of bucket i of the flat gradient buffer; RCCL runs it on its own stream behind everything
        queued on the current stream so far, i.e. it overlaps whatever the caller launches next"""
        if not self.active or i in self._reduced:
            return
        lo, hi = self.engine.bucket_bounds[i]
        self._reduced.add(i)
        self.works.append(dist.all_reduce(self.engine.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def finish(self) -> None:
        """call after backward, before the optimizer step: reduces every bucket the backward hooks did not (a step with
        several forwards fires the hooks only in its last backward; a step whose last backward never ran fires none),
        then waits for all of them"""
        if self.active:
            for i in range(len(self.engine.bucket_bounds)):
                self.reduce_bucket(i)
        for w in self.works:
            w.wait()
        self.works.clear()
        self._reduced.clear()
        self.engine._pending_bwd = 0

    def all_reduce_mean(self, t: torch.Tensor) -> torch.Tensor:
        """``self.log(..., sync_dist=True)`` equivalent for scalars (engine.py:294-303 of the reference)."""
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
            t = t / self.world
        return t


def shard_indices(n: int, rank: int, world: int, seed: int, epoch: int, shuffle: bool = True, drop_last: bool = True):
    """torch ``DistributedSampler`` semantics (what Lightning injects for HCSDataModule.train_dataloader,
    hcs.py:723-735): one global permutation seeded by seed+epoch, rank r takes indices [r::world]."""
    if shuffle:
        g = torch.Generator().manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last:
        total = n // world * world
        idx = idx[:total]
    else:
        total = (n + world - 1) // world * world
        idx += idx[: total - len(idx)]
    return idx[rank:total:world]


def all_gather_with_local_grad(t: torch.Tensor) -> torch.Tensor:
    """[n, D] per rank -> [world * n, D], rank-major, where this rank's own rows keep their autograd history and the other
    ranks' rows are constants (one RCCL all-gather over xGMI, no gradient exchange).  Used for global negatives in the
    DynaCLR NT-Xent loss (BASELINE config 5): every rank evaluates the loss of the GLOBAL batch; its backward yields
    d loss_global / d (local rows), and the data-parallel gradient SUM over ranks is the gradient of the global loss —
    ``scale_for_mean_reduction`` compensates the 1 / world of a mean reduction."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t.detach().contiguous())
    parts[dist.get_rank()] = t
    return torch.cat(parts)


def scale_for_mean_reduction(loss: torch.Tensor) -> torch.Tensor:
    """same value, gradient multiplied by the world size (see ``all_gather_with_local_grad``)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return loss
    w = dist.get_world_size()
    return loss * w - loss.detach() * (w - 1)
