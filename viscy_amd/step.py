"""One training step of the hot path as a replayable HIP graph.

Eager PyTorch issues ~700 kernel launches per UNeXt2 step through Python + ctypes; at the metric's
small patch size (256x256) the launch gaps are a double-digit share of the step.  Every kernel of
the path is launched on the caller's stream with caller-owned buffers (include/vsx.h), so the whole
step — zero-grad, forward, MixedLoss, backward, AdamW — can be captured once into a hipGraph
(``torch.cuda.CUDAGraph`` is the capture/replay plumbing) and replayed with ONE launch per step.
Shapes must stay fixed between replays (the data loader's batch is copied into the static input
buffers); hyper-parameters that change per step (lr schedule, Adam bias corrections) live in a
pinned host buffer that is refreshed before every replay.

With world_size > 1 the graph holds forward + backward; the RCCL all-reduce of the flat gradient
buffer and the fused AdamW launch run eagerly behind it (collectives stay outside the capture).
"""

from __future__ import annotations

import torch


class TrainStep:
    def __init__(self, model, criterion, optimizer, ddp=None, use_graph: bool = True, loss_fn=None):
        """``loss_fn(x, t) -> scalar loss`` replaces ``criterion(model(x), t)`` for steps of another shape (DynaCLR: two
        forwards + NT-Xent on (anchor, positive); FCMAE pre-training: masked forward + MaskedMSELoss with ``t`` unused).
        Everything it launches must be capturable: device-side randomness only, no host synchronisation."""
        self.model, self.crit, self.opt, self.ddp = model, criterion, optimizer, ddp
        self.loss_fn = loss_fn
        self.use_graph = use_graph
        self.graph = None
        self.x = self.t = self.loss = None
        self.world = ddp.world if ddp is not None else 1
        self.dist = bool(ddp is not None and getattr(ddp, "active", self.world > 1))  # collectives outside the graph

    # ---- the captured body
    def _fwd_bwd(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.x, self.t) if self.loss_fn is not None else self.crit(self.model(self.x), self.t)
        loss.backward()
        return loss.detach()

    def _capture(self):
        eng = self.model.engine()
        hook = eng.on_bucket_ready
        eng.on_bucket_ready = None  # collectives are issued outside the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up on a side stream (allocator / workspace sizing)
            for _ in range(2):
                self.opt.host_prepare()
                self._fwd_bwd()
                if not self.dist:
                    self.opt.device_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self.opt.host_prepare()
        with torch.cuda.graph(g):
            self.loss = self._fwd_bwd()
            if not self.dist:
                self.opt.device_step()
        self.opt.t -= 1  # the capture pass records but does not execute: it is not an optimisation step
        self.graph = g
        eng.on_bucket_ready = hook

    def __call__(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        if not self.use_graph:
            self.x, self.t = x, t
            loss = self._fwd_bwd()
            if self.ddp is not None:
                self.ddp.finish()
            self.opt.step()
            return loss
        if self.graph is None:
            self.x, self.t = x.clone(), t.clone()
            self._capture()
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
            self.t.copy_(t, non_blocking=True)
        self.opt.host_prepare()
        self.graph.replay()
        if self.dist:
            import torch.distributed as dist

            dist.all_reduce(self.model.engine().flat_grad, op=dist.ReduceOp.SUM, group=self.ddp.pg)
            self.opt.device_step()
        return self.loss


class InferStep:
    """Forward-only counterpart of ``TrainStep`` for tiled / sliding-window inference (BASELINE config 4): every window
    of a FOV has the same shape, so the ~300 launches of one forward are captured once into a hipGraph and replayed per
    window; the window is copied into a static input buffer, the output buffer is reused (clone it to keep it)."""

    def __init__(self, model, use_graph: bool = True):
        self.model, self.use_graph = model, use_graph
        self.graph = None
        self.x = self.y = None

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not self.use_graph:
            return self.model(x)
        if self.graph is None or self.x.shape != x.shape or self.x.dtype != x.dtype:
            self.x = x.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):  # warm-up on a side stream (allocator, arena sizing, weight preparation)
                for _ in range(2):
                    self.model(self.x)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.y = self.model(self.x)
            self.graph = g
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.y
