"""One training step of the hot path as a replayable HIP graph.

Eager PyTorch issues ~700 kernel launches per UNeXt2 step through Python + ctypes; at the metric's
small patch size (256x256) the launch gaps are a double-digit share of the step.  Every kernel of
the path is launched on the caller's stream with caller-owned buffers (include/vsx.h), so the whole
step — zero-grad, forward, MixedLoss, backward, AdamW — can be captured once into a hipGraph
(``torch.cuda.CUDAGraph`` is the capture/replay plumbing) and replayed with ONE launch per step.
Shapes must stay fixed between replays (the data loader's batch is copied into the static input
buffers); what changes per step (lr schedule position, Adam bias corrections) is device state
advanced by a kernel inside the step (viscy_amd/optim.py) — a replay reads no host memory.

With world_size > 1 the step is captured in segments that end where a gradient bucket completes; the
bucket's RCCL all-reduce is issued between two replays and overlaps the next segment (collectives
stay outside the captures).
"""

from __future__ import annotations

import contextlib
import gc
import os

import torch



@contextlib.contextmanager
def _no_gc():
    """No cyclic garbage collection while a stream is capturing: a collection that runs INSIDE a capture can finalise objects whose
    destructors call APIs a capturing stream forbids (a pinned host buffer of an earlier DataLoader batch -> hipHostFree), which
    aborts the process — seen once in ~3 full `-m gpu` runs, always in the first capture behind a DataLoader test (round 4).
    torch.cuda.graph collects BEFORE it starts capturing; this keeps the collector off until the capture has ended."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()

# Captures run in THREAD-LOCAL error mode.  In the default ("global") mode any thread's call that a capturing stream forbids fails
# the capture — and a process with an RCCL process group has such a thread: the ProcessGroupNCCL watchdog polls the events of the
# collectives still on its list (hipEventQuery), so a capture that begins while the work of an earlier all-reduce (the first
# all-reduce, the parameter checksum of bench.py) has not been retired yet ended the process with "operation not permitted when
# stream is capturing" raised FROM THE WATCHDOG — one `bench.py --force-dp` run in three at round 6's end.  A DataLoader's
# pin-memory thread is the same hazard for Trainer.fit.  The capturing thread itself is checked as before.
_CAPTURE_MODE = os.environ.get("VSX_CAPTURE_MODE", "thread_local")  # ("global" reproduces the failure: tests/test_gpu_soak.py)


class TrainStep:
    """``step(x, t) -> loss`` = zero-grad, forward, loss, backward, gradient all-reduce (if data parallel), AdamW.

    Two drivers:
      * *direct* (``criterion`` given, the UNeXt2 / FCMAE supervised step): the engine's forward and its staged backward
        (``Engine.backward_stages``) are called straight from here, the loss gradient comes from the criterion's own
        backward.  The backward pauses after each gradient bucket, so with data parallelism the step is captured as
        THREE hipGraph segments (… bucket 0 | bucket 1 | bucket 2) plus one for AdamW, and each bucket's RCCL all-reduce is
        issued between two replays: it runs on RCCL's stream underneath the next segment, exactly like the eager hooks
        (reference: DDP's bucketed overlap, ``recipes/topology/ddp_4gpu.yml``).  Collectives themselves stay outside the
        captures.
      * *autograd* (``loss_fn(x, t) -> loss``: DynaCLR's two forwards + NT-Xent, FCMAE masked pre-training): one
        ``loss.backward()``, captured whole; with data parallelism the flat gradient buffer is reduced behind the graph.

    Capturing runs two warm-up steps (allocator / workspace sizing); parameters, Adam moments, the step counter and the
    module buffers are restored afterwards, so N calls are exactly N optimisation steps in every mode."""

    def __init__(self, model, criterion, optimizer, ddp=None, use_graph: bool = True, loss_fn=None, segments: bool | None = None,
                 static_inputs: bool = False):
        """``static_inputs``: the caller hands the SAME two device tensors to every call and refills them in place (a loader
        that writes its host->device copies straight into them; the bench, whose batch is resident): the captured graph reads
        them directly.  Otherwise the step owns static copies and every call pays a device-to-device copy of the batch into them
        (2 GB read + 2 GB written at B = 512: 0.6 - 0.8 ms of a 105 ms step)."""
        self.model, self.crit, self.opt, self.ddp = model, criterion, optimizer, ddp
        self.static_inputs = static_inputs
        self.loss_fn = loss_fn
        self.use_graph = use_graph
        self.graphs = None
        self.gopt = None
        self.x = self.t = self.loss = None
        self.world = ddp.world if ddp is not None else 1
        self.dist = bool(ddp is not None and getattr(ddp, "active", self.world > 1))  # collectives outside the graph
        from .unext2 import UNeXt2

        # the direct driver needs the plain `model(x) -> prediction` form with the engine right behind it
        self.direct = loss_fn is None and isinstance(model, UNeXt2)
        if loss_fn is None and not self.direct:
            self.loss_fn = lambda x, t: criterion(model(x), t)
        self.segments = (self.dist and self.direct) if segments is None else (bool(segments) and self.direct)

    # ---- direct driver
    def _loss_and_grad(self, pred):
        from .losses import MixedLoss, _MixedLossFn

        if isinstance(self.crit, MixedLoss):  # the criterion's kernels, without an autograd graph around them
            class _Ctx:
                pass

            ctx = _Ctx()
            loss = _MixedLossFn.forward(ctx, pred, self.t, float(self.crit.l1_alpha), float(self.crit.l2_alpha),
                                        float(self.crit.ms_dssim_alpha))
            if getattr(self, "_one", None) is None or self._one.device != pred.device:
                self._one = torch.ones((), dtype=torch.float32, device=pred.device)  # created once, outside any capture
            return loss, _MixedLossFn.backward(ctx, self._one)[0]
        p = pred.detach().requires_grad_(True)
        with torch.enable_grad():
            loss = self.crit(p, self.t)
        (dout,) = torch.autograd.grad(loss, p)
        return loss.detach(), dout

    def _direct_stages(self):
        """generator: runs the step up to the end of gradient bucket i, yields i.  The compute dtype is resolved BEFORE autocast
        is switched off (an ambient ``torch.autocast(bfloat16)`` — Lightning's bf16-mixed — selects the bf16 kernels, as in
        ``unext2_apply``), and the thread-local no-grad / no-autocast modes are entered per stretch, never held across a
        ``yield``: collectives, graph-capture exits and user hooks between two segments see the caller's modes."""
        eng = self.model.engine()
        dt = self.model._resolve_dtype()
        self.opt.zero_grad()

        def quiet():
            import contextlib

            st = contextlib.ExitStack()
            st.enter_context(torch.no_grad())
            st.enter_context(torch.autocast("cuda", enabled=False))
            return st

        with quiet():
            out, sv = eng.forward(self.x.float(), dt, True)
            eng._pending_bwd = 0  # this driver runs the backward itself
            loss, dout = self._loss_and_grad(out)
            self.loss = loss.detach()
            del out
            it = eng.backward_stages(sv, dout)
        while True:
            with quiet():
                try:
                    i = next(it)
                except StopIteration:
                    return
            yield i

    # ---- autograd driver
    def _fwd_bwd(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.x, self.t)
        loss.backward()
        return loss.detach()

    def _eager(self, collectives: bool = True):
        if self.direct:
            for i in self._direct_stages():
                if collectives and self.ddp is not None:
                    self.ddp.reduce_bucket(i)
        else:
            self.loss = self._fwd_bwd()
        if collectives and self.ddp is not None:
            self.ddp.finish()

    def _state(self):
        eng = self.model.engine()
        return [eng.flat, self.opt.m, self.opt.v, self.opt.step_dev] + [b for b in self.model.buffers()]

    def _capture(self):
        eng = self.model.engine()
        eng.attach_grads()
        hook = eng.on_bucket_ready
        eng.on_bucket_ready = None  # collectives are issued outside the captures
        state = self._state()
        saved = [t.clone() for t in state]
        t_host = self.opt.t
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up on a side stream (allocator / workspace sizing); its updates are undone below
            for _ in range(2):
                self.opt.host_prepare()
                self._eager(collectives=False)
                self.opt.device_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.opt.host_prepare()
        graphs, pool = [], None

        def capture(body):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            with _no_gc(), torch.cuda.graph(g, pool=pool, capture_error_mode=_CAPTURE_MODE):
                body()
            pool = g.pool()
            graphs.append(g)

        if self.direct and self.segments:
            it = self._direct_stages()
            for _ in range(len(eng.bucket_bounds)):
                capture(lambda: next(it))
            for _ in it:  # (exhausts the generator: nothing is launched after the last bucket)
                raise RuntimeError("backward_stages launched work after the last gradient bucket")
        elif self.direct:
            capture(lambda: [None for _ in self._direct_stages()])
        else:
            def body():
                self.loss = self._fwd_bwd()

            capture(body)
        gopt = torch.cuda.CUDAGraph()  # its own graph: with data parallelism it runs after the last bucket's all-reduce
        with _no_gc(), torch.cuda.graph(gopt, pool=pool, capture_error_mode=_CAPTURE_MODE):
            self.opt.device_step()
        self.gopt = gopt
        # the warm-up steps (and nothing else: a capture records, it does not execute) changed the training state
        with torch.no_grad():
            for t, c in zip(state, saved):
                t.copy_(c)
        self.opt.t = t_host
        self.graphs = graphs
        eng.on_bucket_ready = hook

    def __call__(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        if not self.use_graph:
            self.x, self.t = x, t
            self.opt.host_prepare()
            self._eager()
            self.opt.device_step()
            return self.loss
        if self.graphs is None:
            if self.static_inputs and x.is_contiguous() and t.is_contiguous():
                self.x, self.t = x, t  # the graph is captured on the caller's own buffers
            else:
                self.x, self.t = x.clone(), t.clone()
            self._capture()
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x, non_blocking=True)
            self.t.copy_(t, non_blocking=True)
        self.opt.host_prepare()
        if self.dist and self.segments:
            for i, g in enumerate(self.graphs):
                g.replay()
                self.ddp.reduce_bucket(i)  # async on RCCL's stream: overlaps the next segment
            self.ddp.finish()
        else:
            for g in self.graphs:
                g.replay()
            if self.dist:
                self.ddp.finish()  # reduces all buckets behind the graph
        self.gopt.replay()
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
            with _no_gc(), torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                self.y = self.model(self.x)
            self.graph = g
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.y
