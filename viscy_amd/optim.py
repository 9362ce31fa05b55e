"""AdamW + warm-up cosine schedule on the flat parameter buffer of a viscy_amd model.

Mirrors ``viscy_utils.optimizers.configure_adamw_scheduler``
(/root/reference/packages/viscy-utils/src/viscy_utils/optimizers.py:10-62): torch AdamW defaults
(betas (0.9, 0.999), eps 1e-8, weight_decay 0.01) over *all* parameters, and MONAI
``WarmupCosineSchedule(warmup_steps, t_total, warmup_multiplier)`` stepped per batch
(λ(step) = m + (1-m)·step/max(1,warmup) during warm-up, then max(0, ½(1+cos(π·progress)))).
The update itself is ONE fused HIP launch over the flat fp32 buffers (csrc/optim.hip).

Everything that changes from step to step — the schedule position, the learning rate, Adam's bias corrections — is
DEVICE state: an int32 step counter and an 8-float block that ``vsx_adamw_advance`` (one thread) recomputes right
before the AdamW launch.  The host keeps a mirror of the counter for logging / checkpoints only.  (Round 1 refreshed
the block from a pinned host buffer with an asynchronous copy; the host runs many hipGraph replays ahead of a 140 ms
step, so a step could apply a later step's learning rate.  No per-step host → device traffic is left.)
"""

from __future__ import annotations

import math

import torch

from . import ops as hip_ops


def warmup_cosine_lambda(step: int, warmup_steps: int, t_total: int, warmup_multiplier: float = 0.0,
                         cycles: float = 0.5) -> float:
    if step < warmup_steps:
        f = float(step) / float(max(1.0, warmup_steps))
        return warmup_multiplier + (1 - warmup_multiplier) * f
    progress = float(step - warmup_steps) / float(max(1, t_total - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(cycles) * 2.0 * progress)))


class FlatAdamW:
    """AdamW over ``engine.flat`` / ``engine.flat_grad`` (see viscy_amd.engine_unext2.Engine)."""

    def __init__(self, engine, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01,
                 schedule: str = "Constant", warmup_steps: int = 0, t_total: int = 0, warmup_multiplier: float = 0.0,
                 ops=None):
        self.engine = engine
        self.ops = ops or hip_ops
        self.base_lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.schedule, self.warmup_steps, self.t_total, self.warmup_multiplier = schedule, warmup_steps, t_total, warmup_multiplier
        dev = engine.flat.device
        # a frozen encoder (VSUNet / FcmaeUNet freeze_encoder=True: engine.py:204-206) sits at the tail of the flat buffer
        # (the order is head, decoder, encoder, stem): the fused launch then covers the trainable prefix only
        self.n_active = engine.trainable_numel() if hasattr(engine, "trainable_numel") else engine.flat.numel()
        self.m = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n_active, dtype=torch.float32, device=dev)
        self.t = 0                 # host mirror of the device step counter (logging, checkpoints)
        self.grad_scale = 1.0
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cfg_dev = torch.zeros(16, dtype=torch.float64, device=dev)  # double: see csrc/optim.hip adamw_advance_kernel
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self._cfg_sent = None

    def _cfg(self):
        b1, b2 = self.betas
        return (float(self.base_lr), float(b1), float(b2), float(self.eps), float(self.wd), float(self.grad_scale),
                1.0 if self.schedule == "WarmupCosine" else 0.0, float(self.warmup_steps), float(self.t_total),
                float(self.warmup_multiplier), 0.5)

    def current_lr(self) -> float:
        if self.schedule == "WarmupCosine":
            return self.base_lr * warmup_cosine_lambda(self.t, self.warmup_steps, self.t_total, self.warmup_multiplier)
        return self.base_lr

    def zero_grad(self) -> None:
        if hasattr(self.ops, "fill_"):
            self.ops.fill_(self.engine.flat_grad, 0.0)  # vsx_fill_f32 (no ATen launch in the captured step)
        else:
            self.engine.flat_grad.zero_()
        if hasattr(self.engine, "_pending_bwd"):
            self.engine._pending_bwd = 0  # a new step: no forward of it is waiting for its backward yet

    def host_prepare(self) -> None:
        """host half of a step: advance the mirror counter; upload the schedule CONSTANTS if one of them was changed since
        the last step (a blocking copy from pageable memory — rare, and never while a capture is open)"""
        cfg = self._cfg()
        if cfg != self._cfg_sent:
            if torch.cuda.is_available() and self.cfg_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FlatAdamW: optimiser constants changed inside a hipGraph capture")
            self.cfg_dev[: len(cfg)].copy_(torch.tensor(cfg, dtype=torch.float64))
            self._cfg_sent = cfg
        self.t += 1

    def device_step(self) -> None:
        """device half (hipGraph-capturable): schedule / bias corrections from the device counter + ONE fused AdamW launch"""
        n = self.n_active
        self.ops.adamw_advance(self.cfg_dev, self.step_dev, self.hyper)
        self.ops.adamw(self.engine.flat[:n], self.engine.flat_grad[:n], self.m, self.v, self.hyper)

    def step(self) -> None:
        self.host_prepare()
        self.device_step()

    def set_step(self, t: int) -> None:
        self.t = int(t)
        self.step_dev.fill_(int(t))

    def state_dict(self):
        return {"m": self.m, "v": self.v, "t": self.t}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.set_step(int(sd["t"]))
