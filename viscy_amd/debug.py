"""Debug instruments for the kernel schedule (used by tests/ and tools/, never by the product path).

* ``poison_empty()``: every ``torch.empty`` / ``torch.empty_like`` / ``Tensor.new_empty`` floating-point HIP allocation is
  filled with NaN, so a kernel that reads a buffer (or part of one) nobody wrote turns into a deterministic NaN instead of
  "whatever the caching allocator handed back" — the read-before-write detector behind tests/test_gpu_soak.py.
* ``FiniteGuard``: wraps the launch wrappers of ``viscy_amd.ops`` (and ``viscy_amd.losses`` kernels through the same
  module) and checks after every launch that all floating tensor arguments and results are finite; records / raises at the
  FIRST launch that is not — the bisection tool for a non-finite loss (tools/nan_soak.py).
"""

from __future__ import annotations

import contextlib

import torch


@contextlib.contextmanager
def poison_empty(value: float = float("nan")):
    real_empty, real_like = torch.empty, torch.empty_like

    def _poison(t):
        if torch.is_tensor(t) and t.is_cuda and t.is_floating_point() and t.numel():
            t.fill_(value)
        return t

    def empty(*a, **k):
        return _poison(real_empty(*a, **k))

    def empty_like(*a, **k):
        return _poison(real_like(*a, **k))

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real_empty, real_like


class NonFinite(RuntimeError):
    pass


def _tensors(obj, path=""):
    if torch.is_tensor(obj):
        yield path, obj
    elif isinstance(obj, (tuple, list)):
        for i, o in enumerate(obj):
            yield from _tensors(o, f"{path}[{i}]")
    elif isinstance(obj, dict):
        for k, o in obj.items():
            yield from _tensors(o, f"{path}.{k}")


class FiniteGuard:
    """``with FiniteGuard(ops) as g: step()`` — ``g.first`` is None or (call index, op name, tensor path, count of
    non-finite values, shape).  ``raise_on_first=True`` raises NonFinite at that launch."""

    def __init__(self, *modules, raise_on_first: bool = True, skip=("gemm_z",)):
        self.modules, self.raise_on_first, self.skip = modules, raise_on_first, set(skip)
        self._orig = []
        self._pending = []
        self.first = None
        self.calls = 0

    def _check(self, call, name, items):
        if self.first is not None:
            return
        for path, t in items:
            if t.is_cuda and t.is_floating_point() and t.numel():
                bad = (~torch.isfinite(t)).sum().item()
                if bad:
                    self.first = (call, name, path, int(bad), tuple(t.shape))
                    if self.raise_on_first:
                        raise NonFinite(f"first non-finite values after launch #{call} {name}: {path} "
                                        f"{tuple(t.shape)} has {bad} non-finite elements")
                    break

    def _wrap(self, name, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            self.calls += 1
            if self.first is None:
                items = list(_tensors(a, "arg")) + list(_tensors(k, "kw")) + list(_tensors(out, "out"))
                # while ops.batch is collecting, the weight-space ops are queued, not launched: their outputs are written at
                # the flush — everything seen since then is checked there, in call order
                collecting = any(getattr(m, "_BATCH", None) is not None for m in self.modules)
                if collecting and name not in ("flush", "batch_close"):
                    self._pending.append((self.calls, name, items))
                    return out
                pending, self._pending = self._pending, []
                for c, n, it in pending:
                    self._check(c, n, it)
                self._check(self.calls, name, items)
            return out

        return wrapped

    def __enter__(self):
        for mod in self.modules:
            for name in dir(mod):
                fn = getattr(mod, name)
                if (callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == mod.__name__
                        and name not in self.skip and not isinstance(fn, type)):
                    self._orig.append((mod, name, fn))
                    setattr(mod, name, self._wrap(name, fn))
        return self

    def __exit__(self, *exc):
        for mod, name, fn in self._orig:
            setattr(mod, name, fn)
        self._orig.clear()
        if exc[0] is None:
            pending, self._pending = self._pending, []
            for c, n, it in pending:
                self._check(c, n, it)


def soak(step_fn, n_steps: int, state_tensors, check_tensors=None, on_fail=None):
    """Run ``step_fn()`` (returns the loss tensor) ``n_steps`` times, checking the loss and ``check_tensors`` (a callable
    returning {name: tensor}) for non-finite values after EVERY step; a snapshot of ``state_tensors`` (list of tensors:
    parameters / moments) from before the failing step is kept so the step can be replayed.  Returns
    {"steps": run, "first_bad_step": k | None, "what": [...], "snapshot": [...] | None, "losses": [...]}."""
    snap = [torch.empty_like(t) for t in state_tensors]
    losses = []
    for k in range(n_steps):
        for s, t in zip(snap, state_tensors):
            s.copy_(t)
        loss = step_fn()
        lv = float(loss)
        losses.append(lv)
        what = []
        if not torch.isfinite(loss).all().item():
            what.append("loss")
        if check_tensors is not None:
            for name, t in check_tensors().items():
                if not torch.isfinite(t).all().item():
                    what.append(name)
        if what:
            res = {"steps": k + 1, "first_bad_step": k, "what": what, "snapshot": snap, "losses": losses}
            if on_fail is not None:
                on_fail(res)
            return res
    return {"steps": n_steps, "first_bad_step": None, "what": [], "snapshot": None, "losses": losses}
