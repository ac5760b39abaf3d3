"""Host copies of device scalars without a synchronisation each.

The kernels of the C-ABI take scales / noise BY VALUE, so the softplus-transformed hyper-parameters have to reach the host
once per optimiser step.  `float(tensor)` is a device-to-host copy that drains the queue; a training step used to make a
dozen of them (outputscale, noise three times, kernel constants, the preconditioner's log-determinant), each followed by
an idle device while the host caught up (profiles/r4_step_C2_step_gaps.txt: 12 copy kernels per step with 450 us of idle
time in front of them).  `prefetch` moves any number of scalars in ONE copy and remembers the values on the tensor
objects; `host_float` returns the remembered value, or falls back to the plain copy.

A remembered value is only as good as the tensor content it was read from: it is stored together with the tensor's
autograd version counter and data pointer, and is dropped when either differs (an in-place update by a hand-written SGD
step, `optimizer.step()` on a parameter passed in directly, a finite-difference perturbation, `set_`/`.data =`
rebinding).  The version counter does not see writes through `.data` views that share the storage — tensors that are
updated that way must not be cached: `remember(t, v)` is for detached temporaries only, and `forget(t)` drops a record."""
import torch

_ATTR = "_host_value"


def _stamp(t):
    return (t._version, t.data_ptr())


def peek(t):
    """The remembered host value of `t` if it is still valid for the tensor's current content, else None."""
    rec = getattr(t, _ATTR, None) if isinstance(t, torch.Tensor) else None
    if rec is None:
        return None
    try:
        if rec[0] == _stamp(t):
            return rec[1]
    except Exception:
        pass
    return None


def remember(t, value):
    """Attach a host value known by other means (e.g. returned by a step kernel through pinned memory) to `t`."""
    try:
        setattr(t, _ATTR, (_stamp(t), float(value)))
    except Exception:
        pass


def forget(t):
    if isinstance(t, torch.Tensor) and hasattr(t, _ATTR):
        try:
            delattr(t, _ATTR)
        except Exception:
            pass


def prefetch(*tensors):
    """One device-to-host copy for all the given one-element tensors; afterwards `host_float(t)` is free for each."""
    ts = [t for t in tensors if t is not None and isinstance(t, torch.Tensor) and t.numel() == 1]
    todo = [t for t in ts if peek(t) is None]
    if not todo:
        return
    dev = [t for t in todo if t.is_cuda]
    for t in todo:
        if not t.is_cuda:
            remember(t, float(t.detach()))
    if dev:
        vals = torch.stack([t.detach().reshape(()).double() for t in dev]).cpu().tolist()
        for t, v in zip(dev, vals):
            remember(t, v)


def host_float(t):
    """float(t) through the value remembered by `prefetch` (or by an earlier call) when it is still current."""
    if not isinstance(t, torch.Tensor):
        return float(t)
    v = peek(t)
    if v is None:
        v = float(t.detach())
        remember(t, v)
    return v
