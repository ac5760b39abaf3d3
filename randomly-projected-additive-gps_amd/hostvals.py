"""Host copies of device scalars without a synchronisation each.

The kernels of the C-ABI take scales / noise BY VALUE, so the softplus-transformed hyper-parameters have to reach the host
once per optimiser step.  `float(tensor)` is a device-to-host copy that drains the queue; a training step used to make a
dozen of them (outputscale, noise three times, kernel constants, the preconditioner's log-determinant), each followed by
an idle device while the host caught up (profiles/r4_step_C2_step_gaps.txt: 12 copy kernels per step with 450 us of idle
time in front of them).  `prefetch` moves any number of scalars in ONE copy and remembers the values on the tensor
objects; `host_float` returns the remembered value, or falls back to the plain copy."""
import torch


def prefetch(*tensors):
    """One device-to-host copy for all the given one-element tensors; afterwards `host_float(t)` is free for each."""
    ts = [t for t in tensors if t is not None and isinstance(t, torch.Tensor) and t.numel() == 1]
    todo = [t for t in ts if getattr(t, "_host_value", None) is None]
    if not todo:
        return
    dev = [t for t in todo if t.is_cuda]
    for t in todo:
        if not t.is_cuda:
            t._host_value = float(t.detach())
    if dev:
        vals = torch.stack([t.detach().reshape(()).double() for t in dev]).cpu().tolist()
        for t, v in zip(dev, vals):
            t._host_value = v


def host_float(t):
    """float(t) through the value remembered by `prefetch` (or by an earlier call) when there is one."""
    if not isinstance(t, torch.Tensor):
        return float(t)
    v = getattr(t, "_host_value", None)
    if v is None:
        v = float(t.detach())
        try:
            t._host_value = v
        except Exception:
            pass
    return v
