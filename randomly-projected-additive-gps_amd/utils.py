"""Model-introspection helpers with the names of the reference's utils.py (:7-52), on this build's kernel modules.

An EXTRA outside the hot-path scope (SURVEY.md §2.1 #12 marks the reference file out of scope): kept because users of the
reference's notebooks call these four getters on a fitted model, and they are 40 lines on this package's own kernels."""
import numpy as np
import torch

from . import kernels as _k


def get_lengthscales(kernel):
    """Lengthscales of a (possibly ScaleKernel-wrapped) kernel: a tensor for the ARD kernels, a list of per-component
    lists for the generalised projection family, else None."""
    if isinstance(kernel, _k.ScaleKernel):
        return get_lengthscales(kernel.base_kernel)
    if isinstance(kernel, _k.GeneralizedProjectionKernel):
        flat = kernel.lengthscales.detach().reshape(-1).tolist()
        out, pos = [], 0
        for deg in kernel.component_degrees:
            out.append(flat[pos:pos + deg])
            pos += deg
        return out
    ls = getattr(kernel, "lengthscale", None)
    return ls if isinstance(ls, torch.Tensor) else None


def get_mixins(kernel):
    """Per-component output scales of the generalised projection family, else None."""
    if isinstance(kernel, _k.GeneralizedProjectionKernel):
        return kernel.outputscales.detach().reshape(-1).tolist()
    if isinstance(kernel, _k.ScaleKernel):
        return get_mixins(kernel.base_kernel)
    return None


def get_outputscale(kernel):
    return kernel.outputscale if isinstance(kernel, _k.ScaleKernel) else None


def format_for_str(num_or_list, decimals=3):
    if isinstance(num_or_list, torch.Tensor):
        return format_for_str(num_or_list.tolist(), decimals)
    if isinstance(num_or_list, list):
        return [format_for_str(n, decimals) for n in num_or_list]
    if isinstance(num_or_list, float):
        return np.round(num_or_list, decimals)
    return ""
