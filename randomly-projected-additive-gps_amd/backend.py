"""Compute backend seam.  The product ships exactly ONE backend: `HipBackend` (the gfx950 library behind include/rpgp.h).
`set_backend` exists so that host logic (CG, SLQ, sharding, training loop) can be unit-tested on machines without a
GPU by injecting a test double defined under tests/; the package itself never falls back to anything."""
from . import ops


class HipBackend:
    """Thin object wrapper over rpgp_amd.ops (fails loudly without librpgp.so / without a HIP device)."""
    name = "hip-gfx950"

    project = staticmethod(ops.project)
    project_grad = staticmethod(ops.project_grad)
    mvm_sym = staticmethod(ops.mvm_sym)
    supports_pair_shard = True
    supports_padded_dense = True
    prepare = staticmethod(ops.Prepared)
    mvm_sym_prepared = staticmethod(ops.mvm_sym_prepared)
    mvm_rect = staticmethod(ops.mvm_rect)
    dense = staticmethod(ops.dense)
    bilinear_grad = staticmethod(ops.bilinear_grad)
    bilinear_grad_dense = staticmethod(ops.bilinear_grad_dense)
    dense_mvm = staticmethod(ops.dense_mvm)
    supports_symcache = True
    symcache = staticmethod(ops.SymCache)
    symcache_mvm = staticmethod(ops.symcache_mvm)
    pivoted_cholesky = staticmethod(ops.pivoted_cholesky)
    ski_grid = staticmethod(ops.ski_grid)
    ski_mvm = staticmethod(ops.ski_mvm)
    ski_plan = staticmethod(ops.ski_plan)
    ski_diag = staticmethod(ops.ski_diag)
    ski_grid_from_range = staticmethod(ops.ski_grid_from_range)
    ski_scatter = staticmethod(ops.ski_scatter)
    ski_grid_product = staticmethod(ops.ski_grid_product)
    ski_gather = staticmethod(ops.ski_gather)
    ski_dense = staticmethod(ops.ski_dense)
    ski_pivoted_cholesky = staticmethod(ops.ski_pivoted_cholesky)
    ski_bilinear_grad = staticmethod(ops.ski_bilinear_grad)
    ski_bilinear_grad_comp = staticmethod(ops.ski_bilinear_grad_comp)
    ski_bilinear_scatter = staticmethod(ops.ski_bilinear_scatter)
    ski_bilinear_finish = staticmethod(ops.ski_bilinear_finish)
    make_family = staticmethod(ops.Family)
    family_mvm_sym = staticmethod(ops.family_mvm_sym)
    family_mvm_rect = staticmethod(ops.family_mvm_rect)
    family_dense = staticmethod(ops.family_dense)
    family_bilinear_grad = staticmethod(ops.family_bilinear_grad)
    family_bilinear_grad_dense = staticmethod(ops.family_bilinear_grad_dense)
    family_pivoted_cholesky = staticmethod(ops.family_pivoted_cholesky)
    gram_f64 = staticmethod(ops.gram_f64)
    woodbury_apply = staticmethod(ops.woodbury_apply)
    woodbury_setup = staticmethod(ops.woodbury_setup)
    woodbury_solve = staticmethod(ops.woodbury_solve)
    make_operator_desc = staticmethod(ops.make_operator_desc)
    make_sum_operator_desc = staticmethod(ops.make_sum_operator_desc)
    mbcg_solve = staticmethod(ops.mbcg_solve)
    slq_logdet_history = staticmethod(ops.slq_logdet_history)
    step_hyper = staticmethod(ops.step_hyper)
    step_probes = staticmethod(ops.step_probes)
    step_value = staticmethod(ops.step_value)
    step_value_wait = staticmethod(ops.step_value_wait)
    step_lr = staticmethod(ops.step_lr)
    step_hyper_backward = staticmethod(ops.step_hyper_backward)


_backend = HipBackend()


def get_backend():
    return _backend


def set_backend(b):
    """Install another backend object (tests only). Returns the previous one."""
    global _backend
    prev = _backend
    _backend = b
    return prev
