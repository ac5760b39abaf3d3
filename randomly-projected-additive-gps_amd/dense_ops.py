"""Dense torch-op operator for kernels that are NOT on the hot path (`kind: full`, BASELINE config 1 = runner plumbing
on CPU).  Everything here is ordinary autograd-tracked torch; it never touches the HIP library."""
import torch

from .operators import LinearOperator


class DenseKernelOperator(LinearOperator):
    def __init__(self, kernel, x1, x2, outputscale=None):
        self.kernel = kernel
        self.x1 = x1
        self.x2 = x2
        self.outputscale = outputscale
        self._dense = None

    def to_dense_autograd(self):
        if self._dense is None:
            K = self.kernel.dense(self.x1, self.x2)
            self._dense = K if self.outputscale is None else K * self.outputscale
        return self._dense

    def to_dense(self):
        return self.to_dense_autograd().detach()

    evaluate = to_dense

    def _size(self):
        return torch.Size((self.x1.shape[0], self.x2.shape[0]))

    @property
    def dtype(self):
        return self.x1.dtype

    @property
    def device(self):
        return self.x1.device

    def _matmul(self, rhs):
        return self.to_dense() @ rhs

    def _transpose_nonbatch(self):
        return DenseKernelOperator(self.kernel, self.x2, self.x1, self.outputscale)

    def _diagonal(self):
        return self.to_dense().diagonal()

    def _get_rows(self, idx):
        return self.to_dense().index_select(0, idx)
