"""Gaussian likelihood, noise prior, constant mean and a light MultivariateNormal (GPyTorch counterparts used at
training_routines.py:345-351, gp_models/models.py:14-20; semantics per SURVEY.md A.3 / B.8)."""
import math

import torch
from torch import nn
from torch.nn import functional as F

from .kernels import inv_softplus
from .operators import LinearOperator

LOG2PI = math.log(2.0 * math.pi)


class SmoothedBoxPrior(nn.Module):
    """gpytorch.priors.SmoothedBoxPrior(a, b, sigma): uniform on [a, b] with Gaussian tails."""

    def __init__(self, a, b, sigma=0.01):
        super().__init__()
        self.a, self.b, self.sigma = float(a), float(b), float(sigma)

    def log_prob(self, x):
        center, radius = 0.5 * (self.a + self.b), 0.5 * (self.b - self.a)
        dist = ((x - center).abs() - radius).clamp_min(0.0)
        log_tail = -0.5 * (dist / self.sigma) ** 2 - math.log(self.sigma) - 0.5 * LOG2PI
        m = 1.0 + (self.b - self.a) / (math.sqrt(2.0 * math.pi) * self.sigma)
        return (log_tail - math.log(m)).sum()


class GaussianLikelihood(nn.Module):
    """noise = softplus(raw_noise) + 1e-4  (GreaterThan(1e-4) constraint); optional prior on the noise."""

    MIN_NOISE = 1e-4

    def __init__(self, noise_prior=None):
        super().__init__()
        self.raw_noise = nn.Parameter(torch.zeros(1))
        self.noise_prior = noise_prior

    @property
    def noise(self):
        return F.softplus(self.raw_noise) + self.MIN_NOISE

    @noise.setter
    def noise(self, value):
        value = torch.as_tensor(value, dtype=torch.float64).reshape(1)
        if float(value) <= self.MIN_NOISE:
            raise ValueError("noise must exceed the lower bound %g" % self.MIN_NOISE)
        self.raw_noise.data = inv_softplus(value - self.MIN_NOISE).to(self.raw_noise)

    def log_prior(self):
        if self.noise_prior is None:
            return torch.zeros((), dtype=self.raw_noise.dtype, device=self.raw_noise.device)
        return self.noise_prior.log_prob(self.noise)

    def forward(self, dist):
        """p(y | f): adds the observation noise to the covariance."""
        noise = self.noise.reshape(())
        if hasattr(dist, "with_observation_noise"):     # models.TrainPosterior: the covariance stays implicit
            return dist.with_observation_noise(noise.detach())
        cov = dist.covariance
        if isinstance(cov, LinearOperator):
            return MultivariateNormal(dist.mean, cov.add_diag(noise))
        if cov.dim() == 1:   # only marginal variances are available
            return MultivariateNormal(dist.mean, cov + noise, diagonal_only=True)
        c = cov.clone()
        c.diagonal().add_(noise)
        return MultivariateNormal(dist.mean, c)

    __call__ = forward


class ConstantMean(nn.Module):
    def __init__(self):
        super().__init__()
        self.constant = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return self.constant.expand(x.shape[0])


class MultivariateNormal:
    """mean (N,), covariance: LinearOperator | dense (N x N) | variances (N,) when diagonal_only."""

    def __init__(self, mean, covariance, diagonal_only=False):
        self.mean = mean
        self.covariance = covariance
        self.diagonal_only = diagonal_only

    @property
    def loc(self):
        return self.mean

    @property
    def lazy_covariance_matrix(self):
        return self.covariance

    @property
    def covariance_matrix(self):
        if isinstance(self.covariance, LinearOperator):
            return self.covariance.to_dense()
        if self.diagonal_only:
            return torch.diag(self.covariance)
        return self.covariance

    @property
    def variance(self):
        if isinstance(self.covariance, LinearOperator):
            return self.covariance._diagonal()
        if self.diagonal_only:
            return self.covariance
        return self.covariance.diagonal()

    @property
    def stddev(self):
        return self.variance.clamp_min(1e-12).sqrt()

    def confidence_region(self):
        """mean -/+ 2 stddev (GPyTorch convention), used at training_routines.py:572-575."""
        s2 = self.stddev * 2.0
        return self.mean - s2, self.mean + s2

    def log_prob(self, value):
        """Dense log-density (posterior predictive; float64 Cholesky for stability)."""
        if isinstance(self.covariance, LinearOperator):
            raise RuntimeError("use ExactMarginalLogLikelihood for operator-backed (train-mode) distributions")
        diff = (value - self.mean).double()
        n = diff.shape[0]
        if self.diagonal_only:
            var = self.covariance.double()
            return (-0.5 * (diff * diff / var).sum() - 0.5 * torch.log(var).sum() - 0.5 * n * LOG2PI).to(value.dtype)
        from .inv_quad_logdet import psd_safe_cholesky
        # (the float64 copy keeps the rounding of the dtype it was computed in: jitter on that scale)
        Lc = psd_safe_cholesky(self.covariance.double(), jitter=1e-6 if self.covariance.dtype == torch.float32 else None)
        z = torch.linalg.solve_triangular(Lc, diff.unsqueeze(-1), upper=False).squeeze(-1)
        return (-0.5 * (z * z).sum() - torch.log(Lc.diagonal()).sum() - 0.5 * n * LOG2PI).to(value.dtype)
