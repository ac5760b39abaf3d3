"""Projection-matrix generators: counterparts of rp.gen_rp (rp.py:10-32) and rp.space_equally (rp.py:220-268).

RNG call order matches the reference so that, under the same torch / numpy seeds, the same matrices come out
(pinned by tests/golden/gen_rp.npz and space_equally.npz, generated from the reference itself)."""
import math

import numpy as np
import torch

DISTS = ("gaussian", "sphere", "very-sparse", "bernoulli", "uniform")


def gen_rp(d, k, dist="gaussian"):
    """d x k random projection.  gaussian: N(0,1)/sqrt(k); sphere: unit columns * sqrt(d)/sqrt(k);
    very-sparse: {-1,0,+1} with P(+-1) = 1/(2 sqrt(d)); bernoulli: +-1/sqrt(k); uniform: U(-1,1)*sqrt(3)/sqrt(k)."""
    if dist == "gaussian":
        return torch.randn(d, k) / math.sqrt(k)
    if dist == "sphere":
        w = torch.randn(d, k)
        w = w / w.norm(p=2, dim=0, keepdim=True)
        return w * math.sqrt(d) / math.sqrt(k)
    if dist == "very-sparse":
        p = 1.0 / (2.0 * math.sqrt(d))
        cat = torch.distributions.Categorical(torch.tensor([p, 1.0 - 2.0 * p, p]))
        return (cat.sample(torch.Size([d, k])) - 1).to(torch.float)
    if dist == "bernoulli":
        return (torch.bernoulli(torch.rand(d, k)) * 2 - 1) / math.sqrt(k)
    if dist == "uniform":
        return (torch.rand(d, k) * 2 - 1) / math.sqrt(k) * math.sqrt(3)
    raise ValueError("Not a valid RP distribution")


def _cos4_loss(P):
    """sum_{a != b} cos^4(angle(P_a, P_b))  (rp.py:241-254)."""
    norms = P.pow(2).sum(dim=1, keepdim=True).sqrt()
    cos = (P @ P.t()) / (norms @ norms.t())
    cos = cos - torch.eye(P.shape[0], dtype=P.dtype)
    return cos.pow(4).sum()


def space_equally(P, lr, niter):
    """Diversify J projection directions (rows of P, J x d) — DPA-GP.

    d >= J: the input is ignored and J random orthonormal rows are returned (Gram-Schmidt on numpy-RNG Gaussian
    vectors, rp.py:224-239).  Otherwise `niter` plain gradient steps of size `lr` on the cos^4 energy, followed by
    row normalisation (rp.py:256-266).  Returns (P_new, final_loss or None)."""
    n, d = P.shape
    if d >= n:
        rows = []
        for i in range(n):
            v = np.random.randn(d)
            v = v / np.linalg.norm(v)
            for u in rows:                      # modified Gram-Schmidt: v is updated as it goes
                v -= v.dot(u) * u
            if i > 0:
                v = v / np.linalg.norm(v)
            rows.append(v)
        out = torch.from_numpy(np.vstack(rows)).to(P)
        out.requires_grad = False
        return out, None
    if n <= 64 and d <= 64 and P.dtype == torch.float32 and torch.cuda.is_available():
        # one single-workgroup launch instead of `niter` autograd steps on the host (seconds per model build, more than
        # a whole kin8nm-sized fit on the GPU); same recurrence, fp32 summation order differs (~1e-6 on the result)
        from . import ops
        dev = P.device if P.is_cuda else torch.device("cuda", torch.cuda.current_device())
        out, final = ops.space_equally(P.to(dev), lr, niter)
        out = out.to(P.device)
        out.requires_grad = False
        return out, final.to(P.device)
    Q = P.detach().clone().requires_grad_(True)
    for _ in range(niter):
        loss = _cos4_loss(Q)
        (g,) = torch.autograd.grad(loss, Q)
        with torch.no_grad():
            Q -= lr * g
    with torch.no_grad():
        final = _cos4_loss(Q)
        Q /= Q.pow(2).sum(dim=1, keepdim=True).sqrt()
    out = Q.detach()
    return out, final.reshape(1, 1)
