"""Test helper: the oracle's products by its C / OpenMP restatement when they are large.

`oracle/cmvm.c` holds the same formulas as `oracle/dense_gp.py` (projection outermost, float64, libm exp);
tests/test_oracle_pinned.py checks the two against each other and against the reference-generated golden vectors.  The numpy
forms take 20 - 40 s per product at N ~ 15 000 on one core; these take about a second on the test box's cores."""
import numpy as np

from oracle import cmvm
from oracle import dense_gp as orc

_BIG = 4_000_000          # pairs


def omvm(Z1, Z2, V, scale, noise=0.0):
    """orc.mvm(Z1, Z2, V, scale, noise): scale * K_add(Z1, Z2) V + noise V."""
    if np.shape(Z1)[0] * np.shape(Z2)[0] < _BIG:
        return orc.mvm(Z1, Z2, V, scale, noise)
    return cmvm.mvm(Z1, Z2, V, scale, noise)


def okernel(Z1, Z2, scale=1.0):
    """scale * orc.additive_rbf(Z1, Z2) as a dense float64 matrix."""
    if np.shape(Z1)[0] * np.shape(Z2)[0] < _BIG:
        return scale * orc.additive_rbf(np.asarray(Z1, dtype=np.float64), np.asarray(Z2, dtype=np.float64))
    return cmvm.kernel(Z1, Z2, scale)
