"""Square-root transition utilities (reference: src/pnmol/base/sqrt.py), computed on the GPU.

Each function is the reference's function of the same name; the QR behind it runs on the device
(include/pnmol_sqrt.h).  Factors are returned with a non-negative diagonal (the reference's carry LAPACK's
data-dependent signs); every product formed from them is the same.  No CPU fallback.
"""

import numpy as np

from pnmol import _hip


def _ctx(ctx):
    return ctx if ctx is not None else _hip.Context.default()


def propagate_cholesky_factor(S1, S2, *, ctx=None):
    """Cholesky factor of S1 S1^T + S2 S2^T (sqrt.py:8-12)."""
    return _ctx(ctx).sqrt_propagate(S1, S2)


def sqrtm_to_cholesky(St, *, ctx=None):
    """St = S^T a 'right' square root, M = S S^T: the lower-triangular factor of M (sqrt.py:15-23)."""
    return _ctx(ctx).sqrt_propagate(np.asarray(St).T, None)


def update_sqrt(transition_matrix, cov_cholesky, meascov_sqrtm, *, ctx=None):
    """(posterior factor, Kalman gain, innovation factor), sqrt.py:33-73."""
    return _ctx(ctx).sqrt_update(transition_matrix, cov_cholesky, meascov_sqrtm)


def update_sqrt_no_meascov(transition_matrix, cov_cholesky, *, ctx=None):
    """sqrt.py:76-95."""
    return _ctx(ctx).sqrt_update(transition_matrix, cov_cholesky, None)


def batched_propagate_cholesky_factor(S1s, S2s, *, ctx=None):
    """sqrt.py:27-29 (jax.vmap in the reference)."""
    return np.stack([propagate_cholesky_factor(a, b, ctx=ctx) for a, b in zip(S1s, S2s)])


def batched_sqrtm_to_cholesky(Sts, *, ctx=None):
    """sqrt.py:30."""
    return np.stack([sqrtm_to_cholesky(a, ctx=ctx) for a in Sts])


def batched_update_sqrt(batched_transition_matrix, batched_cov_cholesky, *, ctx=None):
    """sqrt.py:103-111; as there, calls the noise-free update per item (the reference's loop passes two arguments)."""
    out = [update_sqrt_no_meascov(A, SC, ctx=ctx) for A, SC in zip(batched_transition_matrix, batched_cov_cholesky)]
    return tuple(np.stack(x) for x in zip(*out))
