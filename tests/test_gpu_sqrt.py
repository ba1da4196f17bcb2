"""Square-root (QR) primitives on the GPU (include/pnmol_sqrt.h) against LAPACK and against the reference's own
identities for base/sqrt.py (tests/test_base/test_sqrt.py:36-109)."""

import numpy as np
import pytest
import scipy.linalg

import pnmol
from pnmol import _hip

pytestmark = pytest.mark.gpu


def _canonical_r(A):
    R = scipy.linalg.qr(A, mode="r")[0][: A.shape[1]]
    if R.shape[0] < A.shape[1]:
        R = np.vstack([R, np.zeros((A.shape[1] - R.shape[0], A.shape[1]))])
    sgn = np.where(np.diag(R) < 0, -1.0, 1.0)
    return sgn[:, None] * R


@pytest.mark.parametrize("rows,cols", [(32, 32), (64, 32), (7, 5), (300, 40), (257, 256), (1000, 130), (3072, 96),
                                       (20, 33), (2050, 700)])
def test_qr_r_matches_lapack(hip_ctx, rows, cols):
    rng = np.random.default_rng(rows * 1000 + cols)
    A = rng.standard_normal((rows, cols))
    R = hip_ctx.qr_r(A)
    assert np.all(np.tril(R, -1) == 0) and np.all(np.diag(R) >= 0)
    np.testing.assert_allclose(R.T @ R, A.T @ A, rtol=1e-12, atol=1e-12 * np.abs(A.T @ A).max())
    if rows >= cols:
        Rl = _canonical_r(A)
        np.testing.assert_allclose(R, Rl, rtol=1e-9, atol=1e-11 * np.abs(Rl).max())


def test_qr_r_structured_input(hip_ctx):
    """Zero blocks, a triangular block, exactly dependent columns: what the filter's stacked matrices look like."""
    rng = np.random.default_rng(3)
    C = np.tril(rng.standard_normal((96, 96)))
    A = np.vstack([C, np.zeros((40, 96)), rng.standard_normal((64, 96))])
    A[:, 50] = 0.0
    A[:, 60] = A[:, 10]
    R = hip_ctx.qr_r(A)
    np.testing.assert_allclose(R.T @ R, A.T @ A, rtol=1e-12, atol=1e-12 * np.abs(A.T @ A).max())
