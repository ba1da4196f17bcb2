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


# ---- the reference's own tests of base/sqrt.py (tests/test_base/test_sqrt.py:9-109), run against the device functions ----
import pnmol_oracle as o  # noqa: E402
from pnmol.base import sqrt as dsqrt  # noqa: E402


@pytest.fixture
def iwp():
    return pnmol.base.iwp.IntegratedWienerTransition(wiener_process_dimension=1, num_derivatives=1,
                                                     wp_diffusion_sqrtm=np.eye(1))


@pytest.fixture(params=["full", "partial"])
def H_and_SQ(iwp, request):
    H, SQ = iwp.preconditioned_discretize_1d
    return (H, SQ) if request.param == "full" else (H[:1], SQ[:1, :1])


@pytest.fixture
def SC(iwp):
    return iwp.preconditioned_discretize_1d[1]


def test_propagate_cholesky_factor(hip_ctx, H_and_SQ, SC):
    H, SQ = H_and_SQ
    chol = dsqrt.propagate_cholesky_factor(H @ SC, SQ, ctx=hip_ctx)
    cov = H @ SC @ SC.T @ H.T + SQ @ SQ.T
    assert np.allclose(chol @ chol.T, cov)
    assert np.allclose(np.tril(chol), chol)


@pytest.mark.parametrize("noise", [True, False])
def test_update_sqrt(hip_ctx, H_and_SQ, SC, noise):
    H, SQ = H_and_SQ
    if noise:
        SC_new, gain, innov_chol = dsqrt.update_sqrt(H, SC, SQ, ctx=hip_ctx)
        S = H @ SC @ SC.T @ H.T + SQ @ SQ.T
    else:
        SC_new, gain, innov_chol = dsqrt.update_sqrt_no_meascov(H, SC, ctx=hip_ctx)
        S = H @ SC @ SC.T @ H.T
    assert SC_new.shape == SC.shape and gain.shape == (H.shape[1], H.shape[0])
    assert innov_chol.shape == (H.shape[0], H.shape[0])
    K = SC @ SC.T @ H.T @ np.linalg.inv(S)
    C = SC @ SC.T - K @ S @ K.T
    assert np.allclose(SC_new @ SC_new.T, C) and np.allclose(SC_new, np.tril(SC_new))
    assert np.allclose(K, gain)
    assert np.allclose(innov_chol @ innov_chol.T, S) and np.allclose(innov_chol, np.tril(innov_chol))


# ---- at filter sizes, against the oracle's restatement (LAPACK) ------------------------------------------------------
def _canon_lower(L):
    return L * np.where(np.diag(L) < 0, -1.0, 1.0)[None, :]


@pytest.mark.parametrize("n,k1,k2", [(96, 96, 96), (130, 130, 70), (300, 300, 300)])
def test_propagate_matches_oracle(hip_ctx, n, k1, k2):
    rng = np.random.default_rng(n)
    S1, S2 = rng.standard_normal((n, k1)), rng.standard_normal((n, k2))
    got, want = dsqrt.propagate_cholesky_factor(S1, S2, ctx=hip_ctx), _canon_lower(o.propagate_cholesky_factor(S1, S2))
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11 * np.abs(want).max())
    got1 = dsqrt.sqrtm_to_cholesky(S1.T, ctx=hip_ctx)
    np.testing.assert_allclose(got1 @ got1.T, S1 @ S1.T, rtol=1e-11, atol=1e-11 * np.abs(S1 @ S1.T).max())


@pytest.mark.parametrize("m,D,noise", [(34, 96, True), (34, 96, False), (130, 384, True), (258, 768, True), (100, 100, True)])
def test_update_matches_oracle(hip_ctx, m, D, noise):
    rng = np.random.default_rng(m * D)
    H = rng.standard_normal((m, D)) / np.sqrt(D)
    C = np.tril(rng.standard_normal((D, D))) / np.sqrt(D)
    E = np.diag(rng.uniform(0.1, 1.0, m)) if noise else None
    got = dsqrt.update_sqrt(H, C, E, ctx=hip_ctx) if noise else dsqrt.update_sqrt_no_meascov(H, C, ctx=hip_ctx)
    want = o.update_sqrt(H, C, E)
    C_new, K, Sl = got
    wC, wK, wSl = want
    np.testing.assert_allclose(Sl, _canon_lower(wSl), rtol=1e-8, atol=1e-11 * np.abs(wSl).max())
    np.testing.assert_allclose(K, wK, rtol=1e-7, atol=1e-10 * np.abs(wK).max())
    np.testing.assert_allclose(C_new @ C_new.T, wC @ wC.T, rtol=1e-9, atol=1e-11 * np.abs(wC @ wC.T).max())
    assert np.all(np.triu(C_new, 1) == 0) and np.all(np.diag(C_new) >= 0)
