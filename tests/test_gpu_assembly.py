"""On-device problem assembly: the batched kernel-FD stencil solves of
`discretize.fd_coefficients` (/root/reference/src/pnmol/discretize.py:177-201, vmapped at :60,75-80) and
Gamma = chol(spatial_kernel(X, X.T)) (/root/reference/src/pnmol/white.py:84-85), both through the C ABI."""

import numpy as np
import pytest

import pnmol
import pnmol_oracle as oracle
from pnmol import _hip, diffops, discretize, kernels, mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("s", [2, 3, 5, 9, 16])
def test_batched_stencil_solves_match_lapack(hip_ctx, s):
    rng = np.random.default_rng(s)
    N = 1500
    A = rng.standard_normal((N, s, s))
    gram = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(s)          # SPD, condition number O(10..100)
    lk, llk = rng.standard_normal((N, s)), rng.standard_normal(N)
    w, u = hip_ctx.fd_solve_batched(gram, lk, llk)
    w_ref = np.linalg.solve(gram, lk[..., None])[..., 0]
    np.testing.assert_allclose(w, w_ref, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(u, llk - np.einsum("ns,ns->n", w_ref, lk), rtol=1e-11, atol=1e-12)
    # a system that NEEDS the row exchange (zero leading pivot): partial pivoting as in LAPACK
    gram[:, 0, 0] = 0.0
    w2, _ = hip_ctx.fd_solve_batched(gram, lk, llk)
    np.testing.assert_allclose(w2, np.linalg.solve(gram, lk[..., None])[..., 0], rtol=1e-9, atol=1e-11)


def test_fd_known_answer_on_the_device(hip_ctx, monkeypatch):
    """reference tests/test_discretize.py:52-71: Polynomial(order 2, const 1) kernel, dx = 0.1, neighbours (1, 0, 2):
    weights * dx^2 = [-2, 1, 1], uncertainty 0 -- with the stencil systems solved by `pnmol_fd_solve_batched`."""
    monkeypatch.setenv("PNMOL_FD_ON_DEVICE", "1")
    dx = 0.1
    grid = mesh.RectangularMesh.from_bbox_1d(np.array([0.0, 1.0]), dx)
    k = kernels.Polynomial(const=1.0)
    lap = diffops.laplace()
    L_k = kernels.Lambda(lap(k.pairwise, argnums=0), parent=k, spec=(("laplace", 0),))
    LL_k = kernels.Lambda(lap(L_k.pairwise, argnums=1))
    w, unc = discretize.fd_coefficients(x=grid[1], neighbors=grid[((1, 0, 2),)], k=k, L_k=L_k, LL_k=LL_k)
    np.testing.assert_allclose(w * dx ** 2, [-2.0, 1.0, 1.0], atol=1e-8)
    np.testing.assert_allclose(unc, 0.0, atol=1e-8)


def test_device_assembled_operator_matches_the_oracle(hip_ctx, monkeypatch):
    """L, E_sqrtm of a heat problem assembled with device stencil solves against the oracle's (LAPACK) assembly.  With the
    default square-exponential FD kernel the 3 x 3 Gram systems have cond ~ (1/dx)^4, so the two LU codes agree to
    cond * eps, not to eps: the tolerance says so (and is why the device path is opt-in, discretize._fd_batched)."""
    monkeypatch.setenv("PNMOL_FD_ON_DEVICE", "1")
    for dx, tol in ((0.1, 1e-9), (1.0 / 31, 1e-7)):
        kw = dict(dx=dx, tmax=0.1, diffusion_rate=0.05, bcond="dirichlet")
        p = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.SquareExponential(), **kw)
        q = oracle.heat_1d_discretized(kernel=oracle.SquareExponential(), **kw)
        np.testing.assert_allclose(p.L, q.L, rtol=tol, atol=tol * np.abs(q.L).max())
        np.testing.assert_allclose(np.diag(p.E_sqrtm), np.diag(q.E_sqrtm), rtol=0, atol=tol * np.abs(q.L).max())


@pytest.mark.parametrize("n", [7, 100, 544, 700, 1300])
def test_device_cholesky_matches_lapack(hip_ctx, n):
    """n <= 544 runs the register-resident sweep kernel, above it the left-looking one (both are the step's own kernels)."""
    x = np.linspace(0.0, 1.0, n)[:, None]
    K = (kernels.Matern52() + kernels.WhiteNoise())(x, x.T)
    L = hip_ctx.cholesky(K)
    Lref = np.linalg.cholesky(K)
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    np.testing.assert_allclose(L, Lref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(L @ L.T, K, rtol=1e-12, atol=1e-13)


def test_device_cholesky_reports_indefinite_matrices(hip_ctx):
    A = np.eye(40)
    A[17, 17] = -1.0
    with pytest.raises(_hip.PnmolHipError, match="not positive definite at pivot 17"):
        hip_ctx.cholesky(A)


def test_solver_with_device_cholesky_matches_the_oracle(hip_ctx, monkeypatch):
    """initialize + 6 steps with Gamma from the device against the oracle (north_star tolerances)."""
    from helpers import assert_mean_std_parity, make_pair
    monkeypatch.setenv("PNMOL_CHOL_ON_DEVICE", "1")
    pde, solver, opde, osolver = make_pair(96, 2, 2.0 ** -7, 6)
    t, means, stds, sig, _ = solver.solve_marginals(pde)
    osol = osolver.solve(opde)
    om, os_ = oracle.read_mean_and_std(osol, osolver.E0)
    assert_mean_std_parity(means, stds, om, os_)
