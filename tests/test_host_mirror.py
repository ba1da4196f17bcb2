"""CPU tests of the product's host side: the `pnmol` API mirror (cold path) against the oracle and the
reference's known answers, and the C-ABI library (loads, exports every declared symbol; no compute)."""

import ctypes
import pathlib
import re

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from pnmol import _hip, diffops, discretize, kernels, mesh

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.mark.parametrize("N,bcond", [(6, "dirichlet"), (6, "neumann"), (33, "dirichlet"), (128, "neumann")])
def test_heat_problem_matches_oracle(N, bcond):
    kw = dict(tmax=1.0, dx=1.0 / (N - 1), diffusion_rate=0.035, bcond=bcond)
    p = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.SquareExponential(), **kw)
    q = o.heat_1d_discretized(kernel=o.SquareExponential(), **kw)
    for name in ("L", "B", "R_sqrtm", "y0"):
        np.testing.assert_allclose(getattr(p, name), getattr(q, name), rtol=1e-13, atol=1e-15, err_msg=name)
    np.testing.assert_allclose(p.E_sqrtm, q.E_sqrtm, rtol=1e-6, atol=1e-12 * np.abs(p.L).max())
    assert p.L.shape == (N, N) and p.mesh_spatial.points.shape == (N, 1) and p.dimension == 1
    assert p.is_discretized and p.t_span == (0.0, 1.0)


def test_matern_fd_kernel_matches_oracle():  # the figure1 configuration (experiments/figure1.py:118-131)
    kw = dict(t0=0.0, tmax=3.0, dx=0.2, stencil_size_interior=3, stencil_size_boundary=4, diffusion_rate=0.035)
    p = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.Matern52(), **kw)
    q = o.heat_1d_discretized(kernel=o.Matern52(), **kw)
    np.testing.assert_allclose(p.L, q.L, rtol=1e-12)
    np.testing.assert_allclose(p.E_sqrtm, q.E_sqrtm, rtol=1e-9, atol=1e-14)


def test_fd_coefficients_known_answer():  # reference tests/test_discretize.py:52-71
    dx = 0.1
    grid = mesh.RectangularMesh.from_bbox_1d(np.array([0.0, 1.0]), dx)
    k = kernels.Polynomial(const=1.0)
    lap = diffops.laplace()
    L_k = kernels.Lambda(lap(k.pairwise, argnums=0), parent=k, spec=(("laplace", 0),))
    LL_k = kernels.Lambda(lap(L_k.pairwise, argnums=1))
    w, unc = discretize.fd_coefficients(x=grid[1], neighbors=grid[((1, 0, 2),)], k=k, L_k=L_k, LL_k=LL_k)
    np.testing.assert_allclose(w * dx**2, [-2.0, 1.0, 1.0], atol=1e-8)
    np.testing.assert_allclose(unc, 0.0, atol=1e-8)
    L, E = discretize.fd_probabilistic(diffop=lap, mesh_spatial=grid)
    assert L.shape == E.shape == (11, 11)
    np.testing.assert_allclose(E, np.diag(np.diag(E)))
    Bn, Rn = discretize.fd_probabilistic_neumann_1d(grid)
    assert Bn.shape == (2, 11) and Rn.shape == (2, 2)


def test_kernel_call_conventions():  # reference tests/test_kernels.py:51-145
    X = np.linspace(0, 1, 7).reshape(-1, 1)
    for k in (kernels.SquareExponential(), kernels.Matern52(), kernels.Polynomial(), kernels.WhiteNoise(),
              kernels.Matern52() + kernels.WhiteNoise()):
        assert np.shape(k(X[0], X[1])) == ()
        assert k(X, X).shape == (7,)
        assert k(X, X[:5].T).shape == (7, 5)
    np.testing.assert_allclose(kernels.WhiteNoise(output_scale=2.0)(X, X.T), 4.0 * np.eye(7))
    G = (kernels.Matern52() + kernels.WhiteNoise())(X, X.T)
    np.testing.assert_allclose(G, o.Matern52()(X, X.T) + np.eye(7))
    assert kernels.duplicate(kernels.SquareExponential(), 3)(X, X.T).shape == (21, 21)
    with pytest.raises(NotImplementedError):
        diffops.laplace()(lambda x, y: x @ y)


def test_mesh_and_steprules_match_reference_tests():
    g = mesh.RectangularMesh.from_bbox_2d(bbox=[[0.0, 0.0], [1.0, 1.0]], steps=(0.1, 0.1))
    assert g.ndim == 2 and g.dimension == 2 and len(g) == 121
    np.testing.assert_array_equal(g.boundary_projection_matrix @ g.points, g.boundary[0])
    c = pnmol.odetools.step.Constant(0.1)
    assert c.suggest(np.nan, 0.1) == 0.1 and c.is_accepted(0.1) and c.scale_error_estimate(None, None) is None
    a = pnmol.odetools.step.Adaptive(abstol=0.1, reltol=0.01)
    err, ref = np.array([0.5, 0.6]), np.array([2.0, 3.0])
    np.testing.assert_allclose(a.scale_error_estimate(err, ref), o.Adaptive(0.1, 0.01).scale_error_estimate(err, ref))
    assert a.first_dt(pnmol.pde.examples.heat_1d_discretized()) > 0


def test_iwp_mirror():
    w = pnmol.base.iwp.IntegratedWienerTransition(wiener_process_dimension=2, num_derivatives=2,
                                                  wp_diffusion_sqrtm=np.array([[1.0, 0.0], [0.3, 0.9]]))
    wo = o.IWP(2, 2, np.array([[1.0, 0.0], [0.3, 0.9]]))
    for a, b in zip(w.non_preconditioned_discretize(0.1), wo.non_preconditioned_discretize(0.1)):
        np.testing.assert_allclose(a, b)
    np.testing.assert_array_equal(w.projection_matrix(1), wo.projection_matrix(1))
    assert w.state_dimension == 6


@pytest.mark.parametrize("N,nu,bcond", [(6, 2, "dirichlet"), (40, 2, "neumann"), (40, 1, "dirichlet"), (96, 2, "dirichlet")])
def test_structured_initialisation_matches_oracle(N, nu, bcond):
    """`initialize` evaluates white.py:12-80 block-wise in closed form on the host (cold path): same mean and
    covariance as the reference's two QR updates."""
    kw = dict(tmax=1.0, dx=1.0 / (N - 1), diffusion_rate=0.05, bcond=bcond)
    p = pnmol.pde.examples.heat_1d_discretized(kernel=kernels.SquareExponential(), **kw)
    s = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(0.1),
                                        spatial_kernel=kernels.Matern52() + kernels.WhiteNoise())
    s.iwp, s.E0, s.E1, gamma = s.initialize_iwp(p)
    s._gram = gamma @ gamma.T
    mean, blocks = s._initial_moments(p)
    so = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(0.1), spatial_kernel=o.Matern52() + o.WhiteNoise())
    st = so.initialize(o.heat_1d_discretized(kernel=o.SquareExponential(), **kw))
    cov = st.y.cov_sqrtm @ st.y.cov_sqrtm.T
    n = nu + 1
    np.testing.assert_allclose(mean, st.y.mean, rtol=1e-7, atol=1e-9 * np.abs(st.y.mean).max())
    for (a, b), blk in blocks.items():
        ref = cov[a::n, b::n]
        tol = 1e-6 if a == b else 1e-3    # the tiny cross blocks are themselves only resolved to ~1e-5 by the QR form
        np.testing.assert_allclose(blk, ref, rtol=tol, atol=tol * np.abs(ref).max())
    sq_inv, sq_diag = s._error_model(p, 2.0**-7)   # estimate_error's S (white.py:153-162), step-invariant part
    P, _ = so.iwp.nordsieck_preconditioner(2.0**-7)
    _, Ql = so.iwp.preconditioned_discretize
    opde = o.heat_1d_discretized(kernel=o.SquareExponential(), **kw)
    _, H, E = so.evaluate_ode(opde, so.E0 @ P, so.E1 @ P, np.zeros(n * N), 0.0)
    Sq = H @ Ql @ Ql.T @ H.T + E @ E.T
    np.testing.assert_allclose(sq_diag, np.diag(Sq), rtol=1e-10)
    np.testing.assert_allclose(sq_inv @ Sq, np.eye(Sq.shape[0]), atol=1e-6)


def test_library_exports_every_declared_symbol():
    """dlopen works without a GPU and every function include/pnmol_hip.h declares is exported and bound."""
    header = (ROOT / "include" / "pnmol_hip.h").read_text() + (ROOT / "include" / "pnmol_sqrt.h").read_text()
    declared = set(re.findall(r"\b(pnmol_[a-z_0-9]+)\s*\(", header))
    lib = _hip.load_library()
    assert declared == set(_hip.SYMBOLS), declared ^ set(_hip.SYMBOLS)
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    assert lib.pnmol_abi_version() == 3
    n = ctypes.c_int(-1)
    lib.pnmol_device_count(ctypes.byref(n))
    assert n.value >= 0
    assert lib.pnmol_filter_destroy(None) == -1 and lib.pnmol_state_destroy(None) == -1   # argument checks, no GPU


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    """The product must not route through the oracle, and the solver must fail loudly without a device."""
    for path in (ROOT / "pnmol-experiments_amd").rglob("*.py"):
        assert "pnmol_oracle" not in path.read_text(), path
    n = ctypes.c_int(0)
    _hip.load_library().pnmol_device_count(ctypes.byref(n))
    if n.value == 0:
        p = pnmol.pde.examples.heat_1d_discretized(dx=0.2)
        s = pnmol.white.LinearWhiteNoiseEK1(steprule=pnmol.odetools.step.Constant(0.1))
        with pytest.raises(_hip.PnmolHipError):
            s.initialize(p)


@pytest.mark.parametrize("bcond", ["dirichlet", "neumann"])
def test_spruce_budworm_recipe_matches_oracle(bcond):  # reference pde/examples.py:251-341
    kw = dict(tmax=1.0, dx=0.1, diffusion_rate=0.05, bcond=bcond)
    p = pnmol.pde.examples.spruce_budworm_1d_discretized(kernel=kernels.SquareExponential(), **kw)
    q = o.spruce_budworm_1d_discretized(kernel=o.SquareExponential(), **kw)
    for name in ("L", "B", "R_sqrtm", "y0"):
        np.testing.assert_allclose(getattr(p, name), getattr(q, name), rtol=1e-13, atol=1e-15)
    x = np.linspace(0.05, 0.4, 11)
    np.testing.assert_allclose(p.f(0.3, x), q.f(0.3, x))
    np.testing.assert_allclose(p.df(0.3, x), q.df(0.3, x))
    h = 1e-6   # df is the Jacobian of f
    J = np.stack([(p.f(0, x + h * e) - p.f(0, x - h * e)) / (2 * h) for e in np.eye(11)], axis=1)
    np.testing.assert_allclose(p.df(0, x), J, atol=1e-8)
    M, shift = pnmol.white.SemiLinearWhiteNoiseEK1._linearize(p, x, 0.0)
    np.testing.assert_allclose(M, p.L + p.df(0, x))
    np.testing.assert_allclose(shift, p.df(0, x) @ x - p.f(0, x))
