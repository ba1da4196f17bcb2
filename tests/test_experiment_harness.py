"""The callers on the far side of the path: what experiments/figure3.py and figure4.py do with a solver
(`simulate_final_state`, `y.cov_sqrtm @ y.cov_sqrtm.T`, `solver.E0 @ cov @ solver.E0.T`, block extraction, RMSE and
chi^2 against a high-resolution MOL solution).  The harness code below follows figure3.py:37-58,87-93,225-248 and
figure4.py:10-50,75-160 step by step with numpy in place of jax.numpy; it is run once on the GPU solvers and once on
the oracle, and every quantity the figures save is compared."""

import numpy as np
import pytest
import scipy.integrate

import pnmol
import pnmol_oracle as o
from pnmol import kernels
from pnmol.pde import examples

pytestmark = pytest.mark.gpu


def _reference_u(pde_ref, ref_scale, ncomp):
    """figure4.py:23-50 / figure3.py:20-35: high-resolution method-of-lines solution, first component, coarse nodes."""
    ivp = pde_ref.to_ivp()
    sol = scipy.integrate.solve_ivp(ivp.f, (ivp.t0, ivp.tmax), ivp.y0, method="LSODA", atol=1e-10, rtol=1e-10,
                                    t_eval=(ivp.t0, ivp.tmax))
    u_full = np.split(sol.y[:, -1], ncomp)[0]
    return u_full[(ref_scale - 1)::ref_scale]


def _figure4_white(solver, pde, u_reference):
    """figure4.py:122-160."""
    final, info = solver.simulate_final_state(pde, progressbar=False)
    u_full, _ = np.split(np.asarray(final.y.mean[0]), 2)
    u = u_full[1:-1]
    C = np.asarray(final.y.cov_sqrtm)
    cov = solver.E0 @ (C @ C.T) @ solver.E0.T
    cov_u = np.split(np.split(cov, 2, axis=-1)[0], 2, axis=0)[0][1:-1, 1:-1]
    return _scores(u, cov_u, u_reference)


def _figure4_latent(solver, pde, u_reference):
    """figure4.py:75-118."""
    final, info = solver.simulate_final_state(pde, progressbar=False)
    mean_state, _ = np.split(np.asarray(final.y.mean[0]), 2)
    u_full, _ = np.split(mean_state, 2)
    u = u_full[1:-1]
    C = np.asarray(final.y.cov_sqrtm)
    cov = C @ C.T
    cov_no_xi = np.split(np.split(cov, 2, axis=-1)[0], 2, axis=0)[0]
    cov_int = solver.E0 @ cov_no_xi @ solver.E0.T
    cov_u = np.split(np.split(cov_int, 2, axis=-1)[0], 2, axis=0)[0][1:-1, 1:-1]
    return _scores(u, cov_u, u_reference)


def _figure3_white(solver, pde, i_reference):
    """figure3.py:37-58,87-93: SIR, the *second* component's block is what the script keeps (`blocks[1][1]`) next to
    the first component's mean -- reproduced as written."""
    final, _ = solver.simulate_final_state(pde, progressbar=False)
    E0 = solver.iwp.projection_matrix(0)
    mean = np.asarray(final.y.mean[0, :])
    C = np.asarray(final.y.cov_sqrtm)
    cov = E0 @ (C @ C.T) @ E0.T
    std = np.sqrt(np.diagonal(cov))
    i_mean, i_std = np.split(mean, 3)[0][1:-1], np.split(std, 3)[0][1:-1]
    blocks = [np.split(c_row, 3, axis=1) for c_row in np.split(cov, 3, axis=0)]
    i_cov = blocks[1][1][1:-1, 1:-1]
    out = _scores(i_mean, i_cov, i_reference)
    out["mean_std"] = np.mean(i_std)
    return out


def _scores(u, cov_u, u_reference):
    err = np.abs(u - u_reference)
    return dict(u=u, cov_u=cov_u, rmse=np.linalg.norm(err / np.abs(u_reference)) / np.sqrt(u.size),
                chi2=err @ np.linalg.solve(cov_u, err) / err.shape[0])


def _compare(got, want):
    np.testing.assert_allclose(got["u"], want["u"], rtol=1e-5, atol=1e-12)          # north_star: mean rtol 1e-5
    sd = np.sqrt(np.diag(want["cov_u"]))
    np.testing.assert_allclose(np.sqrt(np.diag(got["cov_u"])), sd, rtol=1e-4)        # north_star: std rtol 1e-4
    assert np.max(np.abs(got["cov_u"] - want["cov_u"]) / np.outer(sd, sd)) < 1e-4    # correlations to the same level
    np.testing.assert_allclose(got["rmse"], want["rmse"], rtol=1e-4)
    # chi^2 inverts cov_u: its relative accuracy is cond(cov_u) x (covariance accuracy) in either implementation; where
    # cov_u is numerically singular (latent model at this size: cond ~ 1e17) the figure's number is rounding noise in the
    # reference too and is not compared
    cond = np.linalg.cond(want["cov_u"])
    if cond < 1e8:
        np.testing.assert_allclose(got["chi2"], want["chi2"], rtol=1e-4 * max(1.0, cond * 1e-4))


@pytest.mark.parametrize("latent", [False, True])
def test_figure4_pipeline(latent):
    dx, dt, ref_scale = 1.0 / 15, 2.0 ** -4, 3      # power-of-two dt: exactly 8 steps, no runt step of ~1e-17 (whose
    # Nordsieck scaling dt^(nu+1/2) is below fp64 resolution in any implementation; covered by the no-NaN smoke tests)
    kw = dict(t0=0.0, tmax=0.5, stencil_size_interior=3, stencil_size_boundary=4)
    pde = examples.lotka_volterra_1d_discretized(dx=dx, **kw)
    opde = o.lotka_volterra_1d_discretized(dx=dx, **kw)
    u_ref = _reference_u(examples.lotka_volterra_1d_discretized(dx=dx / ref_scale, **kw), ref_scale, 2)
    k = kernels.duplicate(kernels.Matern52() + kernels.WhiteNoise(), num=2)
    ok = o.duplicate(o.Matern52() + o.WhiteNoise(), 2)
    if latent:
        s = pnmol.latent.SemiLinearLatentForceEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt),
                                                  spatial_kernel=k)
        os_ = o.LatentForceEK1(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=True, canonical_factor_signs=True)
        _compare(_figure4_latent(s, pde, u_ref), _figure4_latent(os_, opde, u_ref))
    else:
        s = pnmol.white.SemiLinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt),
                                                spatial_kernel=k)
        os_ = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=True, canonical_factor_signs=True)
        _compare(_figure4_white(s, pde, u_ref), _figure4_white(os_, opde, u_ref))


def test_figure3_pipeline():
    dx, dt, ref_scale = 1.0 / 12, 0.25, 3
    kw = dict(t0=0.0, tmax=2.0)
    pde = examples.sir_1d_discretized(dx=dx, **kw)
    opde = o.sir_1d_discretized(dx=dx, **kw)
    i_ref = _reference_u(examples.sir_1d_discretized(dx=dx / ref_scale, **kw), ref_scale, 3)
    k = kernels.duplicate(kernels.Matern52() + kernels.WhiteNoise(), num=3)
    ok = o.duplicate(o.Matern52() + o.WhiteNoise(), 3)
    s = pnmol.white.SemiLinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k)
    os_ = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=ok, semilinear=True, canonical_factor_signs=True)
    got, want = _figure3_white(s, pde, i_ref), _figure3_white(os_, opde, i_ref)
    _compare(got, want)
    np.testing.assert_allclose(got["mean_std"], want["mean_std"], rtol=1e-4)
