"""Golden fixtures (tests/golden/*.npz, produced by the oracle -- see make_golden.py for provenance)."""

import pathlib

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from helpers import assert_mean_std_parity

GOLD = pathlib.Path(__file__).parent / "golden"
CASES = ["oracle_heat_smoke_dirichlet", "oracle_heat_smoke_neumann", "oracle_heat_n32_nu1"]


def _setup(mod, g):
    N, nu, dt, tmax, dx = g["config"]
    prior = (mod.SquareExponential() if N == 6 else mod.Matern52()) + mod.WhiteNoise()
    return int(nu), float(dt), float(tmax), float(dx), str(g["bcond"]), prior


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    g = np.load(GOLD / f"{name}.npz")
    nu, dt, tmax, dx, bcond, prior = _setup(o, g)
    pde = o.heat_1d_discretized(tmax=tmax, dx=dx, diffusion_rate=0.05, kernel=o.SquareExponential(), bcond=bcond)
    np.testing.assert_allclose(pde.L, g["L"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pde.y0, g["y0"], rtol=1e-14, atol=1e-16)
    s = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(dt), spatial_kernel=prior, canonical_factor_signs=True)
    sol = s.solve(pde)
    means, stds = o.read_mean_and_std(sol, s.E0)
    assert np.array_equal(sol.t, g["t"])
    k = -1 if dt == 0.1 else None       # the runt step is only resolved to ~1e-3 by the algorithm itself (quirk Q2)
    np.testing.assert_allclose(means[:k], g["means"][:k], rtol=1e-7, atol=1e-10 * np.abs(g["means"]).max())
    np.testing.assert_allclose(stds[:k], g["stds"][:k], rtol=1e-6, atol=1e-9 * np.abs(g["stds"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_matches_golden(hip_ctx, name):
    g = np.load(GOLD / f"{name}.npz")
    nu, dt, tmax, dx, bcond, prior = _setup(pnmol.kernels, g)
    pde = pnmol.pde.examples.heat_1d_discretized(tmax=tmax, dx=dx, diffusion_rate=0.05,
                                                 kernel=pnmol.kernels.SquareExponential(), bcond=bcond)
    solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
                                             spatial_kernel=prior)
    t, means, stds, sig, _ = solver.solve_marginals(pde)
    assert np.array_equal(t, g["t"])
    k = -1 if dt == 0.1 else None
    assert_mean_std_parity(means[:k], stds[:k], g["means"][:k], g["stds"][:k])
    if k is None:
        np.testing.assert_allclose(np.mean(sig), g["diffusion_squared_calibrated_canonical"], rtol=1e-6)


# ---- semilinear / latent-force / system solvers (fixtures made by golden/make_golden.py:MODEL_CASES) ------------------
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("make_golden", GOLD / "make_golden.py")
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)
MODEL_CASES = list(make_golden.MODEL_CASES)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_oracle_reproduces_model_golden(name):
    g = np.load(GOLD / f"{name}.npz")
    pde, k, latent, semilinear, dt = make_golden.build_model(o, o, None, None, name)
    np.testing.assert_allclose(pde.L, g["L"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(pde.y0, g["y0"], rtol=1e-14, atol=1e-16)
    s = (o.LatentForceEK1 if latent else o.WhiteNoiseEK1)(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=k,
                                                          semilinear=semilinear, canonical_factor_signs=True)
    sol = s.solve(pde)
    E0 = s.state_iwp.projection_matrix(0) if latent else s.E0
    means, stds = (o.read_mean_and_std_latent if latent else o.read_mean_and_std)(sol, E0)
    d = pde.y0.shape[0]
    assert np.array_equal(sol.t, g["t"])
    np.testing.assert_allclose(means[:, :d], g["means"], rtol=1e-7, atol=1e-10 * np.abs(g["means"]).max())
    np.testing.assert_allclose(stds[:, :d], g["stds"], rtol=1e-6, atol=1e-9 * np.abs(g["stds"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", MODEL_CASES)
def test_gpu_matches_model_golden(hip_ctx, name):
    g = np.load(GOLD / f"{name}.npz")
    recipe, kw, ncomp, latent, dt, K, kname = make_golden.MODEL_CASES[name]
    pde, k, latent, semilinear, dt = make_golden.build_model(pnmol.kernels, pnmol.pde.examples, None, None, name)
    mod = pnmol.latent if latent else pnmol.white
    cls = getattr(mod, ("SemiLinear" if semilinear else "Linear") + ("LatentForceEK1" if latent else "WhiteNoiseEK1"))
    sol = cls(num_derivatives=2, steprule=pnmol.odetools.step.Constant(dt), spatial_kernel=k).solve(pde)
    d = pde.y0.shape[0]
    N = d // ncomp
    assert np.array_equal(sol.t, g["t"])
    m, s = sol.mean[:, 0, :d], sol.marginal_std[:, 0, :d]
    for c in range(ncomp):            # per component: SIR's compartments differ by three orders of magnitude
        sl = slice(c * N, (c + 1) * N)
        assert_mean_std_parity(m[:, sl], s[:, sl], g["means"][:, sl], g["stds"][:, sl])
    np.testing.assert_allclose(sol.diffusion_squared_calibrated, g["diffusion_squared_calibrated_canonical"], rtol=1e-5)
