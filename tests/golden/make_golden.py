"""Generates tests/golden/*.npz.

PROVENANCE: these vectors are produced by THIS repository's CPU oracle (oracle/pnmol_oracle.py), NOT by the
reference implementation -- the reference (pure Python on JAX) cannot be imported in the build container (jax,
jaxlib, tornadox absent; no network), and it ships no numerical fixture of its own for the filter output.  They
freeze the oracle's output so that a change to the oracle (or to LAPACK's behaviour) is noticed, and give the GPU
tests a fixture that does not need the oracle at run time.  Inputs: the reference's own smoke configuration
(tests/test_pdefilter.py:15-64: heat, dx=0.2, tmax=1, Constant(0.1), nu=2, SE+WhiteNoise) and BASELINE config 0
(N=32, nu=1, dt=2^-7, 100 steps).  Run:  python tests/golden/make_golden.py
"""
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import pnmol_oracle as o  # noqa: E402


def run(name, *, N, nu, dt, tmax, dx, bcond, prior):
    pde = o.heat_1d_discretized(tmax=tmax, dx=dx, diffusion_rate=0.05, kernel=o.SquareExponential(), bcond=bcond)
    s = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(dt), spatial_kernel=prior, canonical_factor_signs=True)
    sol = s.solve(pde)
    means, stds = o.read_mean_and_std(sol, s.E0)
    s2 = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(dt), spatial_kernel=prior)
    d2_as_written = s2.solve(pde).diffusion_squared_calibrated
    np.savez_compressed(pathlib.Path(__file__).parent / f"{name}.npz", t=sol.t, means=means, stds=stds,
                        mean_all=sol.mean, L=pde.L, E_diag=np.diag(pde.E_sqrtm), B=pde.B, R_sqrtm=pde.R_sqrtm, y0=pde.y0,
                        diffusion_squared_calibrated_canonical=sol.diffusion_squared_calibrated,
                        diffusion_squared_calibrated_as_written=d2_as_written,
                        config=np.array([N, nu, dt, tmax, dx]), bcond=bcond)
    print(name, sol.t.shape, means.shape)


# ---- f2/f3 solvers: semilinear, latent-force and system recipes (oracle output as above, same provenance) ----------
MODEL_CASES = {
    # name: (recipe, recipe kwargs, n components, latent, dt, steps, prior kernel name)
    "oracle_spruce_white": ("spruce_budworm_1d_discretized", dict(dx=0.1), 1, False, 2.0 ** -4, 8, "SquareExponential"),
    "oracle_heat_latent_neumann": ("heat_1d_discretized", dict(dx=0.1, bcond="neumann"), 1, True, 2.0 ** -5, 8,
                                   "SquareExponential"),
    "oracle_lv_white": ("lotka_volterra_1d_discretized", dict(dx=1.0 / 15), 2, False, 2.0 ** -5, 8, "Matern52"),
    "oracle_lv_latent": ("lotka_volterra_1d_discretized", dict(dx=1.0 / 15), 2, True, 2.0 ** -5, 8, "Matern52"),
    "oracle_sir_white": ("sir_1d_discretized", dict(dx=1.0 / 12), 3, False, 2.0 ** -3, 8, "Matern52"),
}


def build_model(mod, examples, latent_mod, white_mod, name):
    """The same case in `mod` = the oracle or the product's host mirror (used by tests/test_golden.py too)."""
    recipe, kw, ncomp, latent, dt, K, kname = MODEL_CASES[name]
    pde = getattr(examples, recipe)(tmax=K * dt, **kw)
    k = getattr(mod, kname)() + mod.WhiteNoise()
    if ncomp > 1:
        k = mod.duplicate(k, ncomp)
    semilinear = recipe != "heat_1d_discretized"
    return pde, k, latent, semilinear, dt


def run_model(name):
    pde, k, latent, semilinear, dt = build_model(o, o, None, None, name)
    cls = o.LatentForceEK1 if latent else o.WhiteNoiseEK1
    s = cls(num_derivatives=2, steprule=o.Constant(dt), spatial_kernel=k, semilinear=semilinear,
            canonical_factor_signs=True)
    sol = s.solve(pde)
    if latent:
        means, stds = o.read_mean_and_std_latent(sol, s.state_iwp.projection_matrix(0))
    else:
        means, stds = o.read_mean_and_std(sol, s.E0)
    d = pde.y0.shape[0]
    np.savez_compressed(pathlib.Path(__file__).parent / f"{name}.npz", t=sol.t, means=means[:, :d], stds=stds[:, :d],
                        L=pde.L, y0=pde.y0, diffusion_squared_calibrated_canonical=sol.diffusion_squared_calibrated)
    print(name, sol.t.shape, means.shape, float(np.abs(means).max()), float(stds.max()))


if __name__ == "__main__":
    for name in MODEL_CASES:
        run_model(name)
    for bc in ("dirichlet", "neumann"):
        run(f"oracle_heat_smoke_{bc}", N=6, nu=2, dt=0.1, tmax=1.0, dx=0.2, bcond=bc, prior=o.SquareExponential() + o.WhiteNoise())
    run("oracle_heat_n32_nu1", N=32, nu=1, dt=2.0**-7, tmax=100 * 2.0**-7, dx=1.0 / 31, bcond="dirichlet",
        prior=o.Matern52() + o.WhiteNoise())
