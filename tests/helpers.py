"""Shared test helpers: problem factories for product and oracle, layout conversions."""

import numpy as np

import pnmol
import pnmol_oracle as oracle


def make_pair(N, nu, dt, K, bcond="dirichlet", dx=None, kappa=0.05, canonical=True):
    """Same heat problem + solver in the product (`pnmol`) and in the oracle."""
    dx = 1.0 / (N - 1) if dx is None else dx
    kw = dict(tmax=K * dt, dx=dx, diffusion_rate=kappa, bcond=bcond, stencil_size_interior=3,
              stencil_size_boundary=3, nugget_gram_matrix_fd=0.0)
    pde = pnmol.pde.examples.heat_1d_discretized(kernel=pnmol.kernels.SquareExponential(), **kw)
    solver = pnmol.white.LinearWhiteNoiseEK1(
        num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
        spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    opde = oracle.heat_1d_discretized(kernel=oracle.SquareExponential(), **kw)
    osolver = oracle.WhiteNoiseEK1(num_derivatives=nu, steprule=oracle.Constant(dt),
                                   spatial_kernel=oracle.Matern52() + oracle.WhiteNoise(),
                                   canonical_factor_signs=canonical)
    return pde, solver, opde, osolver


def to_device_layout(M, n, d, dp):
    """(D,D) in the reference's F-order (index j*n+a) -> (n*dp, n*dp) derivative-major, zero padded."""
    out = np.zeros((n * dp, n * dp))
    for a in range(n):
        for b in range(n):
            out[a * dp:a * dp + d, b * dp:b * dp + d] = M[a::n, b::n]
    return out


def assert_mean_std_parity(means, stds, omeans, ostds):
    """north_star tolerances: mean rtol 1e-5, std rtol 1e-4.  The absolute floors cover entries whose
    exact value is 0 (Dirichlet boundary nodes): the covariance form resolves a variance to
    eps*|P-| (|P-| is dominated by the highest derivative's prior, ~1e2..1e3 in the Nordsieck frame),
    i.e. a std to ~2e-6*max(std) -- measured 2.2e-6 at N=256; the floor is 1e-5*max(std)."""
    np.testing.assert_allclose(means, omeans, rtol=1e-5, atol=1e-5 * np.abs(omeans).max())
    np.testing.assert_allclose(stds, ostds, rtol=1e-4, atol=1e-5 * np.abs(ostds).max())
