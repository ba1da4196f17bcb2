"""Full-length runs of the bench workload against oracle fixtures (tests/golden/oracle_heat_n{N}_nu2_k{g}.npz).

BASELINE config 4: the 8-problem diffusion sweep kappa_g = 0.01 * 10^(g/7) at N=512, nu=2, dt=2^-7, 100 steps, and
config 3 (N=1024) with 24 steps.  The fixtures are produced by THIS repository's CPU oracle (the reference algorithm
as written; see tests/golden/make_golden_sweep.py for provenance -- the reference itself cannot run here).  On the GPU
every problem is solved serially and with all eight in flight on one device (`steps_begin` / `steps_end`); the
concurrent run must reproduce the serial one bit for bit.
"""

import pathlib

import numpy as np
import pytest

import pnmol
import pnmol_oracle as o
from helpers import assert_mean_std_parity
from pnmol import batch

GOLD = pathlib.Path(__file__).parent / "golden"
DT = 2.0 ** -7
SWEEP = {512: list(range(8)), 1024: [0, 3, 7]}


def _load(N, g):
    return np.load(GOLD / f"oracle_heat_n{N}_nu2_k{g}.npz")


def _problem(mod, examples, N, kappa, K):
    pde = examples.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3,
                                       stencil_size_boundary=3, t0=0.0, tmax=K * DT, diffusion_rate=kappa,
                                       kernel=mod.SquareExponential(), nugget_gram_matrix_fd=0.0, bcond="dirichlet")
    return pde, mod.Matern52() + mod.WhiteNoise()


@pytest.mark.parametrize("N", [512, 1024])
def test_fixtures_are_the_sweep(N):
    for g in SWEEP[N]:
        f = _load(N, g)
        K = int(f["config"][3])
        assert float(f["kappa"]) == batch.diffusion_sweep(g, 8)
        assert f["means"].shape == (K + 1, N) and f["stds"].shape == (K + 1, N) and f["sigma2"].shape == (K,)
        np.testing.assert_array_equal(f["t"], DT * np.arange(K + 1))
        assert np.all(np.isfinite(f["means"])) and np.all(f["stds"] >= 0) and np.all(f["sigma2"] > 0)
    ks = [float(_load(N, g)["kappa"]) for g in SWEEP[N]]
    assert ks == sorted(ks) and ks[0] == 0.01 and abs(ks[-1] - 0.1) < 1e-15


def test_oracle_reproduces_the_first_steps_of_a_fixture():
    """Ties the N=512 fixtures to the oracle in the CPU suite (2 steps: a few seconds)."""
    f = _load(512, 5)
    pde, prior = _problem(o, o, 512, float(f["kappa"]), 2)
    s = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(DT), spatial_kernel=prior, canonical_factor_signs=True)
    sol = s.solve(pde)
    means, stds = o.read_mean_and_std(sol, s.E0)
    np.testing.assert_allclose(means, f["means"][:3], rtol=1e-9, atol=1e-12 * np.abs(f["means"]).max())
    np.testing.assert_allclose(stds, f["stds"][:3], rtol=1e-7, atol=1e-10 * np.abs(f["stds"]).max())


def _solver_on_own_stream(N, g, K):
    from pnmol import _hip
    f = _load(N, g)
    pde, prior = _problem(pnmol.kernels, pnmol.pde.examples, N, float(f["kappa"]), K)
    solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=2, steprule=pnmol.odetools.step.Constant(DT),
                                             spatial_kernel=prior)
    solver._context = _hip.Context(0)          # own stream: the eight problems may run concurrently
    return f, pde, solver


def _check(f, means, stds, sig):
    assert_mean_std_parity(means, stds, f["means"][1:], f["stds"][1:])
    # quirk Q1: compared with the oracle's canonical-sign value.  |Ls^-T z|^2 / m is a difference-sensitive
    # scalar (values down to 1e-7 at kappa = 0.01): 4e-6 relative observed on 1 of 800 entries
    np.testing.assert_allclose(sig, f["sigma2"], rtol=2e-5)


@pytest.mark.gpu
def test_sweep_n512_serial_and_concurrent():
    K = 100
    runs = []
    for g in SWEEP[512]:
        f, pde, solver = _solver_on_own_stream(512, g, K)
        state = solver.initialize(pde)
        np.testing.assert_allclose(state.y.mean[0], f["means"][0], rtol=1e-5, atol=1e-5 * np.abs(f["means"][0]).max())
        flt, dev = solver._device_filter, state.y.device_state
        solver._ensure_error_model(pde, DT)
        twin = flt.new_state()
        twin.set(pde.t0, dev.mean(), dev.cov())
        m, s, infos = flt.steps(twin, K, DT)                                  # serial: this problem alone on the device
        assert all(i.info == -1 for i in infos)
        sig = np.array([i.diffusion_squared_local for i in infos])
        _check(f, m, s, sig)
        runs.append((f, flt, dev, m, s, sig))
    for _, flt, dev, *_ in runs:                                              # all eight in flight
        flt.steps_begin(dev, K, DT)
    for f, flt, dev, m_ser, s_ser, sig_ser in runs:
        m, s, infos = flt.steps_end(dev)
        assert all(i.info == -1 for i in infos)
        sig = np.array([i.diffusion_squared_local for i in infos])
        assert np.array_equal(m, m_ser) and np.array_equal(s, s_ser) and np.array_equal(sig, sig_ser)
        _check(f, m, s, sig)


@pytest.mark.gpu
@pytest.mark.parametrize("g", SWEEP[1024])
def test_sweep_n1024(g):
    f, pde, solver = _solver_on_own_stream(1024, g, 24)
    t, means, stds, sig, _ = solver.solve_marginals(pde)
    assert np.array_equal(t, f["t"])
    assert_mean_std_parity(means, stds, f["means"], f["stds"])
    # |Ls^-T z|^2 / m (quirk Q1) at N=1024: values of 1e4..1e7, cond(S) ~1e12 -- 1e-4 relative between two factorisation
    # orders (observed 9.7e-5 at kappa = 0.1); mean and std above are what north_star's tolerances are about
    np.testing.assert_allclose(sig, f["sigma2"], rtol=5e-4)


def test_fixture_n256_is_config_2():
    f = _load(256, "c")
    assert float(f["kappa"]) == 0.05 and tuple(f["config"]) == (256, 2, DT, 100)
    assert f["means"].shape == (101, 256) and f["stds"].shape == (101, 256) and f["sigma2"].shape == (100,)
    np.testing.assert_array_equal(f["t"], DT * np.arange(101))


@pytest.mark.gpu
def test_full_length_n256():
    """BASELINE config 2 (N=256, nu=2, fp64; the kappa of bench.py's secondary point) at full length: every one of its 100
    steps against the oracle fixture, mean 1e-5 / std 1e-4."""
    f, pde, solver = _solver_on_own_stream(256, "c", 100)
    t, means, stds, sig, _ = solver.solve_marginals(pde)
    assert np.array_equal(t, f["t"])
    assert_mean_std_parity(means, stds, f["means"], f["stds"])
    np.testing.assert_allclose(sig, f["sigma2"], rtol=2e-5)
