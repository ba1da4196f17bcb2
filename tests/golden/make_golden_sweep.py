"""Generates tests/golden/oracle_heat_n{N}_nu2_k{g}.npz: full-length runs of the bench workload by the CPU oracle.

PROVENANCE: produced by THIS repository's CPU oracle (oracle/pnmol_oracle.py: the reference algorithm as written, two
QRs per step), NOT by the reference implementation (it cannot be imported here: no jax).

Workload = bench.py's / BASELINE config 4: 1-D heat equation, Dirichlet, dx = 1/(N-1), SE finite-difference kernel,
prior Matern52 + WhiteNoise, nu = 2, dt = 2^-7 (no runt step), diffusion sweep kappa_g = 0.01 * 10^(g/7), g = 0..7.
  N = 512 : 100 steps for each of the 8 problems  (about 1-2 min each on 8 cores)
  N = 1024: 24 steps for g = 0, 3 and 7           (about 8 s per step)
  N = 256 : 100 steps, kappa = 0.05 (BASELINE config 2; the value bench.py's secondary point runs) -> ..._n256_nu2_kc.npz
Stored per problem: t (T+1), means (T+1, d) = sol.mean[:, 0], stds (T+1, d) = sqrt(diag(C C^T) E0^T)
(experiments/figure1.py:76-80), sigma2 (T) = diffusion_squared_local with canonical factor signs,
kappa.  float64, compressed.  Run:  python tests/golden/make_golden_sweep.py [512|1024] [g ...]
"""
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import pnmol_oracle as o  # noqa: E402

DT = 2.0 ** -7
STEPS = {512: 100, 1024: 24, 256: 100}
PROBLEMS = {512: list(range(8)), 1024: [0, 3, 7], 256: ["c"]}


def kappa_of(g, count=8, lo=0.01, hi=0.1):   # pnmol/batch.py:diffusion_sweep (kept in step by tests/test_sweep_golden.py)
    return float(lo * (hi / lo) ** (g / (count - 1)))


def run(N, g):
    K, kappa = STEPS[N], (0.05 if g == "c" else kappa_of(g))
    pde = o.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), stencil_size_interior=3, stencil_size_boundary=3,
                                t0=0.0, tmax=K * DT, diffusion_rate=kappa, kernel=o.SquareExponential(),
                                nugget_gram_matrix_fd=0.0, bcond="dirichlet")
    s = o.WhiteNoiseEK1(num_derivatives=2, steprule=o.Constant(DT), spatial_kernel=o.Matern52() + o.WhiteNoise(),
                        canonical_factor_signs=True)
    ts, means, stds, sig = [], [], [], []
    t0 = time.perf_counter()
    for state, _ in s.solution_generator(pde):      # streamed: the factors of all steps would be 1.9 GB at N=512
        C = state.y.cov_sqrtm
        ts.append(state.t)
        means.append(state.y.mean[0].copy())
        stds.append(np.sqrt(np.einsum("ij,ij->i", C, C) @ s.E0.T))
        if not isinstance(state.diffusion_squared_local, list):
            sig.append(state.diffusion_squared_local)
    assert len(ts) == K + 1 and ts[-1] == K * DT
    out = pathlib.Path(__file__).parent / f"oracle_heat_n{N}_nu2_k{g}.npz"
    np.savez_compressed(out, t=np.array(ts), means=np.array(means), stds=np.array(stds), sigma2=np.array(sig),
                        kappa=kappa, config=np.array([N, 2, DT, K]))
    print(f"{out.name}: {K} steps in {time.perf_counter() - t0:.0f} s, max|mean|={np.abs(means[-1]).max():.4g}, "
          f"max std={stds[-1].max():.4g}, sigma2[-1]={sig[-1]:.6g}", flush=True)


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for g in ([(a if a == "c" else int(a)) for a in sys.argv[2:]] or PROBLEMS[N]):
        run(N, g)
