"""Bitwise repeatability of the step loop (a race or a stale-cache read shows up as run-to-run differences) and the
distance to the oracle for one small case."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
for p in (ROOT / "pnmol-experiments_amd", ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np
from helpers import make_pair
import pnmol_oracle as oracle

for N, K in [(64, 100), (96, 60), (256, 40), (512, 30)]:
    pde, solver, opde, osolver = make_pair(N, 2, 2.0 ** -7, K)
    runs = [solver.solve_marginals(pde) for _ in range(4)]
    same = [np.array_equal(runs[0][1], r[1]) and np.array_equal(runs[0][2], r[2]) and np.array_equal(runs[0][3], r[3]) for r in runs[1:]]
    dm = max(np.abs(runs[0][1] - r[1]).max() for r in runs[1:])
    ds = max(np.abs(runs[0][2] - r[2]).max() for r in runs[1:])
    msg = f"N={N} K={K}: repeat runs bitwise equal {same}  max|dmean|={dm:.3e} max|dstd|={ds:.3e}"
    if N <= 96:
        osol = osolver.solve(opde)
        om, os_ = oracle.read_mean_and_std(osol, osolver.E0)
        msg += f"  vs oracle: mean {np.abs(runs[0][1]-om).max()/np.abs(om).max():.2e} std {np.abs(runs[0][2]-os_).max()/os_.max():.2e} (rel. to max)"
    print(msg, flush=True)

# the 2-d mesh of tests/test_gpu_parity.py::test_two_dimensional_mesh: many noise-free boundary rows (exact zero pivots)
import pnmol
dt, K = 2.0 ** -8, 12
pde = pnmol.pde.examples.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=pnmol.kernels.SquareExponential())
opde = oracle.heat_2d_dirichlet_discretized(nums=(12, 12), tmax=K * dt, kernel=oracle.SquareExponential())
solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=1, steprule=pnmol.odetools.step.Constant(dt),
                                         spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
osolver = oracle.WhiteNoiseEK1(num_derivatives=1, steprule=oracle.Constant(dt), canonical_factor_signs=True,
                               spatial_kernel=oracle.Matern52() + oracle.WhiteNoise())
t, means, stds, sig, _ = solver.solve_marginals(pde)
osol = osolver.solve(opde)
om, os_ = oracle.read_mean_and_std(osol, osolver.E0)
err = np.abs(stds - os_)
print(f"2-d 12x12: max|std err| per step / max std: " + " ".join(f"{e.max() / os_.max():.1e}" for e in err))
print(f"   mean err / max: {np.abs(means - om).max() / np.abs(om).max():.2e}; worst entries: exact {os_.ravel()[np.argsort(err.ravel())[-3:]]} got {stds.ravel()[np.argsort(err.ravel())[-3:]]}")
