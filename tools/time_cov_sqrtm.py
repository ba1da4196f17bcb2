"""Timing / accuracy of pnmol_state_get_cov_sqrtm (device Cholesky of the covariance) vs a host eigen factor."""
import sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd")); sys.path.insert(0, str(ROOT))
import numpy as np, bench
from pnmol.base import rv
for N in (64, 256, 512):
    bench.MESH_N = N
    pde, solver = bench.build_problem(0.05, 10)
    state = solver.initialize(pde)
    flt, dev = solver._device_filter, state.y.device_state
    solver._ensure_error_model(pde, bench.DT)
    flt.steps(dev, 10, bench.DT)
    dev.cov_sqrtm()
    t0 = time.perf_counter(); C = dev.cov_sqrtm(); t1 = time.perf_counter()
    cov = dev.cov(); t2 = time.perf_counter()
    Ch = rv.factor_of(cov); t3 = time.perf_counter()
    E = C @ C.T - cov
    lam = np.linalg.eigvalsh(0.5 * (cov + cov.T))
    i, j = np.unravel_index(np.abs(E).argmax(), E.shape)
    print(f"N={N}: device factor {1e3*(t1-t0):.1f} ms, host eigen factor {1e3*(t3-t2):.0f} ms; max|CC^T-cov| = {np.abs(E).max():.2e} at {(i, j)} "
          f"(cov there {cov[i, j]:.2e}, max cov {np.abs(cov).max():.2e}); eig min/max {lam[0]:.2e}/{lam[-1]:.2e}; zero columns {int((np.diag(C) == 0).sum())}")
