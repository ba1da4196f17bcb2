"""Vendor-library bar ("CPU baseline beside it" (ii)): the same covariance-form filter step
written with torch-ROCm library calls (rocBLAS / rocSOLVER behind torch.mm, torch.linalg.cholesky,
torch.linalg.solve_triangular) on the GPU.  NOT part of the product and not used by it: a measurement tool that
answers "what does the step cost if one simply calls the libraries", next to bench.py's hand-written path.

The step uses the structure a library user gets for free (Kronecker predict via reshape, H as a dense (m, D) matrix
in one GEMM each way); fp64, same workload as bench.py.  Prints one JSON line.
"""

import argparse
import json
import pathlib
import sys

import numpy as np
import scipy.linalg

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "pnmol-experiments_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh-n", type=int, default=512)
    ap.add_argument("--nu", type=int, default=2)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import pnmol
    N, nu, dt = args.mesh_n, args.nu, 2.0 ** -7
    n = nu + 1
    pde = pnmol.pde.examples.heat_1d_discretized(bbox=[0.0, 1.0], dx=1.0 / (N - 1), tmax=1.0, diffusion_rate=0.05,
                                                 kernel=pnmol.kernels.SquareExponential(), bcond="dirichlet")
    solver = pnmol.white.LinearWhiteNoiseEK1(num_derivatives=nu, steprule=pnmol.odetools.step.Constant(dt),
                                             spatial_kernel=pnmol.kernels.Matern52() + pnmol.kernels.WhiteNoise())
    solver.iwp, solver.E0, solver.E1, gamma = solver.initialize_iwp(pde)
    solver._gram = gamma @ gamma.T
    mean, blocks = solver._initial_moments(pde)
    d, nB = N, pde.B.shape[0]
    D, m = n * d, d + nB
    s, _ = solver.iwp.nordsieck_preconditioner_1d_raw(dt)
    # derivative-major, preconditioned coordinates
    P0 = np.zeros((D, D))
    for (a, b), blk in blocks.items():
        P0[a * d:(a + 1) * d, b * d:(b + 1) * d] = blk / (s[a] * s[b])
    m0 = (mean / s[:, None]).reshape(-1)
    A1 = np.flip(scipy.linalg.pascal(n, kind="lower"))
    Q1 = np.flip(scipy.linalg.hilbert(n))
    H = np.zeros((m, D))
    H[:d, d:2 * d] = s[1] * np.eye(d)
    H[:d, :d] = -s[0] * pde.L
    H[d:, :d] = s[0] * pde.B
    Ebc = scipy.linalg.block_diag(pde.E_sqrtm, pde.R_sqrtm)
    dev = torch.device("cuda")
    t = lambda x: torch.tensor(x, dtype=torch.float64, device=dev)
    P, mu, A1t, Q1t, Ht, R, K = t(P0), t(m0), t(A1.copy()), t(Q1.copy()), t(H), t(Ebc @ Ebc.T), t(solver._gram)

    def step(P, mu):
        Pb = P.view(n, d, n, d)
        Pm = torch.einsum("ac,cjek,be->ajbk", A1t, Pb, A1t) + Q1t[:, None, :, None] * K[None, :, None, :]
        Pm = Pm.reshape(D, D)
        mp = (A1t @ mu.view(n, d)).reshape(D)
        PHt = Pm @ Ht.T
        S = Ht @ PHt + R
        z = Ht @ mp
        Ls = torch.linalg.cholesky(S)
        Wt = torch.linalg.solve_triangular(Ls, PHt.T, upper=False)
        r = torch.linalg.solve_triangular(Ls, z[:, None], upper=False)[:, 0]
        return Pm - Wt.T @ Wt, mp - Wt.T @ r, (r @ r) / m

    for _ in range(5):
        P, mu, sig = step(P, mu)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        P, mu, sig = step(P, mu)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    print(json.dumps({"tool": "torch_library_step", "mesh_n": N, "nu": nu, "ms_per_step": ms, "steps_per_s": 1e3 / ms,
                      "finite": bool(torch.isfinite(mu).all().item()), "sigma2": float(sig),
                      "mean0_max": float(mu[:d].abs().max() * s[0])}))


if __name__ == "__main__":
    main()
