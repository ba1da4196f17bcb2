"""CPU model of an fp32 build of the square-root (QR) form: is it the route to fp32 at 1-d, nu = 2, where the fp32
covariance form diverges (finding 3; VERDICT round 2, missing #6)?

The model runs the oracle's as-written step (white.py:96-146: two QRs per step) and performs exactly what an fp32 device
QR would: the pre-arrays are rounded to fp32, the Householder QR runs in fp32 (LAPACK sgeqrf), the factor `Cl` is kept in
fp32.  Two variants of the mean path:
  * "all32": gain K = (R1^-1 R2)^T from the fp32 factor, m = m- - K z in fp64 arithmetic on fp32 data;
  * "S64":   the same, but z, m-, and the products with H stay fp64 (what the device would do: vectors are cheap).
Output: relative errors of mean / std against the fp64 oracle per step count, for N = 64, 128, 256, nu = 2, dt = 2^-7.

    python tools/fp32_sqrt_model.py            (log: profiles/r03_fp32_sqrt_model.log)
"""
import pathlib
import sys

import numpy as np
import scipy.linalg

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import pnmol_oracle as o  # noqa: E402


def qr_r(M, fp32):
    if fp32:
        M = M.astype(np.float32)
    R = scipy.linalg.qr(M, mode="r", pivoting=False, check_finite=False)[0][: M.shape[1]]
    return R.astype(np.float64)


def update_sqrt(H, C, E, fp32):
    m, D = H.shape
    bl = np.zeros((m, D))
    bl[:, :m] = E
    big = qr_r(np.block([[C.T @ H.T, C.T], [bl.T, np.zeros((D, D))]]), fp32)
    R1, R2, R3 = big[:m, :m], big[:m, m:], big[m:m + D, m:m + D]
    return R3.T, R1, R2


def run(N, nu, K, fp32, dt=2.0 ** -7, kappa=0.05):
    kw = dict(tmax=K * dt, dx=1.0 / (N - 1), diffusion_rate=kappa, bcond="dirichlet", stencil_size_interior=3,
              stencil_size_boundary=3, nugget_gram_matrix_fd=0.0)
    pde = o.heat_1d_discretized(kernel=o.SquareExponential(), **kw)
    sol = o.WhiteNoiseEK1(num_derivatives=nu, steprule=o.Constant(dt), canonical_factor_signs=True,
                          spatial_kernel=o.Matern52() + o.WhiteNoise())
    st = sol.initialize(pde)
    n, d = nu + 1, pde.y0.shape[0]
    P, Pinv = sol.iwp.nordsieck_preconditioner(dt)
    A, Ql = sol.iwp.preconditioned_discretize
    m = Pinv @ st.y.mean.reshape(-1, order="F")
    Cl = Pinv @ st.y.cov_sqrtm
    means, stds = [], []
    for k in range(K):
        mp = A @ m
        z, H, E = sol.evaluate_ode(pde, sol.E0 @ P, sol.E1 @ P, mp, (k + 1) * dt)
        Clp = qr_r(np.vstack(((A @ Cl).T, Ql.T)), fp32).T
        Cl, R1, R2 = update_sqrt(H, Clp, E, fp32)
        y = scipy.linalg.solve_triangular(R1, z, lower=False, trans="T")
        m = mp - R2.T @ y
        if fp32:
            Cl = Cl.astype(np.float32).astype(np.float64)
        mean = (P @ m).reshape((n, d), order="F")
        C = P @ Cl
        means.append(mean[0]), stds.append(np.sqrt(np.einsum("ij,ij->i", C, C)).reshape((n, d), order="F")[0])
    return np.array(means), np.array(stds)


def main():
    for N, K in ((64, 40), (128, 40), (256, 24)):
        m64, s64 = run(N, 2, K, False)
        m32, s32 = run(N, 2, K, True)
        for k in sorted({0, 3, 7, K // 2, K - 1}):
            em = np.abs(m32[k] - m64[k]).max() / np.abs(m64[k]).max()
            big = s64[k] >= 1e-2 * s64[k].max()
            es = (np.abs(s32[k] - s64[k])[big] / s64[k][big]).max()
            fl = np.abs(s32[k] - s64[k]).max() / s64[k].max()
            print(f"N={N:4d} nu=2 step {k + 1:3d}: mean rel {em:9.2e}   std rel (significant) {es:9.2e}   std floor/max {fl:9.2e}",
                  flush=True)


if __name__ == "__main__":
    main()
