"""CPU model of the fp32-covariance mode (pnmol_filter_desc.dtype = 1): where does its std floor come from, and does an
fp64 copy of diag(P) beside the fp32 matrix remove it?

The model runs the covariance-form step of the oracle in fp64 and rounds exactly what the device stores in fp32: the
posterior covariance P, the predicted covariance P- (formed in fp64 from the fp32 P, stored in fp32) and the down-date
P = P- - W W^T accumulated in fp32 (W itself is fp64, as on the device: S, its factorisation and W = P- H^T Ls^-T are fp64).
Beside it, it carries an fp64 "shadow" of the n x n point-diagonal blocks -- what a side channel can carry:
    shadow- = A1 shadow A1^T + Q1 K_jj,   shadow = shadow- - W_j W_j^T   (W from the fp32-rounded P-)
and reads the stds from the shadow instead of from the fp32 matrix.

Result (python tools/fp32_floor_model.py, log in profiles/r03_fp32_floor_model.log): the model reproduces the device's floor
(12x12: 1.3e-3 of max(std), the figure of DESIGN.md section 11), and the shadow makes it WORSE (1.7e-2): the fp32 matrix is
self-consistent -- its diagonal belongs to the gain computed from it --, the shadow is the Riccati recursion driven by a gain
that is not its own, and without the Joseph form that is not even a covariance update.  The floor comes from H applied to
fp32-rounded rows of P- (a second difference: rounding errors of 6e-8 |P-| amplified by 4 / dx^2), not from the diagonal.
"""
import pathlib
import sys

import numpy as np
import scipy.linalg

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import pnmol_oracle as o  # noqa: E402


def _f32(x):
    return x.astype(np.float32).astype(np.float64)


def model(n_mesh, K, dt=2.0 ** -8, kappa=0.05):
    """2-d Dirichlet heat problem of BASELINE config 5 (nu = 1) on an n_mesh x n_mesh grid, K steps.
    Returns stds (K, d) of: the fp64 covariance form, the fp32-rounded matrix, the fp64 shadow of its diagonal blocks."""
    pde = o.heat_2d_dirichlet_discretized(nums=(n_mesh, n_mesh), tmax=K * dt, diffusion_rate=kappa, kernel=o.SquareExponential())
    sol = o.WhiteNoiseEK1(num_derivatives=1, steprule=o.Constant(dt), canonical_factor_signs=True,
                          spatial_kernel=o.Matern52() + o.WhiteNoise())
    st = sol.initialize(pde)
    n, d = 2, pde.y0.shape[0]
    cov0 = st.y.cov_sqrtm @ st.y.cov_sqrtm.T
    P, Pinv = sol.iwp.nordsieck_preconditioner(dt)
    A, Ql = sol.iwp.preconditioned_discretize
    Q = Ql @ Ql.T
    A1 = A[:n, :n]

    def run(fp32):
        m = Pinv @ st.y.mean.reshape(-1, order="F")
        C = Pinv @ cov0 @ Pinv.T
        if fp32:
            C = _f32(C)
        blk = np.array([C[j * n:(j + 1) * n, j * n:(j + 1) * n] for j in range(d)])
        out_mat, out_sh, means = [], [], []
        for k in range(K):
            mp = A @ m
            Pm = A @ C @ A.T + Q
            if fp32:
                Pm = _f32(Pm)
            blkm = np.array([A1 @ blk[j] @ A1.T + Q[j * n:(j + 1) * n, j * n:(j + 1) * n] for j in range(d)])
            z, H, E = sol.evaluate_ode(pde, sol.E0 @ P, sol.E1 @ P, mp, (k + 1) * dt)
            Ls = np.linalg.cholesky(H @ Pm @ H.T + E @ E.T)
            Wt = scipy.linalg.solve_triangular(Ls, H @ Pm, lower=True)
            m = mp - Wt.T @ scipy.linalg.solve_triangular(Ls, z, lower=True)
            if fp32:
                W32 = Wt.T.astype(np.float32)
                C = (Pm.astype(np.float32) - W32 @ W32.T).astype(np.float64)
            else:
                C = Pm - Wt.T @ Wt
            W = Wt.T
            blk = np.array([blkm[j] - W[j * n:(j + 1) * n] @ W[j * n:(j + 1) * n].T for j in range(d)])
            s0 = P[0, 0]
            means.append(s0 * m[0::n])
            out_mat.append(s0 * np.sqrt(np.maximum(np.diag(C)[0::n], 0.0)))
            out_sh.append(s0 * np.sqrt(np.maximum(blk[:, 0, 0], 0.0)))
        return np.array(means), np.array(out_mat), np.array(out_sh)

    m64, s64, _ = run(False)
    m32, s32, s32sh = run(True)
    mx = s64.max()
    big = s64 >= 1e-2 * mx
    return {
        "mesh": n_mesh, "steps": K,
        "mean_rel": float(np.abs(m32 - m64).max() / np.abs(m64).max()),
        "floor_fp32_matrix": float(np.abs(s32 - s64).max() / mx),
        "rel_on_significant_fp32_matrix": float((np.abs(s32 - s64)[big] / s64[big]).max()),
        "floor_fp64_shadow": float(np.abs(s32sh - s64).max() / mx),
        "rel_on_significant_fp64_shadow": float((np.abs(s32sh - s64)[big] / s64[big]).max()),
    }


if __name__ == "__main__":
    import json
    for n_mesh, K in ((12, 12), (20, 8), (28, 8)):
        print(json.dumps(model(n_mesh, K)), flush=True)
