"""Probabilistic finite differences (reference: src/pnmol/discretize.py).

Batched NumPy: all stencils of a mesh are solved with one `np.linalg.solve` over (N, s, s).
`collocation_global` (figure 2 only) is outside the hot-path scope.
"""

import os

import numpy as np
import scipy.linalg

from . import diffops, kernels


def fd_coefficients(x, neighbors, k, L_k, LL_k, nugget_gram_matrix=0.0):
    """Kernel FD weights and uncertainty of one stencil (discretize.py:177-201).

    weights = (k(X,X) + eta I)^-1 L_k(x, X);   uncertainty = LL_k(x,x) - weights . L_k(x, X).
    """
    w, u = _fd_batched(np.asarray(x)[None], np.asarray(neighbors)[None], k, L_k, LL_k, nugget_gram_matrix)
    return w[0], u[0]


def _fd_batched(xs, nbrs, k, L_k, LL_k, nugget):
    s = nbrs.shape[1]
    gram = k._eval(nbrs[:, :, None, :], nbrs[:, None, :, :]) + nugget * np.eye(s)
    dk = L_k._eval(xs[:, None, :], nbrs)
    if os.environ.get("PNMOL_FD_ON_DEVICE") == "1" and s <= 16:
        # the s x s stencil systems on the device (`pnmol_fd_solve_batched`).  Opt-in: with a smooth kernel and a
        # fine mesh these systems have condition numbers of 1e10 and more, and two LU codes then agree only to cond * eps --
        # the default keeps LAPACK, whose weights are the ones the oracle fixtures were made with.
        from . import _hip
        return _hip.Context.default().fd_solve_batched(gram, dk, LL_k._eval(xs, xs))
    w = np.linalg.solve(gram, dk[..., None])[..., 0]
    unc = LL_k._eval(xs, xs) - np.einsum("ns,ns->n", w, dk)
    return w, unc


def fd_probabilistic(diffop, mesh_spatial, kernel=None, stencil_size_interior=3, stencil_size_boundary=3,
                     nugget_gram_matrix=0.0):
    """Discretise `diffop` on a mesh: (L, E_sqrtm), both (N, N)  (discretize.py:12-113).

    Quirk kept from the reference: the FD *variance* lands on the diagonal of `E_sqrtm`
    unsquared (discretize.py:110-112, :199).
    """
    if kernel is None:
        kernel = kernels.SquareExponential(input_scale=1.0, output_scale=1.0)
    L_kx = kernels.Lambda(diffop(kernel.pairwise, argnums=0), parent=kernel, spec=((diffop.name, 0),))
    LL_kx = kernels.Lambda(diffop(L_kx.pairwise, argnums=1))
    N = mesh_spatial.shape[0]
    L, E_sqrtm = np.zeros((N, N)), np.zeros((N, N))
    for (pts, _, idx), num in ((mesh_spatial.boundary, stencil_size_boundary),
                               (mesh_spatial.interior, stencil_size_interior)):
        if len(idx) == 0:
            continue
        nbrs, nbr_idx = mesh_spatial.neighbours(point=pts, num=num)
        w, unc = _fd_batched(pts, nbrs, kernel, L_kx, LL_kx, nugget_gram_matrix)
        L[idx[:, None], nbr_idx] = w
        E_sqrtm[idx, idx] = unc
    return L, E_sqrtm


def fd_probabilistic_neumann_1d(mesh_spatial, kernel=None, stencil_size=2, nugget_gram_matrix=0.0):
    """Two one-sided normal-derivative rows and their uncertainty (discretize.py:116-158)."""
    if stencil_size != 2:
        raise NotImplementedError
    if kernel is None:
        kernel = kernels.SquareExponential(input_scale=1.0, output_scale=1.0)
    D = diffops.gradient()
    Lk = kernels.Lambda(D(kernel.pairwise, argnums=0), parent=kernel, spec=((D.name, 0),))
    LLk = kernels.Lambda(D(Lk.pairwise, argnums=1))
    pts, N = mesh_spatial.points, len(mesh_spatial)
    wl, ul = fd_coefficients(pts[0], pts[[0, 1]], kernel, Lk, LLk, nugget_gram_matrix)
    wr, ur = fd_coefficients(pts[-1], pts[[-1, -2]], kernel, Lk, LLk, nugget_gram_matrix)
    B = np.eye(N)[[0, 1, N - 1, N - 2]]
    # the left normal derivative points to the left (discretize.py:151-153)
    diffmatrix = scipy.linalg.block_diag(-wl[None, :], wr[None, :])
    return diffmatrix @ B, np.diag(np.array([ul, ur]))
