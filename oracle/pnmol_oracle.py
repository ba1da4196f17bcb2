"""CPU oracle for the PNMOL white-noise EK1 hot path.  TEST INFRASTRUCTURE ONLY.

This file is a NumPy/SciPy (fp64) restatement of the algorithm of the reference
`schmidtjonathan/pnmol-experiments` for the path named in BASELINE.json's
`north_star`.  Every function cites the reference
`file:line` it follows (paths relative to the reference root, `src/pnmol/...`).

Who may use it: `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg -- as the checker / reported baseline.  Nothing under `pnmol-experiments_amd/`
imports it; the product path calls the HIP library and fails loudly without it.

Pinning status (read before trusting):
  * The reference is pure Python on JAX.  JAX / jaxlib / tornadox are not installed in
    the build container and cannot be fetched (`import pnmol` raises an ordinary
    `ModuleNotFoundError: jax`), so no output of the reference itself could be captured.
  * The oracle is pinned by every known-answer / identity test the reference's own
    test-suite holds for this path (tests/test_oracle_pinning.py restates them):
    IBM closed forms (tests/test_base/test_iwp.py:19-62), preconditioner identities
    (:50-93), sqrt == classic Kalman (tests/test_base/test_sqrt.py:36-109), FD weights
    [-2,1,1]/dx^2 with zero uncertainty (tests/test_discretize.py:52-71), E_sqrtm
    diagonal (:98-101), heat IVP Jacobian rows (tests/test_problems.py:103-163), mesh
    facts (tests/test_mesh.py:22-97), step-rule formulas
    (tests/test_odetools/test_step.py:15-122), `solve()` finite on the N=6 heat smoke
    problem incl. the runt 11th step (tests/test_pdefilter.py:15-64,141-146).
  * End-to-end `solve()` numbers are NOT pinned by any fixture of the reference (it has
    none); for those the oracle is "parity unpinned" beyond the identities above, plus
    the independent cross-checks in tests/ (square-root form == covariance form;
    filter mean vs a fine `solve_ivp` of y' = L y as experiments/figure1.py:57-73 does).

Deliberate differences from the reference (all numerically neutral):
  * jax autodiff of kernels (diffops.py:167-202 through discretize.py:51-52) is replaced
    by closed-form derivatives of the SE / Polynomial / Matern-5/2 kernels.
  * jnp -> numpy; `jnp.linalg.qr(mode="r")` -> `scipy.linalg.qr(mode="r")`.
"""

from __future__ import annotations

import math
from collections import namedtuple
from dataclasses import dataclass
from typing import Dict

import numpy as np
import scipy.linalg
import scipy.spatial
import scipy.special

# --------------------------------------------------------------------------------------
# Covariance kernels (reference: kernels.py:16-157)
# --------------------------------------------------------------------------------------


class _Kern:
    """Gram-matrix call conventions of kernels.py:16-47.

    k(x, y) with 1-d x, y -> scalar;  k(X, Y) with equal shapes (N, dim) -> diagonal (N,);
    k(X, Y) with X (N, dim), Y (dim, K) -> full Gram (N, K).
    """

    def pair(self, X, Y):  # X (..., dim), Y (..., dim) broadcast -> (...)
        raise NotImplementedError

    def __call__(self, X, Y):
        X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
        if X.ndim == Y.ndim <= 1:
            return self.pair(X, Y)
        if X.shape == Y.shape:
            return self.pair(X, Y)
        return self.pair(X[:, None, :], Y.T[None, :, :])

    def __add__(self, other):  # kernels.py:50-55
        return _Sum(self, other)


class _Sum(_Kern):
    def __init__(self, a, b):
        self.a, self.b = a, b

    def pair(self, X, Y):
        return self.a.pair(X, Y) + self.b.pair(X, Y)


class SquareExponential(_Kern):
    """k = s^2 exp(-l^2 |x-y|^2 / 2); note l MULTIPLIES the distance (kernels.py:107-111)."""

    def __init__(self, *, output_scale=1.0, input_scale=1.0):
        self.output_scale, self.input_scale = output_scale, input_scale

    def pair(self, X, Y):
        r2 = np.sum((X - Y) ** 2, axis=-1) * self.input_scale**2
        return self.output_scale**2 * np.exp(-r2 / 2.0)

    # closed forms standing in for diffops.laplace()/gradient() applied by jax autodiff
    def laplace_x(self, X, Y):
        c, n = self.input_scale**2, X.shape[-1]
        r2 = np.sum((X - Y) ** 2, axis=-1)
        return (c * c * r2 - c * n) * self.pair(X, Y)

    def laplace_xy(self, X, Y):
        c, n = self.input_scale**2, X.shape[-1]
        r2 = np.sum((X - Y) ** 2, axis=-1)
        phi = c * c * r2 - c * n
        return (phi * phi - 4.0 * c**3 * r2 + 2.0 * c * c * n) * self.pair(X, Y)

    def grad_x_1d(self, X, Y):
        c, u = self.input_scale**2, (X - Y)[..., 0]
        return -c * u * self.pair(X, Y)

    def grad_xy_1d(self, X, Y):
        c, u = self.input_scale**2, (X - Y)[..., 0]
        return (c - c * c * u * u) * self.pair(X, Y)


class Matern52(_Kern):
    """kernels.py:114-124.  Derivatives only in 1-d (the reference patches the autodiff
    NaN at x == y with the Maclaurin values, discretize.py:184-197; the closed forms
    below take exactly those values at r = 0)."""

    def __init__(self, *, output_scale=1.0, input_scale=1.0):
        self.output_scale, self.input_scale = output_scale, input_scale

    def pair(self, X, Y):
        r = np.sqrt(5.0 * np.sum((X - Y) ** 2, axis=-1) * self.input_scale**2)
        return self.output_scale**2 * (1.0 + r + r * r / 3.0) * np.exp(-r)

    def _ar(self, X, Y):
        if X.shape[-1] != 1:
            raise NotImplementedError("Matern52 derivatives: 1-d only")
        a = math.sqrt(5.0) * self.input_scale
        return a, np.abs((X - Y)[..., 0])

    def laplace_x(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale**2 * (a * a / 3.0) * (a * a * r * r - a * r - 1.0) * np.exp(-a * r)

    def laplace_xy(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale**2 * (a**4 / 3.0) * (3.0 - 5.0 * a * r + a * a * r * r) * np.exp(-a * r)

    def grad_x_1d(self, X, Y):
        a, r = self._ar(X, Y)
        u = (X - Y)[..., 0]
        return -self.output_scale**2 * (a * a / 3.0) * u * (1.0 + a * r) * np.exp(-a * r)

    def grad_xy_1d(self, X, Y):
        a, r = self._ar(X, Y)
        return self.output_scale**2 * (a * a / 3.0) * (1.0 + a * r - a * a * r * r) * np.exp(-a * r)


class Polynomial(_Kern):
    """k = (x.y + c)^p  (kernels.py:127-144)."""

    def __init__(self, *, order=2, const=1.0):
        self.order, self.const = order, const

    def pair(self, X, Y):
        return (np.sum(X * Y, axis=-1) + self.const) ** self.order

    @staticmethod
    def _pw(base, e):
        return np.ones_like(base) if e <= 0 else base**e

    def laplace_x(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * (p - 1) * self._pw(s, p - 2) * np.sum(Y * Y, axis=-1)

    def laplace_xy(self, X, Y):
        p, n = self.order, X.shape[-1]
        xy = np.sum(X * Y, axis=-1)
        s = xy + self.const
        x2, y2 = np.sum(X * X, axis=-1), np.sum(Y * Y, axis=-1)
        t = (p - 2) * (p - 3) * self._pw(s, p - 4) * x2 * y2
        t = t + 4.0 * (p - 2) * self._pw(s, p - 3) * xy + 2.0 * n * self._pw(s, p - 2)
        return p * (p - 1) * t

    def grad_x_1d(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * self._pw(s, p - 1) * Y[..., 0]

    def grad_xy_1d(self, X, Y):
        p, s = self.order, np.sum(X * Y, axis=-1) + self.const
        return p * self._pw(s, p - 1) + p * (p - 1) * self._pw(s, p - 2) * X[..., 0] * Y[..., 0]


class WhiteNoise(_Kern):
    """k = s^2 [x == y]  (kernels.py:147-157)."""

    def __init__(self, *, output_scale=1.0):
        self.output_scale = output_scale

    def pair(self, X, Y):
        return self.output_scale**2 * np.all(X == Y, axis=-1)


class _Stacked(_Kern):
    """kernels.py:160-176: stack of kernels whose full Gram matrix is block diagonal (diagonal: concatenated)."""

    def __init__(self, kernel_list):
        self.kernel_list = list(kernel_list)

    def __call__(self, X, Y):
        grams = [k(X, Y) for k in self.kernel_list]
        if np.shape(X) == np.shape(Y):
            return np.concatenate(grams)
        return scipy.linalg.block_diag(*grams)


def duplicate(kernel, num):  # kernels.py:178-183
    return _Stacked([kernel] * num)


# --------------------------------------------------------------------------------------
# Mesh (reference: mesh.py:75-184)
# --------------------------------------------------------------------------------------


class RectMesh:
    def __init__(self, points):
        self.points = np.asarray(points, dtype=np.float64)
        # mesh.py:177-184 read_bbox: one (min, max) row per spatial dimension
        self.bbox = np.stack([self.points.min(axis=0), self.points.max(axis=0)], axis=1)
        self._tree = scipy.spatial.KDTree(self.points)

    @classmethod
    def from_bbox_1d(cls, bbox, step=None, num=None):  # mesh.py:85-98
        bbox = np.asarray(bbox, dtype=np.float64)
        if (step is None) == (num is None):
            raise ValueError("Provide exactly one of step or num.")
        if step is not None:
            num = int((bbox[1] - bbox[0]) / step) + 1
        return cls(np.linspace(bbox[0], bbox[1], num=num, endpoint=True).reshape(-1, 1))

    @classmethod
    def from_bbox_2d(cls, bbox, steps=None, nums=None):  # mesh.py:100-130
        bbox = np.asarray(bbox, dtype=np.float64)
        if (steps is None) == (nums is None):
            raise ValueError("Provide exactly one of step or num.")
        if steps is not None:
            num_y = int((bbox[1, 0] - bbox[0, 0]) / steps[0]) + 1
            num_x = int((bbox[1, 1] - bbox[0, 1]) / steps[1]) + 1
        else:
            num_y, num_x = nums
        Y = np.linspace(bbox[0, 0], bbox[1, 0], num=num_y, endpoint=True)
        X = np.linspace(bbox[0, 1], bbox[1, 1], num=num_x, endpoint=True)
        Xm, Ym = np.meshgrid(X, Y)
        return cls(np.stack([Xm.ravel(), Ym.ravel()], axis=1))

    def __len__(self):
        return len(self.points)

    def __getitem__(self, key):
        return self.points[key]

    @property
    def shape(self):
        return self.points.shape

    def neighbours(self, point, num):  # mesh.py:132-139
        _, idx = self._tree.query(x=point, k=num)
        return self.points[idx], idx

    def _boundary_mask(self):  # mesh.py:141-154
        m = np.zeros(len(self.points), dtype=bool)
        for k in range(self.points.shape[1]):
            m |= (self.points[:, k] == self.bbox[k, 0]) | (self.points[:, k] == self.bbox[k, 1])
        return m

    @property
    def boundary(self):
        m = self._boundary_mask()
        return self.points[m], m, np.nonzero(m)[0]

    @property
    def interior(self):  # mesh.py:156-169
        m = ~self._boundary_mask()
        return self.points[m], m, np.nonzero(m)[0]

    @property
    def boundary_projection_matrix(self):  # mesh.py:171-175
        return np.eye(len(self.points))[self._boundary_mask(), :]


# --------------------------------------------------------------------------------------
# Probabilistic finite differences (reference: discretize.py:12-201)
# --------------------------------------------------------------------------------------


def fd_coefficients(x, neighbors, k, Lk, LLk, nugget_gram_matrix=0.0):
    """discretize.py:177-201: w = (k(X,X)+eta I)^-1 Lk(x,X);  u = LLk(x,x) - w.Lk(x,X).

    `Lk(x_row, Xcols)` / `LLk(x, x)` are callables with the broadcasting `pair` signature.
    """
    X, n = neighbors, neighbors.shape[0]
    gram = k(X, X.T) + nugget_gram_matrix * np.eye(n)
    dk = Lk(x[None, :], X)
    w = np.linalg.solve(gram, dk)
    unc = LLk(x, x) - w @ dk
    return w, unc


def fd_probabilistic_laplace(mesh, kernel=None, stencil_size_interior=3, stencil_size_boundary=3,
                             nugget_gram_matrix=0.0):
    """discretize.py:12-113 with diffop = laplace().  Returns dense L (N,N), E_sqrtm (N,N).

    Quirk kept: the *variance* u is written on the diagonal of `E_sqrtm` unsquared
    (discretize.py:110-112, :199).
    """
    if kernel is None:
        kernel = SquareExponential()
    N = mesh.shape[0]
    L, E = np.zeros((N, N)), np.zeros((N, N))
    for pts, idx, num in ((mesh.boundary[0], mesh.boundary[2], stencil_size_boundary),
                          (mesh.interior[0], mesh.interior[2], stencil_size_interior)):
        if len(idx) == 0:
            continue
        nb_pts, nb_idx = mesh.neighbours(pts, num)
        for row, x, Xn, cols in zip(idx, pts, nb_pts, nb_idx):
            w, u = fd_coefficients(x, Xn, kernel, kernel.laplace_x, kernel.laplace_xy, nugget_gram_matrix)
            L[row, cols] = w
            E[row, row] = u
    return L, E


def fd_probabilistic_neumann_1d(mesh, kernel=None, nugget_gram_matrix=0.0):
    """discretize.py:116-174: 2-point one-sided normal derivative rows + their uncertainty."""
    if kernel is None:
        kernel = SquareExponential()
    pts, N = mesh.points, len(mesh)
    wl, ul = fd_coefficients(pts[0], pts[[0, 1]], kernel, kernel.grad_x_1d, kernel.grad_xy_1d, nugget_gram_matrix)
    wr, ur = fd_coefficients(pts[-1], pts[[-1, -2]], kernel, kernel.grad_x_1d, kernel.grad_xy_1d, nugget_gram_matrix)
    B = np.eye(N)[[0, 1, N - 1, N - 2]]
    diffmatrix = scipy.linalg.block_diag(-wl[None, :], wr[None, :])
    return diffmatrix @ B, np.diag([ul, ur])


# --------------------------------------------------------------------------------------
# PDE problems (reference: pde/examples.py:13-81,347-357; pde/mixins.py:19-59,259-284)
# --------------------------------------------------------------------------------------


class HeatProblem:
    """Duck-typed stand-in for LinearEvolutionDirichlet/Neumann (pde/problems.py:45-66)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def bc_remove_pad(self, x):  # mixins.py:259-284
        return x[1:-1]

    def bc_pad(self, x):
        mode = "edge" if self.bcond == "neumann" else "constant"
        return np.pad(x, 1, mode=mode)

    def ivp_rhs(self, _t, x):  # IVPConversionLinearMixIn.to_tornadox_ivp, mixins.py:177-193
        return self.bc_remove_pad(self.L @ self.bc_pad(x))


def default_heat_y0(x, bbox):  # examples.py:59-61, :347-357
    mid = 0.5 * (bbox[1] + bbox[0])
    return np.exp(-((x - mid) ** 2)) * 0.1 * np.sin(np.pi * x)


def heat_1d_discretized(*, bbox=None, dx=0.05, stencil_size_interior=3, stencil_size_boundary=3,
                        t0=0.0, tmax=5.0, y0_fun=None, diffusion_rate=0.05,
                        nugget_gram_matrix_fd=0.0, kernel=None, bcond="dirichlet"):
    """examples.py:13-81 + DiscretizationMixIn.discretize (mixins.py:19-59)."""
    bbox = np.asarray([0.0, 1.0] if bbox is None else bbox, dtype=np.float64)
    if y0_fun is None:
        y0_fun = lambda x: default_heat_y0(x, bbox)  # noqa: E731
    mesh = RectMesh.from_bbox_1d(bbox, step=dx)
    kernel = SquareExponential() if kernel is None else kernel
    L, E = fd_probabilistic_laplace(mesh, kernel, stencil_size_interior, stencil_size_boundary,
                                    nugget_gram_matrix_fd)
    if bcond == "neumann":
        B, R = fd_probabilistic_neumann_1d(mesh, kernel, nugget_gram_matrix_fd)
    elif bcond == "dirichlet":
        B = mesh.boundary_projection_matrix
        R = np.zeros((B.shape[0], B.shape[0]))
    else:
        raise ValueError
    return HeatProblem(L=diffusion_rate * L, E_sqrtm=diffusion_rate * E, B=B, R_sqrtm=R,
                       y0=y0_fun(mesh.points)[:, 0], t0=t0, tmax=tmax, mesh_spatial=mesh,
                       bbox=bbox, diffop_scale=diffusion_rate, bcond=bcond, f=None, df=None)


def spruce_budworm_1d_discretized(*, bbox=None, t0=0.0, tmax=10.0, diffusion_rate=1.0, dx=0.1, kernel=None,
                                  stencil_size_interior=3, stencil_size_boundary=3, bcond="dirichlet", growth_rate=1.0,
                                  nugget_gram_matrix_fd=0.0):
    """examples.py:251-341: Fisher's equation u_t = kappa u_xx + c u (1 - u); y0 = 0.1 sin(pi x)."""
    p = heat_1d_discretized(bbox=bbox, dx=dx, stencil_size_interior=stencil_size_interior,
                            stencil_size_boundary=stencil_size_boundary, t0=t0, tmax=tmax,
                            y0_fun=lambda x: 0.1 * np.sin(np.pi * x), diffusion_rate=diffusion_rate, kernel=kernel,
                            bcond=bcond, nugget_gram_matrix_fd=nugget_gram_matrix_fd)
    p.f = lambda _t, x: growth_rate * x * (1.0 - x)
    p.df = lambda _t, x: np.diag(growth_rate * (1.0 - 2.0 * x))
    return p


def _system_discretized(*, f, df, y0_fun, diffusion_rates, bbox, t0, tmax, dx, kernel, stencil_size_interior,
                        stencil_size_boundary, nugget_gram_matrix_fd):
    """SystemDiscretizationMixIn.discretize_system (mixins.py:62-125) for SystemSemiLinearEvolutionNeumann
    (problems.py:76-86): block-diagonal L, E_sqrtm (one scaled Laplacian per component), B, R_sqrtm."""
    bbox = np.asarray([0.0, 1.0] if bbox is None else bbox, dtype=np.float64)
    mesh = RectMesh.from_bbox_1d(bbox, step=dx)
    kernel = kernel or SquareExponential()
    L, E = fd_probabilistic_laplace(mesh, kernel, stencil_size_interior, stencil_size_boundary, nugget_gram_matrix_fd)
    B, R = fd_probabilistic_neumann_1d(mesh, kernel, nugget_gram_matrix_fd)
    n = len(diffusion_rates)
    return HeatProblem(L=scipy.linalg.block_diag(*[r * L for r in diffusion_rates]),
                       E_sqrtm=scipy.linalg.block_diag(*[r * E for r in diffusion_rates]),
                       B=scipy.linalg.block_diag(*([B] * n)), R_sqrtm=scipy.linalg.block_diag(*([R] * n)),
                       y0=np.asarray(y0_fun(mesh.points)).squeeze(), t0=t0, tmax=tmax, mesh_spatial=mesh, bbox=bbox,
                       diffop_scale=tuple(diffusion_rates), bcond="neumann", f=f, df=df)


def lotka_volterra_1d_discretized(*, bbox=None, t0=0.0, tmax=10.0, a=0.5, b=0.05, c=0.05, d=0.5, diffusion_scale_u=0.1,
                                  diffusion_scale_v=0.1, dx=0.05, kernel=None, nugget_gram_matrix_fd=0.0,
                                  stencil_size_interior=3, stencil_size_boundary=3):
    """examples.py:181-248: u_t = D_u u_xx + a u - b u v,  v_t = D_v v_xx + c u v - d v, Neumann; state [u; v]."""

    def f(_t, x):
        u, v = np.split(np.asarray(x, dtype=np.float64), 2)
        return np.concatenate((a * u - b * u * v, c * u * v - d * v))

    def df(_t, x):  # jax.jacfwd of f (examples.py:232)
        u, v = np.split(np.asarray(x, dtype=np.float64), 2)
        return np.block([[np.diag(a - b * v), np.diag(-b * u)], [np.diag(c * v), np.diag(c * u - d)]])

    def y0_fun(x):
        return np.concatenate((5.0 * np.ones_like(x), 20.0 * np.exp(-(x ** 2))))

    return _system_discretized(f=f, df=df, y0_fun=y0_fun, diffusion_rates=(diffusion_scale_u, diffusion_scale_v),
                               bbox=bbox, t0=t0, tmax=tmax, dx=dx, kernel=kernel,
                               stencil_size_interior=stencil_size_interior, stencil_size_boundary=stencil_size_boundary,
                               nugget_gram_matrix_fd=nugget_gram_matrix_fd)


def sir_1d_discretized(*, bbox=None, dx=0.05, t0=0.0, tmax=50.0, beta=0.3, gamma=0.07, N=1000.0, diffusion_rate_S=0.1,
                       diffusion_rate_I=0.1, diffusion_rate_R=0.1, kernel=None, nugget_gram_matrix_fd=0.0,
                       stencil_size_interior=3, stencil_size_boundary=3):
    """examples.py:84-178: spatial SIR model, Neumann; state [S; I; R]."""
    bb = np.asarray([0.0, 1.0] if bbox is None else bbox, dtype=np.float64)

    def f(_t, x):
        s, i, r = np.split(np.asarray(x, dtype=np.float64), 3)
        tot = s + i + r
        return np.concatenate((-beta * s * i / tot, beta * s * i / tot - gamma * i, gamma * i))

    def df(_t, x):  # jax.jacfwd of f (examples.py:163)
        s, i, r = np.split(np.asarray(x, dtype=np.float64), 3)
        tot = s + i + r
        g = beta * s * i / tot                       # infection term and its partial derivatives
        gs, gi, gr = beta * i / tot - g / tot, beta * s / tot - g / tot, -g / tot
        D, Z = np.diag, np.zeros((s.size, s.size))
        return np.block([[D(-gs), D(-gi), D(-gr)], [D(gs), D(gi - gamma), D(gr)], [Z, D(gamma * np.ones_like(i)), Z]])

    def y0_fun(x):
        mid = 0.5 * (bb[1] + bb[0])
        inf = 200.0 * np.exp(-((x - mid) ** 2) / 0.5 ** 2) + 1.0
        return np.concatenate((N * np.ones_like(inf) - inf, inf, np.zeros_like(inf)))

    return _system_discretized(f=f, df=df, y0_fun=y0_fun,
                               diffusion_rates=(diffusion_rate_S, diffusion_rate_I, diffusion_rate_R), bbox=bbox, t0=t0,
                               tmax=tmax, dx=dx, kernel=kernel, stencil_size_interior=stencil_size_interior,
                               stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix_fd=nugget_gram_matrix_fd)


def heat_2d_dirichlet_discretized(*, nums=(8, 8), stencil_size_interior=5, stencil_size_boundary=5,
                                  t0=0.0, tmax=1.0, diffusion_rate=0.05, kernel=None):
    """Build-side construction of BASELINE config 5 from reference parts
    (mesh.py:100-130, mixins.py:51-54); the reference never builds a 2-d problem."""
    mesh = RectMesh.from_bbox_2d(np.array([[0.0, 0.0], [1.0, 1.0]]), nums=nums)
    kernel = SquareExponential() if kernel is None else kernel
    L, E = fd_probabilistic_laplace(mesh, kernel, stencil_size_interior, stencil_size_boundary)
    B = mesh.boundary_projection_matrix
    p = mesh.points
    y0 = 0.1 * np.sin(np.pi * p[:, 0]) * np.sin(np.pi * p[:, 1])
    return HeatProblem(L=diffusion_rate * L, E_sqrtm=diffusion_rate * E, B=B,
                       R_sqrtm=np.zeros((B.shape[0], B.shape[0])), y0=y0, t0=t0, tmax=tmax,
                       mesh_spatial=mesh, bbox=mesh.bbox, diffop_scale=diffusion_rate,
                       bcond="dirichlet", f=None, df=None)


# --------------------------------------------------------------------------------------
# Integrated Wiener process prior (reference: base/iwp.py:10-137)
# --------------------------------------------------------------------------------------


class IWP:
    def __init__(self, wiener_process_dimension, num_derivatives, wp_diffusion_sqrtm):
        self.d, self.nu, self.gamma = wiener_process_dimension, num_derivatives, np.asarray(wp_diffusion_sqrtm)
        self._cache = None

    @property
    def preconditioned_discretize_1d(self):  # iwp.py:13-30 (np.flip without axis flips BOTH axes)
        A1 = np.flip(scipy.linalg.pascal(self.nu + 1, kind="lower", exact=False))
        Q1 = np.flip(scipy.linalg.hilbert(self.nu + 1))
        return A1, np.linalg.cholesky(Q1)

    @property
    def preconditioned_discretize(self):  # iwp.py:32-53
        if self._cache is None:
            A1, LQ1 = self.preconditioned_discretize_1d
            self._cache = (np.kron(np.eye(self.d), A1), np.kron(self.gamma, LQ1))
        return self._cache

    def nordsieck_preconditioner_1d_raw(self, dt):  # iwp.py:55-62
        powers = np.arange(self.nu, -1, -1)
        scales = scipy.special.factorial(powers)
        powers = powers + 0.5
        return (np.abs(dt) ** powers) / scales, (np.abs(dt) ** (-powers)) * scales

    def nordsieck_preconditioner(self, dt):  # iwp.py:79-97
        s, sinv = self.nordsieck_preconditioner_1d_raw(dt)
        eye = np.eye(self.d)
        return np.kron(eye, np.diag(s)), np.kron(eye, np.diag(sinv))

    def non_preconditioned_discretize(self, dt):  # iwp.py:99-122
        P, Pinv = self.nordsieck_preconditioner(dt)
        A, Ql = self.preconditioned_discretize
        return P @ A @ Pinv, P @ Ql

    def projection_matrix(self, q):  # iwp.py:125-133
        return np.kron(np.eye(self.d), np.eye(1, self.nu + 1, q))


# --------------------------------------------------------------------------------------
# Square-root algebra (reference: base/sqrt.py:8-95)
# --------------------------------------------------------------------------------------


def _qr_r(M):
    R = scipy.linalg.qr(M, mode="r", pivoting=False, check_finite=False)[0]
    return R[: M.shape[1]]


def sqrtm_to_cholesky(St):  # sqrt.py:15-23
    return _qr_r(St).T


def propagate_cholesky_factor(S1, S2):  # sqrt.py:8-12
    return sqrtm_to_cholesky(np.vstack((S1.T, S2.T)))


def update_sqrt(H, C, meascov_sqrtm=None):
    """sqrt.py:33-73 (and :76-95 when `meascov_sqrtm is None`).  Returns (C_post, K, S_sqrtm)."""
    m, D = H.shape
    bottomleft = np.zeros((m, D))
    if meascov_sqrtm is not None:
        bottomleft[:, :m] = meascov_sqrtm
    block = np.block([[C.T @ H.T, C.T], [bottomleft.T, np.zeros((D, D))]])
    big = _qr_r(block)
    R1, R2, R3 = big[:m, :m], big[:m, m:], big[m:m + D, m:m + D]
    gain = scipy.linalg.solve_triangular(R1, R2, lower=False).T
    return R3.T, gain, R1.T


# --------------------------------------------------------------------------------------
# Step rules (reference: odetools/step.py:30-133)
# --------------------------------------------------------------------------------------


class Constant:
    def __init__(self, dt):
        self.dt = dt

    def suggest(self, previous_dt, scaled_error_estimate, local_convergence_rate=None):
        return self.dt

    def is_accepted(self, scaled_error_estimate):
        return True

    def scale_error_estimate(self, unscaled_error_estimate, reference_state):
        return None

    def first_dt(self, pde):
        return self.dt


class Adaptive:
    def __init__(self, abstol=1e-4, reltol=1e-2, max_changes=(0.2, 10.0), safety_scale=0.95):
        self.abstol, self.reltol, self.max_changes, self.safety_scale = abstol, reltol, max_changes, safety_scale

    def suggest(self, previous_dt, scaled_error_estimate, local_convergence_rate=None):  # step.py:80-91
        if local_convergence_rate is None:
            raise ValueError("Please provide a local convergence rate.")
        small, large = self.max_changes
        change = self.safety_scale * (1.0 / scaled_error_estimate) ** (1.0 / local_convergence_rate)
        return max(small, min(change, large)) * previous_dt

    def is_accepted(self, scaled_error_estimate):
        return scaled_error_estimate < 1

    def scale_error_estimate(self, unscaled_error_estimate, reference_state):  # step.py:96-107
        ratio = np.atleast_1d(unscaled_error_estimate / (self.abstol + self.reltol * reference_state))
        return np.linalg.norm(ratio) / np.sqrt(ratio.shape[0])

    def first_dt(self, pde):  # step.py:109-133
        if getattr(pde, "f", None) is None:
            return 0.01 * np.linalg.norm(pde.y0) / np.linalg.norm(pde.L @ pde.y0)
        return 0.01 * np.linalg.norm(pde.y0) / np.linalg.norm(pde.f(pde.t0, pde.y0))


# --------------------------------------------------------------------------------------
# White-noise EK1 (reference: white.py:11-208) and driver (pdefilter.py:17-256)
# --------------------------------------------------------------------------------------

MVN = namedtuple("MVN", "mean cov_sqrtm")
FilterState = namedtuple("FilterState", "t y error_estimate reference_state diffusion_squared_local")


@dataclass
class Solution:
    t: np.ndarray
    mean: np.ndarray
    cov_sqrtm: np.ndarray
    info: Dict
    diffusion_squared_calibrated: float


class _TimeStopper:  # pdefilter.py:238-256
    def __init__(self, locations):
        self._locations = iter(locations)
        self._next = next(self._locations)

    def adjust_dt_to_time_stops(self, t, dt):
        if t >= self._next:
            try:
                self._next = next(self._locations)
            except StopIteration:
                self._next = np.inf
        if t + dt > self._next:
            dt = self._next - t
        return dt


class WhiteNoiseEK1:
    """`LinearWhiteNoiseEK1` (semilinear=False) / `SemiLinearWhiteNoiseEK1` (True)."""

    def __init__(self, *, steprule=None, num_derivatives=2, spatial_kernel=None,
                 diffuse_prior_scale=1.0, semilinear=False, canonical_factor_signs=False):  # pdefilter.py:37-70
        # canonical_factor_signs: see `attempt_step` (quirk Q1).  False = as written.
        self.canonical_factor_signs = canonical_factor_signs
        self.steprule = steprule or Adaptive()
        self.num_derivatives = num_derivatives
        self.spatial_kernel = spatial_kernel or (Matern52() + WhiteNoise())
        self.diffuse_prior_scale = diffuse_prior_scale
        self.semilinear = semilinear
        self.iwp = self.E0 = self.E1 = None

    # ---- white.py:169-208
    def evaluate_ode(self, pde, p0, p1, m_pred, t):
        L, B = pde.L, pde.B
        m_at = p0 @ m_pred
        if self.semilinear:
            fx, Jx = pde.f(t, m_at), pde.df(t, m_at)
            H_ode = p1 - Jx @ p0 - L @ p0
        else:
            fx, Jx = L @ m_at, L
            H_ode = p1 - Jx @ p0
        b = Jx @ m_at - fx
        H = np.vstack((H_ode, B @ p0))
        z = H @ m_pred + np.hstack((b, np.zeros(B.shape[0])))
        return z, H, scipy.linalg.block_diag(pde.E_sqrtm, pde.R_sqrtm)

    # ---- white.py:82-94
    def initialize_iwp(self, pde):
        X = pde.mesh_spatial.points
        gamma = np.linalg.cholesky(self.spatial_kernel(X, X.T))
        prior = IWP(pde.y0.shape[0], self.num_derivatives, gamma)
        return prior, prior.projection_matrix(0), prior.projection_matrix(1), gamma

    # ---- white.py:12-80
    def initialize(self, pde):
        self.iwp, self.E0, self.E1, gamma = self.initialize_iwp(pde)
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        C0_raw = np.kron(gamma, self.diffuse_prior_scale * np.eye(n))
        C0_y0, k_y0, _ = update_sqrt(self.E0, C0_raw, 1e-10 * np.eye(d))
        m0_y0 = k_y0 @ pde.y0
        z, H, E = self.evaluate_ode(pde, self.E0, self.E1, m0_y0, pde.t0)
        C0, k, _ = update_sqrt(H, C0_y0, E + 1e-10 * np.eye(d + pde.B.shape[0]))
        m0 = m0_y0 - k @ z
        return FilterState(t=pde.t0, y=MVN(m0.reshape((n, d), order="F"), C0), error_estimate=None,
                           reference_state=None, diffusion_squared_local=[])

    # ---- white.py:153-162
    @staticmethod
    def estimate_error(ql, z, h, E_sqrtm):
        S = h @ (ql @ ql.T) @ h.T + E_sqrtm @ E_sqrtm.T
        sigma = np.sqrt(z @ np.linalg.solve(S, z) / z.shape[0])
        return sigma, np.sqrt(np.diag(S)) * sigma

    # ---- white.py:96-146
    def attempt_step(self, state, dt, pde):
        P, Pinv = self.iwp.nordsieck_preconditioner(dt)
        A, Ql = self.iwp.preconditioned_discretize
        n, d = self.num_derivatives + 1, pde.y0.shape[0]
        m = Pinv @ state.y.mean.reshape((-1,), order="F")
        Cl = Pinv @ state.y.cov_sqrtm
        mp = A @ m
        z, H, E = self.evaluate_ode(pde, self.E0 @ P, self.E1 @ P, mp, state.t + dt)
        _, error = self.estimate_error(Ql, z, H, E)
        Clp = propagate_cholesky_factor(A @ Cl, Ql)
        error = error[: -pde.B.shape[0]]
        Cl_new, K, Sl = update_sqrt(H, Clp, E)
        m_new = mp - K @ z
        # Quirk Q1 (white.py:125): the reference solves with Sl.T (= R1), i.e. it forms
        # |Sl^-T z|^2 = z^T (Sl^T Sl)^-1 z, not the whitened residual |Sl^-1 z|^2.  That value
        # depends on the SIGNS of diag(R1), which LAPACK's Householder QR chooses from the data
        # (about half are negative) -- it is not a function of (S, z) alone.  With
        # `canonical_factor_signs` the rows of R1 are flipped to a positive diagonal first
        # (the unique Cholesky factor of S); everything else in the step is sign-invariant.
        if self.canonical_factor_signs:
            Sl = Sl * np.sign(np.diag(Sl))[None, :]
        r = scipy.linalg.solve_triangular(Sl.T, z, lower=False)
        sigma2 = r @ r / r.shape[0]
        error = dt * error
        Cl_new, m_new = P @ Cl_new, (P @ m_new).reshape((n, d), order="F")
        new = FilterState(t=state.t + dt, y=MVN(m_new, Cl_new), error_estimate=error,
                          reference_state=np.abs(m_new[0]), diffusion_squared_local=sigma2)
        return new, dict(num_f_evaluations=1, num_df_evaluations=1)

    # ---- pdefilter.py:177-227
    def perform_full_step(self, state, initial_dt, pde):
        dt, ok, proposed = initial_dt, False, None
        info = dict(num_f_evaluations=0, num_df_evaluations=0, num_df_diagonal_evaluations=0,
                    num_attempted_steps=0)
        while not ok:
            proposed, ainfo = self.attempt_step(state, dt, pde)
            info["num_attempted_steps"] += 1
            for key in ("num_f_evaluations", "num_df_evaluations", "num_df_diagonal_evaluations"):
                info[key] += ainfo.get(key, 0)
            err = dt * proposed.error_estimate if proposed.error_estimate is not None else None
            norm = self.steprule.scale_error_estimate(unscaled_error_estimate=err,
                                                      reference_state=proposed.reference_state)
            ok = self.steprule.is_accepted(norm)
            suggested = self.steprule.suggest(dt, norm, local_convergence_rate=self.num_derivatives + 1)
            dt = min(suggested, pde.tmax - (proposed.t if ok else state.t))
            assert dt >= 0, f"Invalid step size: dt={dt}"
        return proposed, dt, info

    # ---- pdefilter.py:118-165
    def solution_generator(self, pde, *, stop_at=None, progressbar=False):
        stopper = _TimeStopper(stop_at) if stop_at is not None else None
        state = self.initialize(pde)
        info = dict(num_f_evaluations=0, num_df_evaluations=0, num_df_diagonal_evaluations=0,
                    num_steps=0, num_attempted_steps=0)
        yield state, info
        dt = self.steprule.first_dt(pde)
        while state.t < pde.tmax:
            if stopper is not None:
                dt = stopper.adjust_dt_to_time_stops(state.t, dt)
            state, dt, sinfo = self.perform_full_step(state, dt, pde)
            info["num_steps"] += 1
            for key in ("num_f_evaluations", "num_df_evaluations", "num_df_diagonal_evaluations",
                        "num_attempted_steps"):
                info[key] += sinfo[key]
            yield state, info

    # ---- pdefilter.py:75-103
    def solve(self, pde, **kw):
        ts, means, covs, d2, info = [], [], [], [], {}
        for state, info in self.solution_generator(pde, **kw):
            ts.append(state.t), means.append(state.y.mean), covs.append(state.y.cov_sqrtm)
            if isinstance(state.diffusion_squared_local, list):
                d2.extend(state.diffusion_squared_local)
            else:
                d2.append(state.diffusion_squared_local)
        return Solution(np.stack(ts), np.stack(means), np.stack(covs), info, float(np.mean(np.array(d2))))

    # ---- pdefilter.py:105-116
    def simulate_final_state(self, pde, **kw):
        state, info, d2 = None, None, []
        for state, info in self.solution_generator(pde, **kw):
            if isinstance(state.diffusion_squared_local, list):
                d2.extend(state.diffusion_squared_local)
            else:
                d2.append(state.diffusion_squared_local)
        c = state.y.cov_sqrtm * np.sqrt(np.mean(np.array(d2)))
        return state._replace(y=state.y._replace(cov_sqrtm=c)), info


# --------------------------------------------------------------------------------------
# Latent-force EK1 (reference: latent.py:11-292, base/stacked_ssm.py:7-80).  The state is the
# stack [u; eps] of two IWPs (diffusions chol K and pde.E_sqrtm); the PDE rows measure
# E1 u - L E0 u - E0 eps, the BC rows B E0 u, both noise-free (update_sqrt_no_meascov).
# Parity: pinned only by the reference's no-NaN smoke test (tests/test_pdefilter.py:140-145),
# i.e. numerically UNPINNED; the covariance-form restatement below is the cross-check.
# --------------------------------------------------------------------------------------


class LatentForceEK1(WhiteNoiseEK1):
    """`LinearLatentForceEK1` (semilinear=False) / `SemiLinearLatentForceEK1` (True)."""

    def __init__(self, **kw):
        super().__init__(**kw)
        self.state_iwp = self.lf_iwp = None

    # ---- latent.py:241-292
    def evaluate_ode(self, pde, p0, p1, m_pred, t, p_state, p_eps):
        L, B = pde.L, pde.B
        E0s, E0e, E1s = p0 @ p_state, p0 @ p_eps, p1 @ p_state
        m_at = scipy.linalg.block_diag(E0s, E0e) @ m_pred
        state_at = m_at[: m_at.shape[0] // 2]
        if self.semilinear:
            fx, Jx = pde.f(t, state_at), pde.df(t, state_at)
            H_state = E1s - Jx @ E0s - L @ E0s
        else:
            fx, Jx = L @ state_at, L
            H_state = E1s - Jx @ E0s
        Hb = B @ E0s
        H = np.block([[H_state, -E0e], [Hb, np.zeros_like(Hb)]])
        z = H @ m_pred + np.hstack((Jx @ state_at - fx, np.zeros(B.shape[0])))
        return z, H

    # ---- latent.py:136-153
    def initialize_iwp_latent(self, pde):
        X = pde.mesh_spatial.points
        gamma = np.linalg.cholesky(self.spatial_kernel(X, X.T))
        d = pde.y0.shape[0]
        prior_state, prior_latent = IWP(d, self.num_derivatives, gamma), IWP(d, self.num_derivatives, pde.E_sqrtm)
        return prior_state, prior_latent, prior_latent.projection_matrix(0), prior_latent.projection_matrix(1), gamma

    # ---- latent.py:20-134
    def initialize(self, pde):
        self.state_iwp, self.lf_iwp, self.E0, self.E1, gamma = self.initialize_iwp_latent(pde)
        self.iwp = self.state_iwp
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        c0 = self.diffuse_prior_scale * np.eye(n)
        C_state, C_latent = np.kron(gamma, c0), np.kron(pde.E_sqrtm, c0)
        C_y0, k_y0, _ = update_sqrt(self.E0, C_state, 1e-6 * np.eye(d))
        m_stack = np.hstack((k_y0 @ pde.y0, np.zeros(n * d)))
        C_block = scipy.linalg.block_diag(C_y0, C_latent)
        eye = np.eye(n * d)
        z, H = self.evaluate_ode(pde, self.E0, self.E1, m_stack, pde.t0, eye, eye)
        C0, k, _ = update_sqrt(H, C_block, 1e-6 * np.eye(d + pde.B.shape[0]))
        m0 = m_stack - k @ z
        glued = np.hstack((m0[: n * d].reshape((n, d), order="F"), m0[n * d:].reshape((n, d), order="F")))
        return FilterState(t=pde.t0, y=MVN(glued, C0), error_estimate=None, reference_state=None,
                           diffusion_squared_local=[])

    # ---- latent.py:155-233 (stacked_ssm.py:17-52: block_diag of the two processes)
    def attempt_step(self, state, dt, pde):
        Ps, Pis = self.state_iwp.nordsieck_preconditioner(dt)
        Pe, Pie = self.lf_iwp.nordsieck_preconditioner(dt)
        P, Pinv = scipy.linalg.block_diag(Ps, Pe), scipy.linalg.block_diag(Pis, Pie)
        (As, Qs), (Ae, Qe) = self.state_iwp.preconditioned_discretize, self.lf_iwp.preconditioned_discretize
        A, Ql = scipy.linalg.block_diag(As, Ae), scipy.linalg.block_diag(Qs, Qe)
        n, d = self.num_derivatives + 1, pde.y0.shape[0]
        flat = np.hstack((state.y.mean[:, :d].reshape((-1,), order="F"), state.y.mean[:, d:].reshape((-1,), order="F")))
        m, Cl = Pinv @ flat, Pinv @ state.y.cov_sqrtm
        mp = A @ m
        z, H = self.evaluate_ode(pde, self.E0, self.E1, mp, state.t + dt, Ps, Pe)
        Clp = propagate_cholesky_factor(A @ Cl, Ql)
        Cl_new, K, Sl = update_sqrt(H, Clp, None)
        m_new = P @ (mp - K @ z)
        Cl_new = P @ Cl_new
        if self.canonical_factor_signs:  # quirk Q1, as in WhiteNoiseEK1.attempt_step
            Sl = Sl * np.sign(np.diag(Sl))[None, :]
        r = scipy.linalg.solve_triangular(Sl.T, z, lower=False)
        glued = np.hstack((m_new[: n * d].reshape((n, d), order="F"), m_new[n * d:].reshape((n, d), order="F")))
        new = FilterState(t=state.t + dt, y=MVN(glued, Cl_new), error_estimate=None, reference_state=None,
                          diffusion_squared_local=r @ r / r.shape[0])
        return new, dict(num_f_evaluations=1, num_df_evaluations=1)


def read_mean_and_std_latent(sol, E0):
    """experiments/figure1.py:83-89: the state half of the glued mean and marginal std."""
    d = sol.mean.shape[-1] // 2
    var = np.einsum("tij,tij->ti", sol.cov_sqrtm, sol.cov_sqrtm)
    return sol.mean[:, 0, :d], np.sqrt(var[:, : var.shape[1] // 2] @ E0.T)


def latent_covariance_form_step(solver, pde, glued_mean, cov, dt, t):
    """`LatentForceEK1.attempt_step` in covariance form (cov is (2D,2D) in the reference's stacked order).
    Returns (glued mean (n,2d), cov, z^T S^-1 z / m).  S may be semi-definite (exact Dirichlet rows): pinv."""
    n, d = solver.num_derivatives + 1, pde.y0.shape[0]
    Ps, Pis = solver.state_iwp.nordsieck_preconditioner(dt)
    Pe, Pie = solver.lf_iwp.nordsieck_preconditioner(dt)
    P, Pinv = scipy.linalg.block_diag(Ps, Pe), scipy.linalg.block_diag(Pis, Pie)
    (As, Qs), (Ae, Qe) = solver.state_iwp.preconditioned_discretize, solver.lf_iwp.preconditioned_discretize
    A, Ql = scipy.linalg.block_diag(As, Ae), scipy.linalg.block_diag(Qs, Qe)
    flat = np.hstack((glued_mean[:, :d].reshape((-1,), order="F"), glued_mean[:, d:].reshape((-1,), order="F")))
    mp = A @ (Pinv @ flat)
    Pm = A @ (Pinv @ cov @ Pinv.T) @ A.T + Ql @ Ql.T
    z, H = solver.evaluate_ode(pde, solver.E0, solver.E1, mp, t + dt, Ps, Pe)
    S = H @ Pm @ H.T
    Si = np.linalg.pinv(S, rcond=1e-13, hermitian=True)
    K = Pm @ H.T @ Si
    m_new = P @ (mp - K @ z)
    C_new = P @ (Pm - K @ S @ K.T) @ P.T
    glued = np.hstack((m_new[: n * d].reshape((n, d), order="F"), m_new[n * d:].reshape((n, d), order="F")))
    return glued, C_new, z @ Si @ z / z.shape[0]


def read_mean_and_std(sol, E0):
    """experiments/figure1.py:76-80: means (T+1,d), stds = sqrt(diag(C C^T) E0^T) (T+1,d)."""
    means = sol.mean[:, 0]
    var = np.einsum("tij,tij->ti", sol.cov_sqrtm, sol.cov_sqrtm)
    return means, np.sqrt(var @ E0.T)


def calibrated_mean_and_std(sol, E0):
    """experiments/figure1.py:13-24: (means, gamma * stds)."""
    means, stds = read_mean_and_std(sol, E0)
    return means, np.sqrt(sol.diffusion_squared_calibrated) * stds


# --------------------------------------------------------------------------------------
# Classic (covariance / Cholesky) form of the same step -- the form the north-star words
# and the GPU uses; licensed by tests/test_base/test_sqrt.py:48-78 of the reference.
# Used in tests as a second, independent restatement and as the "optimised CPU path".
# --------------------------------------------------------------------------------------


def covariance_form_step(solver, pde, mean_nd, cov, dt, t):
    """One predict+update in covariance form, same quantities as `attempt_step`.

    Returns (mean (n,d), cov (D,D), sigma2_local, error (d,)).  Dense, un-optimised.
    """
    iwp = solver.iwp
    n, d = solver.num_derivatives + 1, pde.y0.shape[0]
    P, Pinv = iwp.nordsieck_preconditioner(dt)
    A, Ql = iwp.preconditioned_discretize
    m = Pinv @ mean_nd.reshape((-1,), order="F")
    Cp = Pinv @ cov @ Pinv.T
    mp = A @ m
    Q = Ql @ Ql.T
    Pm = A @ Cp @ A.T + Q
    z, H, E = solver.evaluate_ode(pde, solver.E0 @ P, solver.E1 @ P, mp, t + dt)
    R = E @ E.T
    S = H @ Pm @ H.T + R
    Ls = np.linalg.cholesky(S)
    Wt = scipy.linalg.solve_triangular(Ls, H @ Pm, lower=True)
    r = scipy.linalg.solve_triangular(Ls, z, lower=True)
    m_new = mp - Wt.T @ r
    C_new = Pm - Wt.T @ Wt
    Sq = H @ Q @ H.T + R
    sig = np.sqrt(z @ np.linalg.solve(Sq, z) / z.shape[0])
    err = dt * (np.sqrt(np.diag(Sq)) * sig)[: -pde.B.shape[0]]
    return (P @ m_new).reshape((n, d), order="F"), P @ C_new @ P.T, r @ r / r.shape[0], err


def covariance_form_initialize(solver, pde):
    """white.py:12-80 in covariance form.  Returns (mean (n,d), cov (D,D))."""
    solver.iwp, solver.E0, solver.E1, gamma = solver.initialize_iwp(pde)
    n, d = solver.num_derivatives + 1, pde.L.shape[0]
    C0 = np.kron(gamma @ gamma.T, solver.diffuse_prior_scale**2 * np.eye(n))

    def upd(mean, cov, H, z, R):
        S = H @ cov @ H.T + R
        Ls = np.linalg.cholesky(S)
        Wt = scipy.linalg.solve_triangular(Ls, H @ cov, lower=True)
        return mean - Wt.T @ scipy.linalg.solve_triangular(Ls, z, lower=True), cov - Wt.T @ Wt

    m1, C1 = upd(np.zeros(n * d), C0, solver.E0, -pde.y0, 1e-20 * np.eye(d))
    z, H, E = solver.evaluate_ode(pde, solver.E0, solver.E1, m1, pde.t0)
    En = E + 1e-10 * np.eye(d + pde.B.shape[0])
    m2, C2 = upd(m1, C1, H, z, En @ En.T)
    return m2.reshape((n, d), order="F"), C2
