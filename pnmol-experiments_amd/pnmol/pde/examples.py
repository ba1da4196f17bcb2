"""Problem recipes (reference: src/pnmol/pde/examples.py:13-81, :347-357).

Heat equation (1-d, and a 2-d Dirichlet recipe assembled from the reference's parts), spruce-budworm (Fisher), and the
Lotka-Volterra / SIR systems of PDEs that figures 3-4 run (examples.py:84-248), with closed-form Jacobians in place of
`jax.jacfwd`.
"""

import functools

import numpy as np

from .. import diffops, kernels, mesh
from . import problems


def heat_1d_discretized(*, bbox=None, dx=0.05, stencil_size_interior=3, stencil_size_boundary=3, t0=0.0, tmax=5.0,
                        y0_fun=None, diffusion_rate=0.05, nugget_gram_matrix_fd=0.0, kernel=None,
                        bcond="dirichlet"):
    heat = heat_1d(bbox=bbox, t0=t0, tmax=tmax, y0_fun=y0_fun, diffusion_rate=diffusion_rate, bcond=bcond)
    mesh_spatial = mesh.RectangularMesh.from_bbox_1d(heat.bbox, step=dx)
    if kernel is None:
        kernel = kernels.SquareExponential()
    heat.discretize(mesh_spatial=mesh_spatial, kernel=kernel, stencil_size_interior=stencil_size_interior,
                    stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix=nugget_gram_matrix_fd)
    return heat


def heat_1d(*, bbox=None, t0=0.0, tmax=5.0, y0_fun=None, diffusion_rate=0.05, bcond="dirichlet"):
    laplace = diffops.laplace()
    if bbox is None:
        bbox = [0.0, 1.0]
    bbox = np.asarray(bbox, dtype=np.float64)
    if y0_fun is None:
        bell_centered = functools.partial(gaussian_bell_1d_centered, bbox=bbox)
        y0_fun = lambda x: bell_centered(x) * sin_bell_1d(x)  # noqa: E731
    cls = {"dirichlet": problems.LinearEvolutionDirichlet, "neumann": problems.LinearEvolutionNeumann}.get(bcond)
    if cls is None:
        raise ValueError
    return cls(diffop=laplace, diffop_scale=diffusion_rate, bbox=bbox, t0=t0, tmax=tmax, y0_fun=y0_fun)


def heat_2d_dirichlet_discretized(*, nums=(64, 64), stencil_size_interior=5, stencil_size_boundary=5, t0=0.0,
                                  tmax=1.0, diffusion_rate=0.05, kernel=None, y0_fun=None):
    """2-d heat problem assembled from the reference's parts (mesh.py:100-130, mixins.py:51-54);
    the reference itself ships no 2-d recipe."""
    bbox = np.array([[0.0, 0.0], [1.0, 1.0]])
    if y0_fun is None:
        y0_fun = lambda p: (0.1 * np.sin(np.pi * p[:, 0]) * np.sin(np.pi * p[:, 1]))[:, None]  # noqa: E731
    heat = problems.LinearEvolutionDirichlet(diffop=diffops.laplace(), diffop_scale=diffusion_rate, bbox=bbox,
                                             t0=t0, tmax=tmax, y0_fun=y0_fun)
    heat.discretize(mesh_spatial=mesh.RectangularMesh.from_bbox_2d(bbox, nums=nums),
                    kernel=kernel or kernels.SquareExponential(), stencil_size_interior=stencil_size_interior,
                    stencil_size_boundary=stencil_size_boundary)
    return heat


def sir_1d_discretized(*, bbox=None, dx=0.05, t0=0.0, tmax=50.0, beta=0.3, gamma=0.07, N=1000.0, diffusion_rate_S=0.1,
                       diffusion_rate_I=0.1, diffusion_rate_R=0.1, kernel=None, nugget_gram_matrix_fd=0.0,
                       stencil_size_interior=3, stencil_size_boundary=3):
    """examples.py:84-124."""
    sir = sir_1d(bbox=bbox, t0=t0, tmax=tmax, diffusion_rate_S=diffusion_rate_S, diffusion_rate_I=diffusion_rate_I,
                 diffusion_rate_R=diffusion_rate_R, beta=beta, gamma=gamma, N=N)
    mesh_spatial = mesh.RectangularMesh.from_bbox_1d(sir.bbox, step=dx)
    if kernel is None:
        kernel = kernels.SquareExponential()
    sir.discretize_system(mesh_spatial=mesh_spatial, kernel=kernel, stencil_size_interior=stencil_size_interior,
                          stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix=nugget_gram_matrix_fd)
    return sir


def sir_1d(*, bbox=None, t0=0.0, tmax=50.0, diffusion_rate_S=0.1, diffusion_rate_I=0.1, diffusion_rate_R=0.1, beta=0.3,
           gamma=0.07, N=1000.0):
    """Spatial SIR model, state [S; I; R], Neumann boundary (examples.py:127-178); `df` is the closed-form Jacobian of
    the reaction term (the reference differentiates it with jax.jacfwd)."""
    if bbox is None:
        bbox = [0.0, 1.0]
    bbox = np.asarray(bbox, dtype=np.float64)

    def y0_fun(x):
        init_infectious = 200.0 * gaussian_bell_1d_centered(x, bbox, width=0.5) + 1.0
        return np.concatenate((N * np.ones_like(init_infectious) - init_infectious, init_infectious,
                               np.zeros_like(init_infectious)))

    def f(_t, x):
        s, i, r = np.split(np.asarray(x, dtype=np.float64), 3)
        spatial_n = s + i + r
        return np.concatenate((-beta * s * i / spatial_n, beta * s * i / spatial_n - gamma * i, gamma * i))

    def df(_t, x):
        s, i, r = np.split(np.asarray(x, dtype=np.float64), 3)
        tot = s + i + r
        g = beta * s * i / tot
        gs, gi, gr = beta * i / tot - g / tot, beta * s / tot - g / tot, -g / tot
        zero = np.zeros((s.size, s.size))
        return np.block([[np.diag(-gs), np.diag(-gi), np.diag(-gr)], [np.diag(gs), np.diag(gi - gamma), np.diag(gr)],
                         [zero, np.diag(gamma * np.ones_like(i)), zero]])

    laplace = diffops.laplace()
    return problems.SystemSemiLinearEvolutionNeumann(
        diffop=(laplace, laplace, laplace), diffop_scale=(diffusion_rate_S, diffusion_rate_I, diffusion_rate_R),
        bbox=bbox, t0=t0, tmax=tmax, y0_fun=y0_fun, f=f, df=df, df_diagonal=None)


def lotka_volterra_1d_discretized(*, dx=0.05, kernel=None, nugget_gram_matrix_fd=0.0, stencil_size_interior=3,
                                  stencil_size_boundary=3, **kwargs):
    """examples.py:181-203."""
    pde = lotka_volterra_1d(**kwargs)
    mesh_spatial = mesh.RectangularMesh.from_bbox_1d(pde.bbox, step=dx)
    if kernel is None:
        kernel = kernels.SquareExponential()
    pde.discretize_system(mesh_spatial=mesh_spatial, kernel=kernel, stencil_size_interior=stencil_size_interior,
                          stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix=nugget_gram_matrix_fd)
    return pde


def lotka_volterra_1d(*, bbox=None, t0=0.0, tmax=10.0, a=0.5, b=0.05, c=0.05, d=0.5, diffusion_scale_u=0.1,
                      diffusion_scale_v=0.1):
    """Predator-prey system u_t = D_u u_xx + a u - b u v, v_t = D_v v_xx + c u v - d v, state [u; v], Neumann boundary
    (examples.py:206-248, the workload of experiments/figure4.py); closed-form Jacobian instead of jax.jacfwd."""
    if bbox is None:
        bbox = [0.0, 1.0]
    bbox = np.asarray(bbox, dtype=np.float64)

    def y0_fun(x):
        return np.concatenate((5 * np.ones_like(x), 20.0 * gaussian_bell_1d(x)))

    def f_lotka_volterra(_, x):
        u, v = np.split(np.asarray(x, dtype=np.float64), 2)
        return np.concatenate((a * u - b * u * v, c * u * v - d * v))

    def df_lotka_volterra(_, x):
        u, v = np.split(np.asarray(x, dtype=np.float64), 2)
        return np.block([[np.diag(a - b * v), np.diag(-b * u)], [np.diag(c * v), np.diag(c * u - d)]])

    laplace = diffops.laplace()
    return problems.SystemSemiLinearEvolutionNeumann(
        diffop=(laplace, laplace), diffop_scale=(diffusion_scale_u, diffusion_scale_v), bbox=bbox, t0=t0, tmax=tmax,
        y0_fun=y0_fun, f=f_lotka_volterra, df=df_lotka_volterra, df_diagonal=None)


def spruce_budworm_1d_discretized(*, bbox=None, t0=0.0, tmax=10.0, diffusion_rate=1.0, y0_fun=None, dx=0.1, kernel=None,
                                  nugget_gram_matrix_fd=0.0, stencil_size_interior=3, stencil_size_boundary=3,
                                  bcond="dirichlet", growth_rate=1.0):
    """examples.py:251-289."""
    spruce = spruce_budworm_1d(bbox=bbox, t0=t0, tmax=tmax, diffusion_rate=diffusion_rate, y0_fun=y0_fun, bcond=bcond,
                               growth_rate=growth_rate)
    mesh_spatial = mesh.RectangularMesh.from_bbox_1d(spruce.bbox, step=dx)
    if kernel is None:
        kernel = kernels.SquareExponential()
    spruce.discretize(mesh_spatial=mesh_spatial, kernel=kernel, stencil_size_interior=stencil_size_interior,
                      stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix=nugget_gram_matrix_fd)
    return spruce


def spruce_budworm_1d(*, bbox=None, t0=0.0, tmax=10.0, diffusion_rate=0.1, y0_fun=None, bcond="dirichlet",
                      growth_rate=1.0):
    """Fisher's equation u_t = kappa u_xx + c u (1 - u)  (examples.py:292-341); df is the (diagonal) Jacobian."""
    if bbox is None:
        bbox = [0.0, 1.0]
    bbox = np.asarray(bbox, dtype=np.float64)
    if y0_fun is None:
        y0_fun = sin_bell_1d

    def f_spruce(_, x, c=growth_rate):
        return c * x * (1.0 - x)

    def df_spruce(_, x, c=growth_rate):
        return np.diag(c * (1.0 - 2.0 * np.asarray(x)))

    def df_diagonal_spruce(_, x, c=growth_rate):   # (the reference passes df_diagonal=None; the Jacobian IS diagonal)
        return c * (1.0 - 2.0 * np.asarray(x))

    cls = {"dirichlet": problems.SemiLinearEvolutionDirichlet, "neumann": problems.SemiLinearEvolutionNeumann}.get(bcond)
    if cls is None:
        raise ValueError
    return cls(t0=t0, tmax=tmax, y0_fun=y0_fun, bbox=bbox, diffop=diffops.laplace(), diffop_scale=diffusion_rate,
               f=f_spruce, df=df_spruce, df_diagonal=df_diagonal_spruce)


# Initial-condition defaults; they adhere to Dirichlet conditions (examples.py:344-357)


def gaussian_bell_1d_centered(x, bbox, width=1.0):
    midpoint = 0.5 * (bbox[1] + bbox[0])
    return np.exp(-((x - midpoint) ** 2) / width ** 2)


def gaussian_bell_1d(x):
    return np.exp(-(x ** 2))


def sin_bell_1d(x):
    return 0.1 * np.sin(np.pi * x)
