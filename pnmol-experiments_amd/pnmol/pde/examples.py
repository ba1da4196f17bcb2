"""Problem recipes (reference: src/pnmol/pde/examples.py:13-81, :347-357).

Heat equation and spruce-budworm (Fisher) recipes; the SIR / Lotka-Volterra systems of PDEs are out of scope
(SURVEY.md section 2, row 11).
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

    cls = {"dirichlet": problems.SemiLinearEvolutionDirichlet, "neumann": problems.SemiLinearEvolutionNeumann}.get(bcond)
    if cls is None:
        raise ValueError
    return cls(t0=t0, tmax=tmax, y0_fun=y0_fun, bbox=bbox, diffop=diffops.laplace(), diffop_scale=diffusion_rate,
               f=f_spruce, df=df_spruce, df_diagonal=None)


# Initial-condition defaults; they adhere to Dirichlet conditions (examples.py:344-357)


def gaussian_bell_1d_centered(x, bbox, width=1.0):
    midpoint = 0.5 * (bbox[1] + bbox[0])
    return np.exp(-((x - midpoint) ** 2) / width ** 2)


def gaussian_bell_1d(x):
    return np.exp(-(x ** 2))


def sin_bell_1d(x):
    return 0.1 * np.sin(np.pi * x)
