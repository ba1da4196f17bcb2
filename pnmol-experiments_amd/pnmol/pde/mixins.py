"""Problem mix-ins: discretisation, IVP data, boundary conditions (reference: src/pnmol/pde/mixins.py).

The tornadox IVP conversion of the reference (`to_tornadox_ivp`, mixins.py:149-214) is replaced by
`to_ivp()`, which returns the same (f, df, y0, t0, tmax) without the third-party container.
"""

from types import SimpleNamespace

import numpy as np

from .. import discretize


class DiscretizationMixIn:
    """`discretize()` fills L, E_sqrtm, B, R_sqrtm, y0, mesh_spatial (mixins.py:16-59)."""

    def discretize(self, *, mesh_spatial, kernel, stencil_size_interior, stencil_size_boundary,
                   nugget_gram_matrix=0.0):
        L, E_sqrtm = discretize.fd_probabilistic(
            self.diffop, mesh_spatial=mesh_spatial, kernel=kernel, stencil_size_interior=stencil_size_interior,
            stencil_size_boundary=stencil_size_boundary, nugget_gram_matrix=nugget_gram_matrix)
        self.L = self.diffop_scale * L
        self.E_sqrtm = self.diffop_scale * E_sqrtm
        self.mesh_spatial = mesh_spatial
        if isinstance(self, NeumannMixIn):
            if self.dimension > 1:
                raise NotImplementedError
            self.B, self.R_sqrtm = discretize.fd_probabilistic_neumann_1d(
                mesh_spatial=mesh_spatial, kernel=kernel, stencil_size=2, nugget_gram_matrix=nugget_gram_matrix)
        elif isinstance(self, DirichletMixIn):
            self.B = mesh_spatial.boundary_projection_matrix
            self.R_sqrtm = np.zeros((self.B.shape[0], self.B.shape[0]))
        if isinstance(self, IVPMixIn):
            self.y0 = self.y0_fun(mesh_spatial.points)[:, 0]


class IVPMixIn:
    """t0, tmax, y0_fun (mixins.py:128-146)."""

    def __init__(self, *, t0, tmax, y0_fun, **kwargs):
        self.t0, self.tmax, self.y0_fun = t0, tmax, y0_fun
        self.y0 = None
        super().__init__(**kwargs)

    @property
    def t_span(self):
        return self.t0, self.tmax


class IVPConversionLinearMixIn:
    """Method-of-lines IVP y' = L y on the interior nodes (mixins.py:177-193)."""

    def to_ivp(self):
        if self.L is None:
            raise AttributeError("Conversion to an IVP requires prior discretization.")
        if self.dimension > 1:
            raise NotImplementedError
        n_in = self.L.shape[0] - 2
        J = np.stack([self.bc_remove_pad(self.L @ self.bc_pad(e)) for e in np.eye(n_in)], axis=1)
        return SimpleNamespace(f=lambda _t, x: self.bc_remove_pad(self.L @ self.bc_pad(x)), df=lambda _t, _x: J,
                               y0=self.bc_remove_pad(self.y0), t0=self.t0, tmax=self.tmax,
                               t_span=(self.t0, self.tmax))


class _BoundaryConditionMixInInterface:
    def __init__(self, **kwargs):
        self.B = None
        self.R_sqrtm = None
        super().__init__(**kwargs)


class NeumannMixIn(_BoundaryConditionMixInInterface):
    def bc_pad(self, x):  # mixins.py:259-266
        return np.pad(x, pad_width=1, mode="edge")

    def bc_remove_pad(self, x):
        return x[1:-1]


class DirichletMixIn(_BoundaryConditionMixInInterface):
    def bc_pad(self, x):  # mixins.py:269-284
        return np.pad(x, pad_width=1, mode="constant", constant_values=0.0)

    def bc_remove_pad(self, x):
        return x[1:-1]


class NonLinearMixIn:
    def __init__(self, *, f, df, df_diagonal, **kwargs):
        self.f, self.df, self.df_diagonal = f, df, df_diagonal
        super().__init__(**kwargs)
