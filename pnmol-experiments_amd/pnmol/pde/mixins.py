"""Problem mix-ins: discretisation, IVP data, boundary conditions (reference: src/pnmol/pde/mixins.py).

The tornadox IVP conversion of the reference (`to_tornadox_ivp`, mixins.py:149-214) is replaced by
`to_ivp()`, which returns the same (f, df, y0, t0, tmax) without the third-party container.
"""

from types import SimpleNamespace

import functools

import numpy as np
import scipy.linalg

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


class SystemDiscretizationMixIn:
    """`discretize_system()` for systems of PDEs (mixins.py:62-125): `diffop` / `diffop_scale` are tuples, one per
    component; L and E_sqrtm are block diagonal, B and R_sqrtm repeat the scalar boundary operator per component."""

    def discretize_system(self, *, mesh_spatial, kernel, stencil_size_interior, stencil_size_boundary,
                          nugget_gram_matrix=0.0):
        fd = functools.partial(discretize.fd_probabilistic, mesh_spatial=mesh_spatial, kernel=kernel,
                               stencil_size_interior=stencil_size_interior, stencil_size_boundary=stencil_size_boundary,
                               nugget_gram_matrix=nugget_gram_matrix)
        scaled = [(s * L, s * E) for s, (L, E) in zip(self.diffop_scale, map(fd, self.diffop))]
        self.L = scipy.linalg.block_diag(*[L for L, _ in scaled])
        self.E_sqrtm = scipy.linalg.block_diag(*[E for _, E in scaled])
        self.mesh_spatial = mesh_spatial
        if isinstance(self, _BoundaryConditionMixInInterface):
            if isinstance(self, (NeumannMixIn, SystemNeumannMixIn)):
                if self.dimension > 1:
                    raise NotImplementedError
                B, R_sqrtm = discretize.fd_probabilistic_neumann_1d(
                    mesh_spatial=mesh_spatial, kernel=kernel, stencil_size=2, nugget_gram_matrix=nugget_gram_matrix)
            else:   # (the reference reads self.B before it is set here, mixins.py:113; no Dirichlet system class uses it)
                B = mesh_spatial.boundary_projection_matrix
                R_sqrtm = np.zeros((B.shape[0], B.shape[0]))
            n = len(self.diffop)
            self.B = scipy.linalg.block_diag(*([B] * n))
            self.R_sqrtm = scipy.linalg.block_diag(*([R_sqrtm] * n))
        if isinstance(self, IVPMixIn):
            self.y0 = np.asarray(self.y0_fun(mesh_spatial.points)).squeeze()


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


class _SystemBoundaryConditionMixinInterface(_BoundaryConditionMixInInterface):
    """Component-wise padding of a stacked state (mixins.py:223-244)."""

    def __init__(self, *, bc, **kwargs):
        self.bc = bc
        super().__init__(**kwargs)

    def bc_pad(self, x):
        n = len(self.diffop)
        return np.apply_along_axis(self.bc.bc_pad, -1, np.reshape(x, (n, -1))).reshape((-1,))

    def bc_remove_pad(self, x):
        n = len(self.diffop)
        return np.apply_along_axis(self.bc.bc_remove_pad, -1, np.reshape(x, (n, -1))).reshape((-1,))


class SystemNeumannMixIn(_SystemBoundaryConditionMixinInterface):
    def __init__(self, **kwargs):
        super().__init__(bc=NeumannMixIn(), **kwargs)


class SystemDirichletMixIn(_SystemBoundaryConditionMixinInterface):
    def __init__(self, **kwargs):
        super().__init__(bc=DirichletMixIn(), **kwargs)


class IVPConversionSemiLinearMixIn:
    """Method-of-lines IVP y' = L y + f(t, y) on the interior nodes (mixins.py:195-214); `df` by central differences of
    the padded right-hand side (the reference uses jax.jacfwd)."""

    def to_ivp(self):
        if self.L is None:
            raise AttributeError("Conversion to an IVP requires prior discretization.")

        def f_new(t, x):
            xp = self.bc_pad(x)
            return self.bc_remove_pad(self.L @ xp + self.f(t, xp))

        def df_new(t, x, h=1e-6):
            x = np.asarray(x, dtype=np.float64)
            return np.stack([(f_new(t, x + h * e) - f_new(t, x - h * e)) / (2 * h) for e in np.eye(x.size)], axis=1)

        return SimpleNamespace(f=f_new, df=df_new, y0=self.bc_remove_pad(self.y0), t0=self.t0, tmax=self.tmax,
                               t_span=(self.t0, self.tmax))


class NonLinearMixIn:
    def __init__(self, *, f, df, df_diagonal, **kwargs):
        self.f, self.df, self.df_diagonal = f, df, df_diagonal
        super().__init__(**kwargs)
