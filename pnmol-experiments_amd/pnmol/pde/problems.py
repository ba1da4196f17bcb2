"""PDE problem classes (reference: src/pnmol/pde/problems.py)."""

import numpy as np

from . import mixins


class PDE:
    """Differential operator + scale + bounding box; `discretize()` fills L, E_sqrtm (problems.py:11-42)."""

    def __init__(self, *, diffop, diffop_scale, bbox, **kwargs):
        self.diffop, self.diffop_scale, self.bbox = diffop, diffop_scale, np.asarray(bbox, dtype=np.float64)
        self.L = None
        self.E_sqrtm = None
        self.mesh_spatial = None
        super().__init__(**kwargs)

    def __repr__(self):
        return f"{self.__class__.__name__}(is_discretized={self.is_discretized})"

    @property
    def is_discretized(self):
        return self.L is not None

    @property
    def dimension(self):
        return self.bbox.ndim


class LinearEvolutionDirichlet(mixins.IVPMixIn, mixins.IVPConversionLinearMixIn, mixins.DiscretizationMixIn,
                               mixins.DirichletMixIn, PDE):
    """Linear evolution equation, Dirichlet boundary (problems.py:45-54)."""


class LinearEvolutionNeumann(mixins.IVPMixIn, mixins.IVPConversionLinearMixIn, mixins.DiscretizationMixIn,
                             mixins.NeumannMixIn, PDE):
    """Linear evolution equation, Neumann boundary (problems.py:57-66)."""


class SemiLinearEvolutionDirichlet(mixins.IVPMixIn, mixins.NonLinearMixIn, mixins.DiscretizationMixIn,
                                   mixins.DirichletMixIn, PDE):
    pass


class SemiLinearEvolutionNeumann(mixins.IVPMixIn, mixins.NonLinearMixIn, mixins.DiscretizationMixIn,
                                 mixins.NeumannMixIn, PDE):
    pass


class SystemLinearPDENeumann(mixins.SystemDiscretizationMixIn, mixins.NeumannMixIn, PDE):
    """Systems of linear PDEs with Neumann boundary conditions (problems.py:70-73)."""


class SystemSemiLinearEvolutionNeumann(mixins.IVPMixIn, mixins.NonLinearMixIn, mixins.IVPConversionSemiLinearMixIn,
                                       mixins.SystemDiscretizationMixIn, mixins.SystemNeumannMixIn, PDE):
    """Systems of semilinear, time-dependent PDEs with Neumann boundary conditions (problems.py:76-86)."""
