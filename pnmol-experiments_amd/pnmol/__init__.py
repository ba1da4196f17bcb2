"""PNMOL on MI355X: host-side mirror of the reference's `pnmol` package for the white-noise
and latent-force EK1 paths (reference: src/pnmol/__init__.py).  Cold path (mesh, kernels, discretisation,
problem recipes) is NumPy on the host; the filter step runs in libpnmol_hip.so."""

from . import diffops, discretize, kernels, latent, mesh, odetools, pde, pdefilter, sqrtform, white  # noqa: F401

__version__ = "0.1.0"
