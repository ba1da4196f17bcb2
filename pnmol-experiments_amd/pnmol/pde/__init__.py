"""PDE problems (reference: src/pnmol/pde/__init__.py)."""

from . import examples, mixins, problems  # noqa: F401
