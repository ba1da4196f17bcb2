"""Step-size selection (reference: src/pnmol/odetools/)."""

from . import step  # noqa: F401
