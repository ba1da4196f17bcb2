"""Auxiliary pieces: IWP prior, random variables (reference: src/pnmol/base/)."""

from . import iwp, rv, sqrt, stacked_ssm  # noqa: F401
