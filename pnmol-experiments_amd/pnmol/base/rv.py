"""Random variables (reference: src/pnmol/base/rv.py)."""

from collections import namedtuple

import numpy as np


class MultivariateNormal(namedtuple("_MultivariateNormal", "mean cov_sqrtm")):
    """N(mean, cov_sqrtm cov_sqrtm^T) with host arrays (rv.py:9-14)."""

    @property
    def cov(self):
        return self.cov_sqrtm @ self.cov_sqrtm.T


def factor_of(cov):
    """A matrix C with C C^T = cov (projected to PSD).  Not triangular: the GPU path carries the
    covariance, and the reference's consumers only ever use C C^T (figure1.py:76-80)."""
    lam, V = np.linalg.eigh(0.5 * (cov + cov.T))
    return V * np.sqrt(np.maximum(lam, 0.0))[None, :]


class DeviceMultivariateNormal:
    """Same attribute surface as `MultivariateNormal`, covariance resident on the GPU.

    `mean` is a host (n, d) array; `cov`, `cov_sqrtm`, `marginal_var` are fetched / computed on the device on demand.
    """

    def __init__(self, mean, device_state):
        self.mean = mean
        self.device_state = device_state
        self._cov_sqrtm = None

    @property
    def cov(self):
        return self.device_state.cov()

    @property
    def marginal_var(self):
        return self.device_state.marginal_var()

    @property
    def cov_sqrtm(self):
        """Lower-triangular Cholesky factor of the covariance, computed on the device on first use (the reference
        carries a QR-derived lower-triangular factor with arbitrary column signs; this is its positive-diagonal
        representative; a direction of exactly zero variance gives a zero column)."""
        if self._cov_sqrtm is None:
            self._cov_sqrtm = self.device_state.cov_sqrtm()
        return self._cov_sqrtm

    def _replace(self, **kw):
        return MultivariateNormal(mean=kw.get("mean", self.mean), cov_sqrtm=kw.get("cov_sqrtm", None)
                                  if "cov_sqrtm" in kw else self.cov_sqrtm)

    def __iter__(self):
        return iter((self.mean, self.cov_sqrtm))
