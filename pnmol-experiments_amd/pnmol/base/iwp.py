"""Integrated Wiener process prior (reference: src/pnmol/base/iwp.py).

Host-side description of the prior.  The dense Kronecker matrices are only materialised when a
caller asks for them (tests, `projection_matrix`); the device works on the n x n blocks A1, Q1.
"""

from collections import namedtuple
from functools import cached_property

import numpy as np
import scipy.linalg
import scipy.special


class IntegratedWienerTransition(namedtuple("_IWP", "wiener_process_dimension num_derivatives wp_diffusion_sqrtm")):
    @cached_property
    def preconditioned_discretize_1d(self):
        """(A_1d, chol(Q_1d)); np.flip without axis reverses both axes (iwp.py:13-30)."""
        n = self.num_derivatives + 1
        A_1d = np.flip(scipy.linalg.pascal(n, kind="lower", exact=False))
        Q_1d = np.flip(scipy.linalg.hilbert(n))
        return A_1d, np.linalg.cholesky(Q_1d)

    @cached_property
    def preconditioned_discretize(self):
        A_1d, L_Q1d = self.preconditioned_discretize_1d
        return np.kron(np.eye(self.wiener_process_dimension), A_1d), np.kron(self.wp_diffusion_sqrtm, L_Q1d)

    def nordsieck_preconditioner_1d_raw(self, dt):  # iwp.py:55-62
        powers = np.arange(self.num_derivatives, -1, -1)
        scales = scipy.special.factorial(powers)
        powers = powers + 0.5
        return (np.abs(dt) ** powers) / scales, (np.abs(dt) ** (-powers)) * scales

    def nordsieck_preconditioner_1d(self, dt):
        s, sinv = self.nordsieck_preconditioner_1d_raw(dt)
        return np.diag(s), np.diag(sinv)

    def nordsieck_preconditioner(self, dt):
        p, pinv = self.nordsieck_preconditioner_1d(dt)
        eye = np.eye(self.wiener_process_dimension)
        return np.kron(eye, p), np.kron(eye, pinv)

    def non_preconditioned_discretize(self, dt):
        P, Pinv = self.nordsieck_preconditioner(dt)
        A, Ql = self.preconditioned_discretize
        return P @ A @ Pinv, P @ Ql

    def projection_matrix(self, derivative_to_project_onto):
        # = np.kron(np.eye(d), e_q^T) (base/iwp.py:125-133), written directly: kron of a 4096-identity takes seconds
        d, n = self.wiener_process_dimension, self.num_derivatives + 1
        out = np.zeros((d, n * d))
        out[np.arange(d), np.arange(d) * n + derivative_to_project_onto] = 1.0
        return out

    def projection_matrix_1d(self, derivative_to_project_onto):
        return np.eye(1, self.num_derivatives + 1, derivative_to_project_onto)

    @property
    def state_dimension(self):
        return self.wiener_process_dimension * (self.num_derivatives + 1)
