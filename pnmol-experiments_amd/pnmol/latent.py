"""Latent-force EK1 PDE filters (reference: src/pnmol/latent.py).

The discretisation error is a second integrated Wiener process eps (diffusion pde.E_sqrtm) stacked under the
PDE state u; the PDE rows measure  E1 u - L E0 u - E0 eps = 0  and the boundary rows  B E0 u = 0, both without
measurement noise (latent.py:197).  On the device the stack is one IWP over 2d spatial components:
Gram = blockdiag(K, E E^T), stencil rows [L, I], boundary rows [B, 0] -- the same kernels as the white-noise
filter run it (`d_state = 2d` in `pnmol_filter_desc`).  State layout at the boundary is the reference's:
glued mean (n, 2d) = [u | eps], covariance in the stacked order [flat(u); flat(eps)] (latent.py:163-175).

Deviations: those of `pnmol.white` (non-triangular `cov_sqrtm`, Cholesky-factor `diffusion_squared_local`).
"""

import numpy as np
import scipy.linalg

from . import _hip, pdefilter, white
from .base import iwp, rv, stacked_ssm


class _LatentForceEK1Base(white._WhiteNoiseEK1Base):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.ssm = None
        self.state_iwp = None
        self.lf_iwp = None

    # ------------------------------------------------------------------ cold path
    def initialize_iwp_latent(self, pde):
        """Two IWPs: state (diffusion chol k(X,X)) and latent force (diffusion pde.E_sqrtm), latent.py:136-153."""
        X = pde.mesh_spatial.points
        diffusion_state_sqrtm = np.linalg.cholesky(self.spatial_kernel(X, X.T))
        d = pde.y0.shape[0]
        prior_state = iwp.IntegratedWienerTransition(num_derivatives=self.num_derivatives, wiener_process_dimension=d,
                                                     wp_diffusion_sqrtm=diffusion_state_sqrtm)
        prior_latent = iwp.IntegratedWienerTransition(num_derivatives=self.num_derivatives, wiener_process_dimension=d,
                                                      wp_diffusion_sqrtm=pde.E_sqrtm)
        return (prior_state, prior_latent, prior_latent.projection_matrix(0), prior_latent.projection_matrix(1),
                diffusion_state_sqrtm)

    @staticmethod
    def _stacked_operator(M):
        return np.hstack((M, np.eye(M.shape[0])))               # rows of  -(M E0 u + E0 eps)

    def _bind(self, pde, gamma):
        ctx = self._context or _hip.Context.default()
        d, nB = pde.L.shape[0], pde.B.shape[0]
        E = np.asarray(pde.E_sqrtm, dtype=np.float64)
        if np.any(np.triu(E, 1) != 0.0):
            raise ValueError("pde.E_sqrtm must be lower triangular (it is diagonal for every recipe, discretize.py:107-113)")
        self._device_filter = _hip.Filter(
            ctx, L=self._stacked_operator(pde.L), B=np.hstack((pde.B, np.zeros((nB, d)))),
            E_sqrtm=np.zeros((d, d)), R_sqrtm=np.zeros((nB, nB)), Gamma=scipy.linalg.block_diag(gamma, E),
            num_derivatives=self.num_derivatives)
        self._device_pde = pde
        self._gram = gamma @ gamma.T
        self._gram_latent = E @ E.T
        self._error_models = {}

    def initialize(self, pde):
        """Prior conditioned on y0 (nugget 1e-6), then the stack on the PDE/BC residual at t0 (nugget 1e-6)
        (latent.py:20-134), block-wise in closed form like `white._WhiteNoiseEK1Base.initialize`."""
        self.state_iwp, self.lf_iwp, self.E0, self.E1, gamma = self.initialize_iwp_latent(pde)
        self.iwp = self.state_iwp
        self.ssm = stacked_ssm.StackedSSM(processes=[self.state_iwp, self.lf_iwp])
        self._bind(pde, gamma)
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        mean, blocks = self._initial_moments(pde)
        D = n * d
        cov = np.zeros((2 * D, 2 * D))
        for (pa, a, pb, b), blk in blocks.items():              # (process, derivative) x (process, derivative)
            cov[pa * D + a:(pa + 1) * D:n, pb * D + b:(pb + 1) * D:n] = blk
        dev = self._device_filter.new_state()
        dev.set(pde.t0, mean, cov)
        return pdefilter.PDEFilterState(t=pde.t0, y=rv.DeviceMultivariateNormal(mean, dev), error_estimate=None,
                                        reference_state=None, diffusion_squared_local=[])

    def _initial_moments(self, pde):
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        nug2 = 1e-6 ** 2                                        # meascov_sqrtm = 1e-6 I, latent.py:70-75, :101-106
        c2 = self.diffuse_prior_scale ** 2
        Kc, Kec = c2 * self._gram, c2 * self._gram_latent
        S1 = scipy.linalg.cho_factor(Kc + nug2 * np.eye(d), lower=True)
        G1 = scipy.linalg.cho_solve(S1, Kc).T
        V0 = nug2 * 0.5 * (G1 + G1.T)
        m0 = G1 @ pde.y0
        M, shift = self._linearize(pde, m0, pde.t0)
        nB = pde.B.shape[0]
        if nB > 0:                                              # boundary rows touch derivative 0 of u only
            Sb = pde.B @ V0 @ pde.B.T + nug2 * np.eye(nB)
            Gb = np.linalg.solve(Sb, pde.B @ V0).T
            # the reference updates PDE and BC rows jointly at the linearisation point m0 (latent.py:88-107);
            # the rows are conditionally independent given the state, so processing them in sequence is the same
            m0b = m0 - Gb @ (pde.B @ m0)
            V0 = V0 - Gb @ (pde.B @ V0)
            V0 = 0.5 * (V0 + V0.T)
        else:
            m0b = m0
        # PDE rows: z = u1 - M u0 - eps0 + shift; prior blockdiag(V0, Kc, Kec); Sp = Kc + T
        T = M @ V0 @ M.T + Kec + nug2 * np.eye(d)
        Sp = scipy.linalg.cho_factor(Kc + T, lower=True)
        z = -M @ m0b + shift
        Spz = scipy.linalg.cho_solve(Sp, z)
        SpT = scipy.linalg.cho_solve(Sp, T)
        SpKe = scipy.linalg.cho_solve(Sp, Kec)
        V0Mt = V0 @ M.T
        mean = np.zeros((n, 2 * d))
        mean[0, :d] = m0b + V0Mt @ Spz
        mean[1, :d] = -(z - T @ Spz)
        mean[0, d:] = Kec @ Spz
        P00 = V0 - V0Mt @ scipy.linalg.cho_solve(Sp, V0Mt.T)
        P01 = V0Mt - V0Mt @ SpT
        P0e = -V0Mt @ SpKe
        P11 = T - T @ SpT
        P1e = Kec - T @ SpKe
        Pee = Kec - Kec @ SpKe
        sym = lambda X: 0.5 * (X + X.T)
        blocks = {(0, 0, 0, 0): sym(P00), (0, 0, 0, 1): P01, (0, 1, 0, 0): P01.T, (0, 1, 0, 1): sym(P11),
                  (0, 0, 1, 0): P0e, (1, 0, 0, 0): P0e.T, (0, 1, 1, 0): P1e, (1, 0, 0, 1): P1e.T,
                  (1, 0, 1, 0): sym(Pee)}
        for q in range(2, n):
            blocks[(0, q, 0, q)] = Kc
        for q in range(1, n):
            blocks[(1, q, 1, q)] = Kec
        return mean, blocks

    # ------------------------------------------------------------------ hot path
    def _ensure_error_model(self, pde, dt):
        pass                                                    # no error estimate in this model (latent.py:224)

    def attempt_step(self, state, dt, pde):
        """One predict + update + calibrate step on the GPU (latent.py:155-233); `state` is not modified."""
        dev_in = self._device_state_of(state, pde)
        flt = self._device_filter
        if self.semilinear:
            m_at = flt.predict_mean(dev_in, dt)                 # E0 u of the predicted mean
            M, shift = self._linearize(pde, m_at, state.t + dt)
            flt.set_operator(self._stacked_operator(M), shift)
        dev_out, info, _ = flt.step(dev_in, dt)
        self.last_step_info = info
        new_state = pdefilter.PDEFilterState(
            t=state.t + dt, error_estimate=None, reference_state=None,
            y=rv.DeviceMultivariateNormal(dev_out.mean(), dev_out),
            diffusion_squared_local=info.diffusion_squared_local)
        return new_state, dict(num_f_evaluations=1, num_df_evaluations=1)


class LinearLatentForceEK1(_LatentForceEK1Base):
    """u_t = L u (latent.py:241-263): H = [[E1 - L E0, -E0], [B E0, 0]], no shift."""

    _linearize = staticmethod(white.LinearWhiteNoiseEK1._linearize)


class SemiLinearLatentForceEK1(_LatentForceEK1Base):
    """u_t = L u + f(t, u) (latent.py:266-292): H = [[E1 - (J_x + L) E0, -E0], [B E0, 0]], shift J_x m_at - f."""

    semilinear = True
    _linearize = staticmethod(white.SemiLinearWhiteNoiseEK1._linearize)

    def solve_marginals(self, pde, *, num_steps=None):
        raise TypeError("solve_marginals keeps the loop on the device and needs a linear PDE; use solve()")
