"""The white-noise EK1 solvers with the step in square-root (QR) form on the GPU.

`pnmol.white.*` advance the covariance (the fast path).  The classes here advance the reference's own state
`MultivariateNormal(mean, cov_sqrtm)` by the reference's own sequence -- `propagate_cholesky_factor(A @ Cl, Ql)`, then
`update_sqrt(H, Clp, E)` (white.py:96-146, base/sqrt.py) -- with both QRs on the device (include/pnmol_sqrt.h).  Use them
when the factor itself is wanted or the covariance form's resolution is not enough; a step costs about 40x the
covariance form's flops.  `estimate_error` (white.py:153-162) is evaluated in square-root form too, when the step rule
uses it (Adaptive).  No CPU fallback.

`dtype = "f32"` on a solver of this module runs the QRs in fp32 (work matrices, Householder reflectors and trailing updates on
`v_mfma_f32_16x16x4_f32`; state, mean path and scalars fp64).  Unlike the fp32 covariance form it holds the north-star
tolerances at `num_derivatives = 2` and has no std floor (`tools/fp32_sqrt_model.py` predicts it,
`tests/test_gpu_sqrt_fp32.py` asserts it): a factor loses relative 6e-8 where a covariance loses 6e-8 of its largest entry.
Its std error grows with the mesh (7e-5 relative at N = 512 .. 768): the mode is specified up to 768 mesh points in 1-d with
`num_derivatives = 2`; at N = 1024 it is finite and the mean holds 1e-5, but the stds are at 2.6e-4.
"""

import numpy as np

import scipy.linalg

from pnmol import _hip, latent, pdefilter, white
from pnmol.base import rv
from pnmol.odetools import step as _step


class _SqrtFormMixin:
    _sqrt_filter = None
    _sqrt_last = None      # the state object whose factor is resident in the device filter

    _err_dt = None         # the dt the device holds the error model of (linear PDE: reused until dt changes)

    def initialize(self, pde):
        """The reference's own initialisation (white.py:12-80), its two `update_sqrt` calls on the device: prior
        kron(Gamma, c I), conditioned on y0 (nugget 1e-10), then on the PDE/BC residual at t0."""
        from pnmol.base import sqrt as dsqrt

        self.iwp, self.E0, self.E1, gamma = self.initialize_iwp(pde)
        self._gram = gamma @ gamma.T
        self._device_pde = pde
        ctx = self._context or _hip.Context.default()
        self._sqrt_filter = _hip.SqrtFilter(ctx, L=pde.L, B=pde.B, E_sqrtm=pde.E_sqrtm, R_sqrtm=pde.R_sqrtm, Gamma=gamma,
                                            num_derivatives=self.num_derivatives, dtype=self.dtype)
        n, d, nB = self.num_derivatives + 1, pde.L.shape[0], pde.B.shape[0]
        C0_raw = np.kron(gamma, self.diffuse_prior_scale * np.eye(n))
        C0_y0, k_y0, _ = dsqrt.update_sqrt(self.E0, C0_raw, 1e-10 * np.eye(d), ctx=ctx)
        m0_y0 = k_y0 @ pde.y0
        # evaluate_ode at t0 with p0 = E0, p1 = E1 (white.py:42-48): H = [E1 - M E0; B E0] written by index
        M, shift = self._linearize(pde, m0_y0[0::n], pde.t0)
        H = np.zeros((d + nB, n * d))
        H[:d, 0::n] = -M
        H[np.arange(d), np.arange(d) * n + 1] += 1.0
        H[d:, 0::n] = pde.B
        z = H @ m0_y0 + np.concatenate([shift, np.zeros(nB)])
        E = np.zeros((d + nB, d + nB))
        E[:d, :d], E[d:, d:] = pde.E_sqrtm, pde.R_sqrtm
        C0, k, _ = dsqrt.update_sqrt(H, C0_y0, E + 1e-10 * np.eye(d + nB), ctx=ctx)
        m0 = m0_y0 - k @ z
        state = pdefilter.PDEFilterState(t=pde.t0, y=rv.MultivariateNormal(m0.reshape((n, d), order="F"), C0),
                                         error_estimate=None, reference_state=None, diffusion_squared_local=[])
        self._sqrt_last, self._err_dt = None, None
        return state

    def _stack(self, M):
        return M                                   # white-noise model: rows of -(M E0 u)

    def _load(self, state, pde):
        if self._sqrt_filter is None or self._device_pde is not pde:
            raise RuntimeError("call initialize(pde) before attempt_step (the device model is bound there)")
        if self._sqrt_last is not state:       # rejected / foreign state: upload it
            self._sqrt_filter.set_state(state.t, np.asarray(state.y.mean), np.asarray(state.y.cov_sqrtm))

    def attempt_step(self, state, dt, pde):
        flt = self._sqrt_filter
        self._load(state, pde)
        if self.semilinear:
            m_at = flt.predict_mean(dt)
            M, shift = self._linearize(pde, m_at, state.t + dt)
            flt.set_operator(self._stack(M), shift)
            self._err_dt = None
        error = None
        if self._wants_error:
            if self._err_dt != dt:                 # estimate_error's Sq depends on (operator, dt) only (white.py:153-162)
                flt.prepare_error_model(dt)
                self._err_dt = dt
            info, error = flt.step(dt, want_error=True)
        else:
            info = flt.step(dt)
        self.last_step_info = info
        _, mean, C = flt.get_state()
        new = pdefilter.PDEFilterState(t=state.t + dt, y=rv.MultivariateNormal(mean, C), error_estimate=error,
                                       reference_state=self._reference_state(mean), diffusion_squared_local=info.diffusion_squared_local)
        self._sqrt_last = new
        return new, dict(num_f_evaluations=1, num_df_evaluations=1)

    @staticmethod
    def _reference_state(mean):
        return np.abs(mean[0])                     # white.py:141

    @property
    def _wants_error(self):
        return not isinstance(self.steprule, _step.Constant)   # Constant discards it (odetools/step.py:49-52)

    def solve_marginals(self, pde, *, num_steps=None):
        """As `pnmol.white.LinearWhiteNoiseEK1.solve_marginals`, the loop kept on the device in square-root form."""
        if self.semilinear:
            raise TypeError("solve_marginals keeps the loop on the device and needs a linear PDE; use solve()")
        state = self.initialize(pde)
        self._load(state, pde)
        flt, dt0 = self._sqrt_filter, self.steprule.first_dt(pde)
        ts, dts, t, dt = [pde.t0], [], pde.t0, dt0
        while t < pde.tmax and (num_steps is None or len(dts) < num_steps):
            dts.append(dt)
            t = t + dt
            ts.append(t)
            dt = min(dt0, pde.tmax - t)
        C0 = np.asarray(state.y.cov_sqrtm)
        d = pde.L.shape[0]                         # (the latent-force state carries eps behind u)
        means, stds, sig = [state.y.mean[0][:d]], [np.sqrt(np.einsum("ij,ij->i", C0, C0)[:: self.num_derivatives + 1][:d])], []
        i = 0
        while i < len(dts):
            j = i
            while j < len(dts) and dts[j] == dts[i]:
                j += 1
            mk, sk, infos = flt.steps(j - i, dts[i])
            means.extend(mk[:, :d]), stds.extend(sk[:, :d])
            sig.extend(o.diffusion_squared_local for o in infos)
            i = j
        _, mean, C = flt.get_state()
        final = pdefilter.PDEFilterState(t=ts[-1], y=rv.MultivariateNormal(mean, C), error_estimate=None,
                                         reference_state=self._reference_state(mean), diffusion_squared_local=sig[-1] if sig else [])
        self._sqrt_last = final
        return np.array(ts), np.array(means), np.array(stds), np.array(sig), final


class LinearWhiteNoiseEK1(_SqrtFormMixin, white.LinearWhiteNoiseEK1):
    """`pnmol.white.LinearWhiteNoiseEK1` (white.py:169-186), square-root form."""


class SemiLinearWhiteNoiseEK1(_SqrtFormMixin, white.SemiLinearWhiteNoiseEK1):
    """`pnmol.white.SemiLinearWhiteNoiseEK1` (white.py:189-208), square-root form."""


class _SqrtFormLatentMixin(_SqrtFormMixin):
    """Latent-force model (latent.py:11-292) in square-root form: the stack [u; eps] is one state of 2d components."""

    def _stack(self, M):
        return self._stacked_operator(M)           # rows of -(M E0 u + E0 eps), latent.py:253-257

    @staticmethod
    def _reference_state(mean):
        return None                                # latent.py:224

    _wants_error = False                           # no error estimate in this model (latent.py:224)

    def initialize(self, pde):
        """The reference's own initialisation (latent.py:20-134), its two `update_sqrt` calls (nuggets 1e-6) on the device."""
        from pnmol.base import sqrt as dsqrt
        from pnmol.base import stacked_ssm

        self.state_iwp, self.lf_iwp, self.E0, self.E1, gamma = self.initialize_iwp_latent(pde)
        self.iwp = self.state_iwp
        self.ssm = stacked_ssm.StackedSSM(processes=[self.state_iwp, self.lf_iwp])
        self._device_pde = pde
        ctx = self._context or _hip.Context.default()
        n, d, nB = self.num_derivatives + 1, pde.L.shape[0], pde.B.shape[0]
        D = n * d
        E = np.asarray(pde.E_sqrtm, dtype=np.float64)
        self._sqrt_filter = _hip.SqrtFilter(
            ctx, L=self._stacked_operator(pde.L), B=np.hstack((pde.B, np.zeros((nB, d)))), E_sqrtm=np.zeros((d, d)),
            R_sqrtm=np.zeros((nB, nB)), Gamma=scipy.linalg.block_diag(gamma, E), num_derivatives=self.num_derivatives,
            dtype=self.dtype)
        c0 = self.diffuse_prior_scale * np.eye(n)
        C_state, k_y0, _ = dsqrt.update_sqrt(self.E0, np.kron(gamma, c0), 1e-6 * np.eye(d), ctx=ctx)
        m_stack = np.concatenate([k_y0 @ pde.y0, np.zeros(D)])
        C_block = scipy.linalg.block_diag(C_state, np.kron(E, c0))
        # evaluate_ode at t0 (latent.py:88-98): H = [[E1 - M E0, -E0], [B E0, 0]] written by index
        M, shift = self._linearize(pde, m_stack[0:D:n], pde.t0)
        H = np.zeros((d + nB, 2 * D))
        H[:d, 0:D:n] = -M
        H[np.arange(d), np.arange(d) * n + 1] += 1.0
        H[np.arange(d), D + np.arange(d) * n] = -1.0
        H[d:, 0:D:n] = pde.B
        z = H @ m_stack + np.concatenate([shift, np.zeros(nB)])
        C0, k, _ = dsqrt.update_sqrt(H, C_block, 1e-6 * np.eye(d + nB), ctx=ctx)
        m0 = m_stack - k @ z
        mean = np.concatenate([m0[:D].reshape((n, d), order="F"), m0[D:].reshape((n, d), order="F")], axis=1)
        self._sqrt_last = None
        return pdefilter.PDEFilterState(t=pde.t0, y=rv.MultivariateNormal(mean, C0), error_estimate=None,
                                        reference_state=None, diffusion_squared_local=[])


class LinearLatentForceEK1(_SqrtFormLatentMixin, latent.LinearLatentForceEK1):
    """`pnmol.latent.LinearLatentForceEK1` (latent.py:241-263), square-root form."""


class SemiLinearLatentForceEK1(_SqrtFormLatentMixin, latent.SemiLinearLatentForceEK1):
    """`pnmol.latent.SemiLinearLatentForceEK1` (latent.py:266-292), square-root form."""
