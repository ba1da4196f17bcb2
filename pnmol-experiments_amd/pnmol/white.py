"""White-noise EK1 PDE filters (reference: src/pnmol/white.py).

`attempt_step` and the constant-step loop run on the GPU through libpnmol_hip.so in the
covariance (Cholesky) form that the reference's own tests prove equivalent to its QR form
(tests/test_base/test_sqrt.py:48-78).  There is no CPU implementation of the step in this package.

Documented deviations from the reference's numbers:
  * `cov_sqrtm` is the lower-triangular Cholesky factor of the same covariance (device Cholesky); the reference's QR
    factor differs by column signs, and arbitrarily where a pivot is exactly zero (noise-free Dirichlet nodes).
  * `diffusion_squared_local`: the reference evaluates |R1^-1 z|^2 / m with the QR factor R1 of the
    innovation (white.py:125), whose row signs LAPACK chooses from the data; here the same formula is
    evaluated with the Cholesky factor (positive diagonal).  `last_step_info.sigma2_whitened` holds
    z^T S^-1 z / m.
"""

import os

import numpy as np
import scipy.linalg
import scipy.sparse

from . import _hip, pdefilter
from .base import iwp, rv
from .odetools import step as _step


class _WhiteNoiseEK1Base(pdefilter.PDEFilter):
    _device_filter = None
    _device_pde = None
    _error_models = None
    last_step_info = None

    # ------------------------------------------------------------------ cold path
    def initialize_iwp(self, pde):
        """Gamma = chol(k(X, X^T)); IWP prior; E0, E1 (white.py:82-94)."""
        X = pde.mesh_spatial.points
        gram = self.spatial_kernel(X, X.T)
        self._last_gram = gram          # (handed to the device filter as it is: no Gamma Gamma^T product, `_bind`)
        diffusion_state_sqrtm = self._cholesky(gram)
        prior = iwp.IntegratedWienerTransition(num_derivatives=self.num_derivatives,
                                               wiener_process_dimension=pde.y0.shape[0],
                                               wp_diffusion_sqrtm=diffusion_state_sqrtm)
        return prior, prior.projection_matrix(0), prior.projection_matrix(1), diffusion_state_sqrtm

    # Gamma = chol(K) on the device (`pnmol_cholesky_lower`) from d >= CHOLESKY_ON_DEVICE_FROM mesh points on
    # (PNMOL_CHOL_ON_DEVICE=0/1 forces it off / on); LAPACK on the host below that (the factors agree to rounding:
    # tests/test_gpu_assembly.py)
    CHOLESKY_ON_DEVICE_FROM = 2048

    def _cholesky(self, gram):
        mode = os.environ.get("PNMOL_CHOL_ON_DEVICE", "")
        on_device = gram.shape[0] >= self.CHOLESKY_ON_DEVICE_FROM if mode not in ("0", "1") else mode == "1"
        if on_device:
            return (self._context or _hip.Context.default()).cholesky(gram)
        return np.linalg.cholesky(gram)

    _last_gram = None
    _context = None   # set to an `_hip.Context` to run this solver on its own device / stream
    # Build-side option (the reference is fp64 throughout, src/pnmol/__init__.py:9-11):
    # "f32" keeps the covariance and its bulk kernels in fp32 (include/pnmol_hip.h, pnmol_filter_desc.dtype;
    # BASELINE config 5).  Accuracy: DESIGN.md section 11.
    dtype = "f64"
    # The fp32 covariance diverges without an error flag for num_derivatives >= 2 (mean errors of 1e10
    # after 40 steps at N=256): refused unless a study asks for exactly that.
    allow_unstable_f32 = False

    def _bind(self, pde, gamma):
        if self.dtype == "f32" and self.num_derivatives > 1 and not self.allow_unstable_f32:
            raise ValueError('dtype="f32" keeps the covariance in single precision, which is only accurate for '
                             'num_derivatives = 1 (DESIGN.md section 11); use the fp64 path, or pnmol.sqrtform with dtype="f32"')
        ctx = self._context or _hip.Context.default()
        gram = self._last_gram if (self._last_gram is not None and self._last_gram.shape == gamma.shape) else gamma @ gamma.T
        self._device_filter = _hip.Filter(ctx, L=pde.L, B=pde.B, E_sqrtm=pde.E_sqrtm, R_sqrtm=pde.R_sqrtm, Gamma=gamma,
                                          num_derivatives=self.num_derivatives, dtype=self.dtype, K=gram)
        self._device_pde = pde
        self._gram = gram
        self._error_models = {}

    # True: the reference's own two `update_sqrt` calls, on the device (below); False: closed form on the host (O(d^3) LAPACK);
    # None (default; PNMOL_INIT_ON_DEVICE=0/1 overrides): on the device from n d >= 4096 on, where the host form takes
    # seconds (64x64 mesh: 18 s against 10 s), closed form below that (its rounding-level entries are what the
    # factor-level tests are tuned to; both match the oracle at the north-star tolerances)
    initialize_on_device = {"0": False, "1": True}.get(os.environ.get("PNMOL_INIT_ON_DEVICE", ""), None)
    INIT_ON_DEVICE_FROM = 4096

    def _initialize_on_device(self, pde, gamma):
        """white.py:12-80 as written: prior kron(Gamma, c I) -> update_sqrt on y0 (nugget 1e-10) -> update_sqrt on the
        PDE/BC residual at t0, both QRs on the device (include/pnmol_sqrt.h); the factor goes to the device state, which
        forms C C^T itself (`pnmol_state_set_sqrtm`).  No O(d^3) host work."""
        from .base import sqrt as dsqrt

        ctx = self._device_filter.ctx
        n, d, nB = self.num_derivatives + 1, pde.L.shape[0], pde.B.shape[0]
        C0_raw = np.zeros((n * d, n * d))               # = np.kron(gamma, c I_n) (white.py:21-24), without kron's temporaries
        for a in range(n):
            C0_raw[a::n, a::n] = self.diffuse_prior_scale * gamma
        C0_y0, k_y0, _ = dsqrt.update_sqrt(self.E0, C0_raw, 1e-10 * np.eye(d), ctx=ctx)
        m0_y0 = k_y0 @ pde.y0
        M, shift = self._linearize(pde, m0_y0[0::n], pde.t0)
        H = np.zeros((d + nB, n * d))
        H[:d, 0::n] = -M
        H[np.arange(d), np.arange(d) * n + 1] += 1.0
        H[d:, 0::n] = pde.B
        z = H @ m0_y0 + np.concatenate([shift, np.zeros(nB)])
        E = np.zeros((d + nB, d + nB))
        E[:d, :d], E[d:, d:] = pde.E_sqrtm, pde.R_sqrtm
        C0, k, _ = dsqrt.update_sqrt(H, C0_y0, E + 1e-10 * np.eye(d + nB), ctx=ctx)
        mean = (m0_y0 - k @ z).reshape((n, d), order="F")
        dev = self._device_filter.new_state()
        dev.set_sqrtm(pde.t0, mean, C0)
        return mean, dev

    def initialize(self, pde):
        """Initial state: prior conditioned on y0 and on the PDE/BC residual at t0 (white.py:12-80).

        The two updates of the reference have observation noise ~1e-20, which a plain covariance
        update cannot resolve; here they are evaluated block-wise in closed form (the prior
        Gamma Gamma^T (x) c^2 I is Kronecker, the derivative blocks decouple), every difference
        written so that it does not cancel.  O(d^3) on the host, once per solve.
        """
        self.iwp, self.E0, self.E1, gamma = self.initialize_iwp(pde)
        self._bind(pde, gamma)
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        on_device = self.initialize_on_device
        if on_device is None:
            on_device = n * d >= self.INIT_ON_DEVICE_FROM
        if on_device:
            mean, dev = self._initialize_on_device(pde, gamma)
            return pdefilter.PDEFilterState(t=pde.t0, y=rv.DeviceMultivariateNormal(mean, dev), error_estimate=None,
                                            reference_state=None, diffusion_squared_local=[])
        mean, blocks = self._initial_moments(pde)
        cov = np.zeros((n * d, n * d))
        for (a, b), blk in blocks.items():
            cov[a::n, b::n] = blk
        dev = self._device_filter.new_state()
        dev.set(pde.t0, mean, cov)
        return pdefilter.PDEFilterState(t=pde.t0, y=rv.DeviceMultivariateNormal(mean, dev), error_estimate=None,
                                        reference_state=None, diffusion_squared_local=[])

    def _initial_moments(self, pde):
        n, d = self.num_derivatives + 1, pde.L.shape[0]
        eps2 = 1e-10 ** 2                                       # (1e-10 I)(1e-10 I)^T, white.py:33-39
        Kc = self.diffuse_prior_scale ** 2 * self._gram         # C0 C0^T = K (x) c^2 I, white.py:21-24
        # update 1: observe E0 x = y0.  V0 = Kc - Kc S1^-1 Kc = eps2 Kc S1^-1 (no cancellation)
        S1 = scipy.linalg.cho_factor(Kc + eps2 * np.eye(d), lower=True)
        G1 = scipy.linalg.cho_solve(S1, Kc).T                   # Kc S1^-1
        V0 = eps2 * 0.5 * (G1 + G1.T)
        m0 = G1 @ pde.y0
        # update 2 (white.py:42-58): rows [E1 - M E0 ; B E0], noise (E_bc + 1e-10 I)(...)^T, block diagonal
        M, shift = self._linearize(pde, m0, pde.t0)
        nB = pde.B.shape[0]
        En = pde.E_sqrtm + 1e-10 * np.eye(d)
        Rn = pde.R_sqrtm + 1e-10 * np.eye(nB)
        Re, Rb = En @ En.T, Rn @ Rn.T
        # (a) boundary rows involve derivative 0 only
        if nB > 0:
            Sb = pde.B @ V0 @ pde.B.T + Rb
            Gb = np.linalg.solve(Sb, pde.B @ V0).T              # V0 B^T Sb^-1
            m0 = m0 - Gb @ (pde.B @ m0)
            V0 = V0 - Gb @ (pde.B @ V0)
            V0 = 0.5 * (V0 + V0.T)
        # (b) PDE rows: z = x1 - M x0 + shift, prior blockdiag(V0, Kc);  Sp = Kc + T, T tiny
        T = M @ V0 @ M.T + Re
        Sp = scipy.linalg.cho_factor(Kc + T, lower=True)
        z = -M @ m0 + shift
        Spz = scipy.linalg.cho_solve(Sp, z)
        SpT = scipy.linalg.cho_solve(Sp, T)                     # Sp^-1 T
        V0Mt = V0 @ M.T
        mean = np.zeros((n, d))
        mean[0] = m0 + V0Mt @ Spz
        mean[1] = -(z - T @ Spz)                                # -Kc Sp^-1 z
        P00 = V0 - V0Mt @ scipy.linalg.cho_solve(Sp, V0Mt.T)
        P01 = V0Mt - V0Mt @ SpT                                 # V0 M^T Sp^-1 Kc
        P11 = T - T @ SpT                                       # Kc - Kc Sp^-1 Kc
        blocks = {(0, 0): 0.5 * (P00 + P00.T), (0, 1): P01, (1, 0): P01.T, (1, 1): 0.5 * (P11 + P11.T)}
        for q in range(2, n):
            blocks[(q, q)] = Kc
        return mean, blocks

    # ------------------------------------------------------------------ step-invariant error model
    def _error_model(self, pde, dt, M=None):
        """Sq^-1 and diag(Sq) of `estimate_error` (white.py:153-162) in the Nordsieck frame of dt.
        `M` is the operator of the linearisation (L for linear problems: cached per dt; J_x + L otherwise)."""
        key = float(dt)
        cache = M is None
        M = pde.L if M is None else M
        if not cache or self._error_models is None or key not in self._error_models:
            d, nB = pde.L.shape[0], pde.B.shape[0]
            s, _ = self.iwp.nordsieck_preconditioner_1d_raw(dt)
            n = self.num_derivatives + 1
            Q1 = np.flip(scipy.linalg.hilbert(n))
            Hv = scipy.sparse.csr_matrix(np.vstack((-M, pde.B)))
            F1K = np.vstack((self._gram, np.zeros((nB, d))))    # [I;0] K
            HvK = Hv @ self._gram
            cross = (Hv @ F1K.T).T                              # F1 K Hv^T
            Sq = Q1[1, 1] * s[1] ** 2 * np.hstack((F1K, np.zeros((d + nB, nB))))
            Sq = Sq + Q1[0, 1] * s[0] * s[1] * (cross + cross.T) + Q1[0, 0] * s[0] ** 2 * (Hv @ HvK.T).T
            Ebc = scipy.linalg.block_diag(pde.E_sqrtm, pde.R_sqrtm)
            Sq = Sq + Ebc @ Ebc.T
            Sq = 0.5 * (Sq + Sq.T)
            inv = scipy.linalg.cho_solve(scipy.linalg.cho_factor(Sq, lower=True), np.eye(d + nB))
            model = (0.5 * (inv + inv.T), np.diag(Sq).copy())
            if not cache:
                return model
            self._error_models = {key: model}
        return self._error_models[key]

    error_model_on_host = False   # True: Sq^-1 from the host (`_error_model`, an O(m^3) solve per new dt) instead of the GPU

    def _ensure_error_model(self, pde, dt):
        if self._device_filter.error_model_dt != float(dt):
            if self.error_model_on_host:
                self._device_filter.set_error_model(dt, *self._error_model(pde, dt))
            else:
                self._device_filter.prepare_error_model(dt)   # Sq factorised by the step's own kernels

    # ------------------------------------------------------------------ hot path
    def _device_state_of(self, state, pde):
        if self._device_filter is None or self._device_pde is not pde:
            raise RuntimeError("call initialize(pde) before attempt_step (the device model is bound there)")
        y = state.y
        if isinstance(y, rv.DeviceMultivariateNormal) and y.device_state.filter is self._device_filter:
            return y.device_state
        dev = self._device_filter.new_state()                   # host-side state: upload mean and C C^T
        C = np.asarray(y.cov_sqrtm)
        dev.set(state.t, np.asarray(y.mean), C @ C.T)
        return dev

    semilinear = False

    @staticmethod
    def _jacobian_diagonal(pde, m_at, t):
        return None

    def attempt_step(self, state, dt, pde):
        """One predict + update + calibrate step on the GPU (white.py:96-146); `state` is not modified."""
        dev_in = self._device_state_of(state, pde)
        if self.semilinear:
            # EK1 linearisation at the predicted mean (white.py:192-208): f and df are host callables, so the
            # predicted point comes back once per step; the new stencil rows and shift go to the device
            flt = self._device_filter
            m_at = flt.predict_mean(dev_in, dt)
            jdiag = self._jacobian_diagonal(pde, m_at, state.t + dt)
            if jdiag is not None and not self.error_model_on_host:
                # pointwise nonlinearity (`df_diagonal`): 2 d numbers go to the device, which patches its stencil rows itself
                fx = np.asarray(pde.f(state.t + dt, m_at), dtype=np.float64)
                flt.set_operator_diagonal(jdiag, jdiag * m_at - fx)
                flt.prepare_error_model(dt)
            else:
                M, shift = self._linearize(pde, m_at, state.t + dt)
                flt.set_operator(M, shift)
                if self.error_model_on_host:
                    flt.set_error_model(dt, *self._error_model(pde, dt, M=M))
                else:
                    flt.prepare_error_model(dt)
        else:
            self._ensure_error_model(pde, dt)
        dev_out, info, error = self._device_filter.step(dev_in, dt)
        self.last_step_info = info
        m_new = dev_out.mean()
        new_state = pdefilter.PDEFilterState(
            t=state.t + dt, error_estimate=error, reference_state=np.abs(m_new[0]),
            y=rv.DeviceMultivariateNormal(m_new, dev_out), diffusion_squared_local=info.diffusion_squared_local)
        return new_state, dict(num_f_evaluations=1, num_df_evaluations=1)

    def solve_marginals(self, pde, *, num_steps=None):
        """Constant-step solve that keeps everything on the device: returns
        (t (T+1,), means (T+1,d), stds (T+1,d), diffusion_squared_local (T,), final PDEFilterState).

        `means`/`stds` are `sol.mean[:, 0]` and sqrt(diag(cov) E0^T) of the reference's read-out
        (experiments/figure1.py:76-80).  Steps follow the t-accumulation of the reference's loop
        (pdefilter.py:140-160, :220-223), including a runt final step when sum(dt) lands short of tmax.
        """
        if not isinstance(self.steprule, _step.Constant):
            raise TypeError("solve_marginals needs the Constant step rule")
        state = self.initialize(pde)
        dev = state.y.device_state
        dt0 = self.steprule.first_dt(pde)
        ts, dts, t, dt = [pde.t0], [], pde.t0, dt0
        while t < pde.tmax and (num_steps is None or len(dts) < num_steps):
            dts.append(dt)
            t = t + dt
            ts.append(t)
            dt = min(dt0, pde.tmax - t)
        d = pde.L.shape[0]                                      # (the latent-force state carries eps behind u)
        means = [state.y.mean[0][:d]]
        stds = [np.sqrt(np.maximum(state.y.marginal_var[0][:d], 0.0))]
        sig = []
        i = 0
        while i < len(dts):                                     # runs of equal dt -> one device call each
            j = i
            while j < len(dts) and dts[j] == dts[i]:
                j += 1
            self._ensure_error_model(pde, dts[i])
            mk, sk, infos = self._device_filter.steps(dev, j - i, dts[i])
            means.extend(mk), stds.extend(sk)
            sig.extend(o.diffusion_squared_local for o in infos)
            i = j
        m_final = dev.mean()
        final = pdefilter.PDEFilterState(t=ts[-1], y=rv.DeviceMultivariateNormal(m_final, dev), error_estimate=None,
                                         reference_state=np.abs(m_final[0][:d]),
                                         diffusion_squared_local=sig[-1] if sig else [])
        return np.array(ts), np.array(means), np.array(stds), np.array(sig), final


class LinearWhiteNoiseEK1(_WhiteNoiseEK1Base):
    """EK1 for linear PDEs u_t = L u (white.py:169-186): H = [E1 - L E0 ; B E0], no shift."""

    @staticmethod
    def _linearize(pde, m_at, t):
        return pde.L, np.zeros(pde.L.shape[0])


class SemiLinearWhiteNoiseEK1(_WhiteNoiseEK1Base):
    """EK1 for semilinear PDEs u_t = L u + f(t, u) (white.py:189-208): H = [E1 - (J_x + L) E0 ; B E0] and
    z = H m + [J_x m_at - f(t, m_at); 0], re-linearised at the predicted mean of every step."""

    semilinear = True

    @staticmethod
    def _linearize(pde, m_at, t):
        fx = np.asarray(pde.f(t, m_at), dtype=np.float64)
        Jx = np.asarray(pde.df(t, m_at), dtype=np.float64)
        return pde.L + Jx, Jx @ m_at - fx

    @staticmethod
    def _jacobian_diagonal(pde, m_at, t):
        """diag(J_x) where the problem says its Jacobian is diagonal (`df_diagonal`, pde/problems.py:11-42), else None."""
        dfd = getattr(pde, "df_diagonal", None)
        if dfd is None or pde.L.shape[0] != pde.L.shape[1]:
            return None
        return np.asarray(dfd(t, m_at), dtype=np.float64)

    def solve_marginals(self, pde, *, num_steps=None):
        raise TypeError("solve_marginals keeps the loop on the device and needs a linear PDE; use solve()")
