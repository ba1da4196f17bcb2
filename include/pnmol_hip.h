/* pnmol_hip.h -- C ABI of the MI355X-native PNMOL white-noise EK1 filter step.
 *
 * The reference (schmidtjonathan/pnmol-experiments) is pure Python/JAX and has no
 * FFI of its own; its boundary for this path is the Python class
 * contract `pnmol.white.LinearWhiteNoiseEK1.{initialize,attempt_step,solve}`.  This header
 * is the C boundary a binding for that contract binds to; every entry point names the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - return 0 = ok, <0 = error: -1 bad argument, -2 HIP error, -3 innovation matrix not
 *     positive definite (see `pnmol_last_error`), -4 out of memory.
 *   - caller owns every host buffer; the library owns all device memory behind the
 *     opaque handles.  Matrices are row-major.
 *   - state vectors / covariances crossing this boundary use the reference's
 *     F-flattened order (`mean.reshape(-1, order="F")`, white.py:104): index j*n + i is
 *     derivative i at mesh point j.  `mean_nd` buffers are (n, d) row-major like
 *     `state.y.mean`.  Everything is in the NON-preconditioned ("raw") coordinates the
 *     reference's `PDEFilterState` carries.
 *   - one ctx <-> one device <-> one HIP stream.  A ctx is not thread-safe; different
 *     ctxs are independent (one thread or process per GPU).
 *   - Lifetimes: a handle keeps its parent alive.  Destroy in the order states -> filter(s) -> ctx.  A destroy call on
 *     a parent that still has live children does NOTHING and returns -1 (`pnmol_last_error` says how many children are
 *     left): `pnmol_filter_destroy` while any `pnmol_state` of the filter lives, `pnmol_ctx_destroy` while any
 *     `pnmol_filter` / `pnmol_sqrt_filter` of the ctx lives, `pnmol_state_destroy` of the target of an unfinished
 *     `pnmol_filter_steps_begin`.  The handle stays valid after a refused destroy; call it again once the children are gone.
 *   - `pnmol_abi_version()` = 3 (1: before `pnmol_filter_desc.dtype`, the lifetime rule and `pnmol_filter_sweep_layout`;
 *     2: `pnmol_sqrt_filter_create` refused `dtype = 1`, which now selects the fp32 QR of include/pnmol_sqrt.h).
 *     Zero-initialise `pnmol_filter_desc`: unknown `dtype` values are rejected with -1.
 *   - dtype: fp64 (the reference runs with jax_enable_x64, src/pnmol/__init__.py:9-11); `pnmol_filter_desc.dtype = 1`
 *     keeps the covariance and its bulk kernels in fp32 (build-side option).
 */
#ifndef PNMOL_HIP_H
#define PNMOL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pnmol_ctx pnmol_ctx;
typedef struct pnmol_filter pnmol_filter; /* model + workspace: L, B, E, R, Gamma, nu       */
typedef struct pnmol_state pnmol_state;   /* device-resident (mean, covariance, t)          */

/* library / device ----------------------------------------------------------------- */
int pnmol_abi_version(void);
int pnmol_device_count(int* count);
int pnmol_ctx_create(int device, pnmol_ctx** out);
int pnmol_ctx_destroy(pnmol_ctx* ctx); /* -1, nothing freed, while filters of ctx are alive (see Lifetimes) */
int pnmol_ctx_synchronize(pnmol_ctx* ctx);
const char* pnmol_last_error(pnmol_ctx* ctx);

/* cold-path assembly on the device ----------------------------------------------------------
 * Batched kernel finite-difference stencils, `discretize.fd_coefficients` under `jax.vmap` (discretize.py:60,75-80,177-201):
 * for every mesh point p,  weights_p = (gram_p)^-1 lk_p  (gram_p = k(X_p, X_p) + nugget I, (s,s); lk_p = L k(x_p, X_p), (s)),
 * uncertainty_p = llk_p - weights_p . lk_p  (llk_p = L L k(x_p, x_p)).  1 <= s <= 16.  LU with partial pivoting per point. */
int pnmol_fd_solve_batched(pnmol_ctx* ctx, const double* gram_nss, const double* lk_ns, const double* llk_n, int N, int s,
                           double* weights_ns, double* uncertainty_n);
/* Gamma = chol(K): `jnp.linalg.cholesky(spatial_kernel(X, X.T))` (white.py:84-85, latent.py:139) with the step's own sweep
 * kernels.  A (n,n) symmetric positive definite, row-major; L (n,n) lower triangular (upper part zeroed).
 * -3: not positive definite (the failing pivot is in `pnmol_last_error`). */
int pnmol_cholesky_lower(pnmol_ctx* ctx, const double* A_nn, int n, double* L_nn);

/* problem description = the attributes `attempt_step` reads from `pde`
 * (white.py:96-146, :169-186; pde/mixins.py:19-59) and from the solver
 * (`num_derivatives`, pdefilter.py:37-70; Gamma = chol(spatial_kernel(X, X.T)),
 * white.py:82-94). */
typedef struct pnmol_filter_desc {
    int d;                 /* mesh points = pde.L.shape[0]                            */
    int num_derivatives;   /* nu; n = nu + 1 in {2,3,4}                               */
    int nB;                /* rows of pde.B                                           */
    const double* L;       /* (d,d_state)  pde.L                                      */
    const double* B;       /* (nB,d_state) pde.B                                      */
    const double* E_sqrtm; /* (d,d)  pde.E_sqrtm                                      */
    const double* R_sqrtm; /* (nB,nB) pde.R_sqrtm                                     */
    const double* Gamma;   /* (d,d) lower; iwp.wp_diffusion_sqrtm (base/iwp.py:10)    */
    int d_state;           /* 0 or d: white-noise model.  2d: latent-force model (latent.py:11-292), whose state is
                              [u; eps] = two stacked IWPs (base/stacked_ssm.py): then L is (d, 2d) = [L, I]
                              (H_ode = E1 - L E0 - E0_eps, latent.py:253-257), B is (nB, 2d) = [B, 0], Gamma is
                              (2d, 2d) = blockdiag(chol K, E_sqrtm) (latent.py:136-153), E_sqrtm/R_sqrtm are the
                              measurement noise factors (zero: update_sqrt_no_meascov, latent.py:197).  State
                              buffers are then (n, 2d) / (2D, 2D) in the reference's glued order (latent.py:163-175). */
    int dtype;             /* 0: fp64 (the reference's arithmetic, src/pnmol/__init__.py:9-11).
                              1: fp32 covariance (BASELINE config 5's "fp32 with tolerance study"): the state
                              covariance, the predicted covariance and Q live in HBM as fp32, and the bulk of the step --
                              P- = A P A^T + Q, the stencil gather P- H^T, the down-date P = P- - W W^T (fp32 MFMA,
                              v_mfma_f32_16x16x4_f32) -- runs on them; the stencil weights, S = H P- H^T + R, its Cholesky
                              sweep (Ls, W, r), the mean and every scalar stay fp64.  Buffers crossing this boundary are
                              double either way.  Needs num_derivatives <= 2.  What it costs in accuracy: DESIGN.md
                              section 11 (tolerance study). */
    const double* K;       /* optional (d_state,d_state): the Gram matrix Gamma Gamma^T = spatial_kernel(X, X^T) of white.py:84-85
                              (base/iwp.py:49-52 uses it as Ql Ql^T = Q1 (x) K).  NULL: the library forms Gamma Gamma^T itself,
                              an O(d^3) scalar loop on the host (seconds at d = 4096); a caller that has the Gram matrix anyway
                              passes it.  Ignored by pnmol_sqrt_filter_create (the QR form works on Gamma itself). */
} pnmol_filter_desc;

int pnmol_filter_create(pnmol_ctx* ctx, const pnmol_filter_desc* desc, pnmol_filter** out);
int pnmol_filter_destroy(pnmol_filter* f); /* -1, nothing freed, while states of f are alive (see Lifetimes) */
/* Which sweep launch this filter's steps use (build-side diagnostic; no reference counterpart): *kernel = 0 per-panel
 * launches, 1 left-looking dataflow kernel (wide matrices: N = 1024, 2-d meshes), 2 register-resident right-looking kernel
 * (d + nB <= 544); *xcd_home = the XCD its critical workgroups are placed on (XCD-local layout: the first such filter alive on
 * a device, or PNMOL_HIP_SWEEP_XL=1) or -1 (spread layout).  Timings of the two layouts differ; benchmarks report this. */
int pnmol_filter_sweep_layout(const pnmol_filter* f, int* kernel, int* xcd_home);

/* Step-invariant part of `estimate_error` (white.py:153-162) for step size dt:
 * Sq = H (Ql Ql^T) H^T + E E^T depends only on dt for a linear PDE.  The caller passes
 * Sq^-1 (m,m) and diag(Sq) (m), m = d + nB; the per-step part z^T Sq^-1 z runs on device.
 * Without it, the error estimate of a step with this dt is reported as NaN. */
int pnmol_filter_set_error_model(pnmol_filter* f, double dt, const double* Sq_inv,
                                 const double* Sq_diag);
/* The same on the device, from the operator currently set (L, or J_x + L after `pnmol_filter_set_operator`):
 * Sq is the innovation matrix of a filter whose predicted covariance is Ql Ql^T = Q1 (x) K, so the step's own
 * kernels factorise it ([Sq; I] -> [Lq; Lq^-T], Sq^-1 = Lq^-T Lq^-1); diag(Sq) is kept for the error vector.  No host
 * O(m^3) work per new dt / per semilinear step. */
int pnmol_filter_prepare_error_model(pnmol_filter* f, double dt);

/* Semilinear EK1 (`SemiLinearWhiteNoiseEK1.evaluate_ode`, white.py:189-208): the measurement rows become
 * H_ode = E1 - (J_x + L) E0 with shift b = J_x m_at - f(t, m_at), re-linearised at every step.
 * `pnmol_filter_predict_mean` returns m_at = E0 P m^- (predicted derivative-0 mean, raw coordinates) for a step
 * of size dt from `in`; the caller evaluates f and df there and passes M = J_x + L (d,d_state) and shift (d) to
 * `pnmol_filter_set_operator` (NULL shift = zeros), which replaces the stencil rows used by the following
 * `pnmol_filter_step(s)` calls.  The boundary rows B are kept. */
int pnmol_filter_predict_mean(pnmol_filter* f, const pnmol_state* in, double dt, double* m_at_d);
int pnmol_filter_set_operator(pnmol_filter* f, const double* M_dds, const double* shift_d);
/* The same for a POINTWISE nonlinearity, whose Jacobian is diagonal (`df_diagonal` of the reference's problem classes,
 * pde/problems.py; spruce budworm: pde/examples.py:292-341): M = L + diag(jdiag) with the L given at creation.  d + d numbers
 * cross the bus instead of a dense (d, d_state) matrix, the stencil rows are patched on the device (no host scan of M, no
 * stream synchronisation, captured graphs stay valid).  -1 if a row of L has no diagonal entry. */
int pnmol_filter_set_operator_diagonal(pnmol_filter* f, const double* jdiag_d, const double* shift_d);

/* states ----------------------------------------------------------------------------- */
int pnmol_state_create(pnmol_filter* f, pnmol_state** out);
int pnmol_state_destroy(pnmol_state* s); /* -1 for the target of an unfinished pnmol_filter_steps_begin */
int pnmol_state_clone(const pnmol_state* s, pnmol_state** out); /* reject/retry, pdefilter.py:192-223 */
/* upload `PDEFilterState(t, y=(mean, cov))`; cov = cov_sqrtm @ cov_sqrtm.T (base/rv.py:12-14) */
int pnmol_state_set(pnmol_state* s, double t, const double* mean_nd, const double* cov_DD);
/* the same from a square root: any C (D,D) with C C^T = cov, e.g. the reference's `cov_sqrtm`; C C^T is formed on the
 * device (no O(D^3) host work).  Used by `initialize` (white.py:12-80), whose factor comes from `pnmol_sqrt_update`. */
int pnmol_state_set_sqrtm(pnmol_state* s, double t, const double* mean_nd, const double* cov_sqrtm_DD);
int pnmol_state_get_time(const pnmol_state* s, double* t);
int pnmol_state_get_mean(const pnmol_state* s, double* mean_nd);       /* (n,d)            */
int pnmol_state_get_cov(const pnmol_state* s, double* cov_DD);         /* (D,D) F-order    */
/* The lower-triangular Cholesky factor C of the covariance, C C^T = cov, (D,D) in the F-flattened order -- the
 * canonical representative (positive diagonal) of what the reference carries as `cov_sqrtm` (base/rv.py:9-14; its
 * QR factors differ from it by column signs only).  Computed on the device by the step's sweep kernel; a direction
 * whose pivot falls below 1e-13 of its diagonal entry (noise-free Dirichlet node, numerically deterministic
 * combinations) gives a zero column: diag(C C^T) stays accurate (1e-6 relative or better), the off-diagonal entries of such a direction j
 * are lost, |(C C^T - cov)_ij| <= sqrt(c_ii * 1e-13 c_jj).  Never fails on indefiniteness (the covariance of the
 * recursion is PSD only up to rounding); -3 only for NaN. */
int pnmol_state_get_cov_sqrtm(const pnmol_state* s, double* C_DD);
/* diag(cov) as (n,d): what experiments/figure1.py:76-80 reads out (`stds**2`)            */
int pnmol_state_get_marginal_var(const pnmol_state* s, double* var_nd);

/* one step ----------------------------------------------------------------------------- */
typedef struct pnmol_step_out {
    double t_new;                   /* state.t + dt                                       */
    double diffusion_squared_local; /* white.py:125-128 formula with the Cholesky factor
                                       of S (positive diagonal): |Ls^-T z|^2 / m          */
    double sigma2_whitened;         /* z^T S^-1 z / m (the quasi-MLE the comment intends) */
    double error_sigma2;            /* z^T Sq^-1 z / m of estimate_error (NaN if no model)*/
    int info;                       /* -1 ok, else index of first non-positive pivot      */
} pnmol_step_out;

/* `_WhiteNoiseEK1Base.attempt_step` with `LinearWhiteNoiseEK1.evaluate_ode`
 * (white.py:96-146, :169-186).  `in` is not modified; `out` may not alias `in`.
 * `error_estimate_d` (d) optional: dt * sqrt(diag Sq) * sigma (white.py:117-129). */
int pnmol_filter_step(pnmol_filter* f, const pnmol_state* in, double dt, pnmol_state* out,
                      pnmol_step_out* info, double* error_estimate_d);

/* k steps of constant dt with no host synchronisation in between -- the loop body of
 * `PDEFilter.solution_generator` under `step.Constant` (pdefilter.py:140-160,
 * odetools/step.py:30-55).  `s` is advanced in place.  Optional outputs, each written
 * once after the last step: means_kd / stds_kd (k,d) = `sol.mean[1:, 0]` and
 * sqrt(diag(cov) E0^T) per step (figure1.py:76-80, uncalibrated); info_k (k entries). */
int pnmol_filter_steps(pnmol_filter* f, pnmol_state* s, int k, double dt, double* means_kd,
                       double* stds_kd, pnmol_step_out* info_k);

/* The same loop split in two, so that one host thread can keep several contexts (several problems on one GPU, or
 * several GPUs) busy: `_begin` enqueues the k steps and the read-out copies and returns without waiting, `_end`
 * waits for that context's stream and fills the caller's buffers.  One `_begin` may be outstanding per filter. */
int pnmol_filter_steps_begin(pnmol_filter* f, pnmol_state* s, int k, double dt);
int pnmol_filter_steps_end(pnmol_filter* f, pnmol_state* s, double* means_kd, double* stds_kd,
                           pnmol_step_out* info_k);

/* Optional: do the one-off host work of a following `pnmol_filter_steps(f, s, k, dt, ...)` now
 * (output buffers, capture + instantiation of the hipGraphs the step loop is replayed from). */
int pnmol_filter_prepare_steps(pnmol_filter* f, pnmol_state* s, int k, double dt);

/* timing hooks for bench.py: HIP events on the ctx stream around the last `steps` call */
int pnmol_filter_last_steps_ms(pnmol_filter* f, float* ms);

/* debugging / tests: copy an internal device buffer to the host.
 * which: 0 = predicted covariance (Dp*Dp, derivative-major padded), 1 = G work matrix,
 * 2 = F factor matrix [Ls; W; r^T], 3 = predicted mean (Dp), 4 = z (mp).  `count` doubles. */
int pnmol_filter_debug_read(pnmol_filter* f, int which, double* dst, long count);
/* Test hook: fills the buffers the sweep kernels hand data over through (F, the L_jj^-1 tiles, the feed / scratch tiles) with
 * NaN, as stale contents of an earlier launch.  A following step must give the same bits as without it
 * (tests/test_gpu_parity.py::test_stale_hand_over_buffers_are_never_read): every reader takes published data only. */
int pnmol_filter_debug_poison(pnmol_filter* f);
int pnmol_filter_dims(const pnmol_filter* f, int* d, int* n, int* m, int* dp, int* mp);

#ifdef __cplusplus
}
#endif
#endif /* PNMOL_HIP_H */
