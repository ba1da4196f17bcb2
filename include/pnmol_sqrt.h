/* pnmol_sqrt.h -- C ABI of the square-root (QR) primitives of the PNMOL filter on MI355X.
 *
 * Replaces, one entry point each, the functions of the reference's `src/pnmol/base/sqrt.py` (paths relative to the
 * reference root).  Same conventions as pnmol_hip.h: return 0 = ok, -1 bad argument, -2 HIP error, -4 out of memory;
 * caller owns the host buffers; matrices are row-major fp64; one ctx <-> one device <-> one stream.
 *
 * Sign convention: the reference's factors come out of LAPACK's Householder QR with data-dependent row signs; the
 * factors returned here are the unique representatives with a NON-NEGATIVE DIAGONAL (every product the reference
 * forms from them -- C C^T, Sl Sl^T, the gain K -- is sign-invariant).
 */
#ifndef PNMOL_SQRT_H
#define PNMOL_SQRT_H

#include "pnmol_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* R (cols x cols, upper triangular, diag >= 0) of A (rows x cols): `jnp.linalg.qr(A, mode="r")` as used at
 * base/sqrt.py:21, :66, :88.  rows < cols is allowed (A is zero-padded: R's trailing rows are then zero). */
int pnmol_qr_r(pnmol_ctx* ctx, const double* A, int rows, int cols, double* R);
/* device time (HIP events) of the factorisation inside the calling thread's last pnmol_qr_r / pnmol_sqrt_* call */
int pnmol_qr_last_ms(float* ms);

/* `propagate_cholesky_factor(S1, S2)` (base/sqrt.py:8-12) = `sqrtm_to_cholesky(vstack(S1^T, S2^T))` (:15-23): the lower
 * triangular factor of S1 S1^T + S2 S2^T.  S1 (n,k1), S2 (n,k2) or NULL; chol_nn (n,n). */
int pnmol_sqrt_propagate_cholesky_factor(pnmol_ctx* ctx, const double* S1, int n, int k1, const double* S2, int k2,
                                         double* chol_nn);

/* `update_sqrt(transition_matrix, cov_cholesky, meascov_sqrtm)` (base/sqrt.py:33-73): QR of
 * [[C^T H^T, C^T], [E^T, 0]].  H (m,D), C (D,D), meascov_sqrtm E (m,m), m <= D.  Outputs (each may be NULL):
 * C_new (D,D) = R3^T lower, gain (D,m) = (R1^-1 R2)^T, Sl (m,m) = R1^T lower.  A singular innovation matrix gives a
 * non-finite gain, as `solve_triangular` does in the reference. */
int pnmol_sqrt_update(pnmol_ctx* ctx, const double* H, int m, int D, const double* C, const double* meascov_sqrtm,
                      double* C_new, double* gain, double* Sl);
/* `update_sqrt_no_meascov(transition_matrix, cov_cholesky)` (base/sqrt.py:76-95) */
int pnmol_sqrt_update_no_meascov(pnmol_ctx* ctx, const double* H, int m, int D, const double* C, double* C_new,
                                 double* gain, double* Sl);

/* ---- the filter step in square-root form ----------------------------------------------------------------------------
 * `_WhiteNoiseEK1Base.attempt_step` as the reference writes it (white.py:96-146): the state is (mean, cov_sqrtm); the
 * predict is `propagate_cholesky_factor(A @ Cl, Ql)` (:114), the update `update_sqrt(H, Clp, E)` (:120), each one QR on
 * the device -- and, from the second consecutive step with the same (dt, operator) on, ONE QR: the rows of the stacked
 * pre-array that do not depend on the state are factored once (same R; PNMOL_SQRT_ONE_QR=0 keeps the
 * two).  About 14x the flops of the covariance form behind pnmol_hip.h (which is the fast path); this is the
 * form to use when the factor itself is wanted, or when the covariance form's resolution (eps |P-|) is not enough.
 * Dense H.  The filter owns ONE state, advanced in place.  `pnmol_filter_desc` as in pnmol_hip.h: d_state 0 or d = white-noise model; d_state = 2d = latent-force model
 * (latent.py:155-233: L = [L, I], B = [B, 0], Gamma = blockdiag(chol K, E_sqrtm), zero noise factors -> the update is
 * `update_sqrt_no_meascov`); mean buffers are then (n, 2d) glued, the factor (2D, 2D) in the stacked order.
 * `desc->dtype`: 0 = fp64 (the reference's arithmetic, src/pnmol/__init__.py:9-11); 1 = the QRs in fp32 (pre-arrays rounded
 * to fp32, Householder QR and compact-WY updates in fp32 on v_mfma_f32_16x16x4_f32; state, mean path, sigma^2 and all
 * buffers of this interface stay double) -- the fp32 mode for num_derivatives >= 2, where the fp32 covariance form of
 * pnmol_hip.h diverges (mean 1e-5 / std 1e-4 up to 768 mesh points at nu = 2); other values are
 * refused with -1. */
typedef struct pnmol_sqrt_filter pnmol_sqrt_filter;
int pnmol_sqrt_filter_create(pnmol_ctx* ctx, const pnmol_filter_desc* desc, pnmol_sqrt_filter** out);
int pnmol_sqrt_filter_destroy(pnmol_sqrt_filter* f);
/* `PDEFilterState(t, y=MultivariateNormal(mean (n,d), cov_sqrtm (D,D)))`, cov_sqrtm in the F-flattened order; any
 * matrix C with C C^T = cov is accepted on input, the factor read back is lower triangular with diag >= 0 */
int pnmol_sqrt_filter_set_state(pnmol_sqrt_filter* f, double t, const double* mean_nd, const double* cov_sqrtm_DD);
int pnmol_sqrt_filter_get_state(pnmol_sqrt_filter* f, double* t, double* mean_nd, double* cov_sqrtm_DD);
/* semilinear EK1 (white.py:189-208), as pnmol_filter_predict_mean / pnmol_filter_set_operator: M = J_x + L (d,d_state) */
int pnmol_sqrt_filter_predict_mean(pnmol_sqrt_filter* f, double dt, double* m_at_d);
int pnmol_sqrt_filter_set_operator(pnmol_sqrt_filter* f, const double* M_dd, const double* shift_d);
/* `estimate_error` (white.py:153-162) in square-root form: Sq = H Ql Ql^T H^T + E E^T = Rq^T Rq, Rq the R factor of
 * [(H Ql)^T; E^T], for the operator currently set and step size dt; kept until dt or the operator changes.  Without it
 * a step of this dt reports error_sigma2 = NaN and a NaN error estimate (fine under the Constant rule, which discards it). */
int pnmol_sqrt_filter_prepare_error_model(pnmol_sqrt_filter* f, double dt);
/* one step / k steps of size dt.  info: diffusion_squared_local = |R1c^-1 z|^2 / m (white.py:125-128 with the
 * positive-diagonal factor), sigma2_whitened = z^T S^-1 z / m, error_sigma2 = z^T Sq^-1 z / m, info = -1 ok / 0
 * non-finite.  error_estimate_d (d), optional: dt * sqrt(diag Sq) * sigma (white.py:117-129). */
int pnmol_sqrt_filter_step(pnmol_sqrt_filter* f, double dt, pnmol_step_out* info, double* error_estimate_d);
/* means_kd / stds_kd: (k, d_state) */
int pnmol_sqrt_filter_steps(pnmol_sqrt_filter* f, int k, double dt, double* means_kd, double* stds_kd,
                            pnmol_step_out* info_k);
int pnmol_sqrt_filter_last_steps_ms(pnmol_sqrt_filter* f, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* PNMOL_SQRT_H */
