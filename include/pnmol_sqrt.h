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

#ifdef __cplusplus
}
#endif
#endif /* PNMOL_SQRT_H */
