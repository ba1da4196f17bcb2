// pnmol_sqrt.hip -- square-root (QR) primitives of the PNMOL filter on MI355X (gfx950), fp64.
//
// What the reference does with `jnp.linalg.qr(mode="r")` (base/sqrt.py:8-95): the R factor of a tall stacked matrix.
// Here: a communication-avoiding Householder QR (TSQR panels + compact-WY trailing updates on the f64 MFMA).
//
//   work matrix W   row-major, Mp x ld doubles, Mp and ld multiples of 32 (zero padded), Mp >= ld
//   panel p         columns [32p, 32p+32); active row blocks p .. nrb-1 (32 rows each)
//   tree level s    members = row blocks p + s*k; a chunk = FAN consecutive members (FAN*32 stacked rows).  k_qr_factor
//                   does a Householder QR of the chunk's 32 panel columns in LDS (LAPACK dlarfg/dlarft conventions), leaves
//                   R in the chunk's first member and V, T in a workspace; k_qr_apply applies (I - V T V^T)^T to the
//                   chunk's rows of every trailing column.  Survivors (first members) form level s*FAN, until one is left.
//
// Only R is kept (the reference never forms Q either).  Rows of R are sign-normalised to a non-negative diagonal at the
// end: the canonical representative of the reference's factor (its signs are LAPACK's, arbitrary).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "pnmol_internal.hpp"
#include "pnmol_sqrt.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int QB = 32;            // panel width = row-block height
constexpr int FAN = 8;            // members per chunk
constexpr int CR = FAN * QB;      // stacked rows of a chunk (256)
constexpr int FT = FAN * 64;      // threads of k_qr_factor: thread (k, g) = (t & 31, t >> 5), rows g, g + NG, ...
constexpr int NG = FT / 32;       // row groups (16)
constexpr int RPT = CR / NG;      // rows per thread (16)
constexpr int VLD = 33;           // LDS row stride of V in k_qr_apply (bank-conflict-free column walks)

#define QCHECK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -2;                                                                     \
        }                                                                                  \
    } while (0)

struct FactorLds {
    double A[CR * QB];      // the chunk's panel columns, stacked; becomes V (unit lower trapezoidal)
    double R[QB * QB];      // rows of R as they are finished
    double red[FAN * QB];   // per-wave partial inner products
    double G[QB * QB];      // V^T V
    double tau[QB], scale[QB];
};

// Member q of chunk c at tree level s of panel p.
__device__ __forceinline__ int member_rb(int p, int s, int c, int q) { return p + s * (FAN * c + q); }

// Householder QR of one chunk's panel (CR x 32, in LDS).  Column step J, with the inner products of column J against all
// columns k >= J taken in one pass (s_k = sum_{i>J} A[i][J] A[i][k]; k = J gives the dlarfg sigma), so a step costs two
// block barriers.  Nothing is updated in place that another thread still reads in the same phase: row J of R goes to
// L.R, the reflector stays unscaled in A's column J (scale[J] applied at the end).
__global__ __launch_bounds__(FT) void k_qr_factor(double* __restrict__ W, long ld, int nrb, int p, int s,
                                                  double* __restrict__ Vws, double* __restrict__ Tws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    FactorLds& L = *reinterpret_cast<FactorLds*>(qr_lds_raw);
    const int c = blockIdx.x, t = threadIdx.x, k = t & 31, g = t >> 5, w = t >> 6, lane = t & 63;
    int nm = 0;
    for (int q = 0; q < FAN; ++q) nm += member_rb(p, s, c, q) < nrb;
    if (nm == 0 || (nm == 1 && s > 1)) return;  // a lone survivor is already triangular (k_qr_apply skips it too)
    const int rows = nm * QB;

    for (int r = 0; r < RPT; ++r) {
        const int i = g + NG * r;
        const int q = i >> 5;
        double v = 0.0;
        if (q < nm) v = W[((long)member_rb(p, s, c, q) * QB + (i & 31)) * ld + (long)p * QB + k];
        L.A[i * QB + k] = v;
    }
    __syncthreads();

    for (int J = 0; J < QB; ++J) {
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = g + NG * r;
            const double a = (i > J && i < rows) ? L.A[i * QB + J] : 0.0;
            part += a * L.A[i * QB + k];
        }
        part += __shfl_xor(part, 32);  // the wave's two row groups
        if (lane < 32) L.red[w * QB + k] = part;
        __syncthreads();
        double sk = 0.0, sJ = 0.0;
#pragma unroll
        for (int ww = 0; ww < FAN; ++ww) {
            sk += L.red[ww * QB + k];
            sJ += L.red[ww * QB + J];
        }
        // dlarfg: H = I - tau v v^T, v = [1; x / (alpha - beta)], beta = -sign(alpha) |(alpha, x)|
        const double alpha = L.A[J * QB + J];
        double beta = alpha, tau = 0.0, scale = 0.0;
        if (sJ != 0.0) {
            beta = -copysign(sqrt(alpha * alpha + sJ), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        const double wk = L.A[J * QB + k] + sk * scale;  // v^T A[:, k]
        const double f = tau * wk;
        if (g == 0) {
            L.R[J * QB + k] = (k > J) ? L.A[J * QB + k] - f : (k == J ? beta : 0.0);
            if (k == J) {
                L.tau[J] = tau;
                L.scale[J] = scale;
            }
        }
        if (k > J) {
            const double fs = f * scale;
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int i = g + NG * r;
                if (i > J && i < rows) L.A[i * QB + k] -= fs * L.A[i * QB + J];
            }
        }
        __syncthreads();
    }

    // V: unit lower trapezoidal
    {
        const double sc = L.scale[k];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = g + NG * r;
            const double a = L.A[i * QB + k];
            L.A[i * QB + k] = (i > k && i < rows) ? a * sc : (i == k ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    // G = V^T V on the MFMA: waves 0..3 take one 16x16 tile each
    if (w < 4) {
        const int fr = lane & 15, fk = lane >> 4, jt = w >> 1, ct = w & 1;
        d4 acc = {0, 0, 0, 0};
        for (int st = 0; st < CR / 4; ++st) {
            const double a = L.A[(4 * st + fk) * QB + jt * 16 + fr];
            const double b = L.A[(4 * st + fk) * QB + ct * 16 + fr];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) L.G[(jt * 16 + fk + 4 * r) * QB + ct * 16 + fr] = acc[r];
    }
    __syncthreads();
    // dlarft (forward, columnwise): T[0:J, J] = -tau_J T[0:J, 0:J] (V^T V)[0:J, J]; lane j keeps row j of T
    double* Tg = Tws + (long)c * QB * QB;
    if (t < QB) {
        double trow[QB];
#pragma unroll
        for (int J = 0; J < QB; ++J) {
            double acc = 0.0;
#pragma unroll
            for (int l = 0; l < J; ++l) acc += trow[l] * L.G[l * QB + J];   // trow[l] = 0 for l < t
            const double tj = L.tau[J];
            trow[J] = (t < J) ? -tj * acc : (t == J ? tj : 0.0);
        }
#pragma unroll
        for (int J = 0; J < QB; ++J) Tg[t * QB + J] = trow[J];
    }
    // V to the workspace, R (upper, zero below) to the first member
    double* Vg = Vws + (long)c * CR * QB;
    for (int r = 0; r < RPT; ++r) {
        const int i = g + NG * r;
        Vg[i * QB + k] = L.A[i * QB + k];
    }
    if (g < 2) {
        for (int r = g; r < QB; r += 2)
            W[((long)member_rb(p, s, c, 0) * QB + r) * ld + (long)p * QB + k] = L.R[r * QB + k];
    }
}

// C <- (I - V T V^T)^T C = C - V (T^T (V^T C)) on the chunk's rows of 64 trailing columns per block, 16 per wave.  The
// wave keeps its 256 x 16 slab of C in registers in the MFMA's B-operand layout (lane (fr, fk) holds C[4s + fk][fr]),
// which is also the accumulator layout of the 16-row tile s / 4 (row fk + 4 (s % 4)): the last product accumulates
// straight into the slab.
__global__ __launch_bounds__(256) void k_qr_apply(double* __restrict__ W, long ld, int nrb, int p, int s,
                                                  const double* __restrict__ Vws, const double* __restrict__ Tws,
                                                  int col0, int ncols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    double* sV = reinterpret_cast<double*>(qr_lds_raw);  // CR x VLD
    double* sT = sV + CR * VLD;                           // 32 x VLD
    double* sWall = sT + QB * VLD;                        // per wave: 32 x 17
    const int c = blockIdx.x, t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4;
    int nm = 0;
    for (int q = 0; q < FAN; ++q) nm += member_rb(p, s, c, q) < nrb;
    if (nm == 0 || (nm == 1 && s > 1)) return;
    const double* Vg = Vws + (long)c * CR * QB;
    const double* Tg = Tws + (long)c * QB * QB;
    for (int e = t; e < CR * QB; e += 256) sV[(e >> 5) * VLD + (e & 31)] = Vg[e];
    for (int e = t; e < QB * QB; e += 256) sT[(e >> 5) * VLD + (e & 31)] = Tg[e];

    const int cw = blockIdx.y * 64 + w * 16;       // this wave's first column, relative to col0
    const bool active = cw < ncols;                // ncols is a multiple of 16
    const long col = col0 + (active ? cw : 0) + fr;
    double cs[CR / 4];
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        const int i = 4 * st + fk, q = i >> 5;
        cs[st] = (q < nm) ? W[((long)member_rb(p, s, c, q) * QB + (i & 31)) * ld + col] : 0.0;
    }
    __syncthreads();

    // W1 = V^T C  (32 x 16)
    d4 w1[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        w1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(4 * st + fk) * VLD + fr], cs[st], w1[0], 0, 0, 0);
        w1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(4 * st + fk) * VLD + 16 + fr], cs[st], w1[1], 0, 0, 0);
    }
    double* sW = sWall + w * (QB * 17);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[(jt * 16 + fk + 4 * r) * 17 + fr] = w1[jt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // W2 = T^T W1: D[j][n] = sum_l T[l][j] W1[l][n]
    d4 w2[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int st = 0; st < QB / 4; ++st) {
        const double b = sW[(4 * st + fk) * 17 + fr];
        w2[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(4 * st + fk) * VLD + fr], b, w2[0], 0, 0, 0);
        w2[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(4 * st + fk) * VLD + 16 + fr], b, w2[1], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[(jt * 16 + fk + 4 * r) * 17 + fr] = w2[jt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double b2[QB / 4];
#pragma unroll
    for (int st = 0; st < QB / 4; ++st) b2[st] = sW[(4 * st + fk) * 17 + fr];
    // C -= V W2, 16 rows at a time
#pragma unroll
    for (int it = 0; it < CR / 16; ++it) {
        d4 acc = {cs[4 * it], cs[4 * it + 1], cs[4 * it + 2], cs[4 * it + 3]};
#pragma unroll
        for (int st = 0; st < QB / 4; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sV[(it * 16 + fr) * VLD + 4 * st + fk], b2[st], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[4 * it + r] = acc[r];
    }
    if (!active) return;
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        const int i = 4 * st + fk, q = i >> 5;
        if (q < nm) W[((long)member_rb(p, s, c, q) * QB + (i & 31)) * ld + col] = cs[st];
    }
}

// dst[j][i] = src[i][j], i < nr, j < nc (tile transpose through LDS)
__global__ __launch_bounds__(256) void k_copy_t(double* __restrict__ dst, long ldd, const double* __restrict__ src,
                                                long lds, int nr, int nc) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < nr && j < nc) ? src[(long)i * lds + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + tx;
        if (i < nr && j < nc) dst[(long)j * ldd + i] = tile[tx][r];
    }
}

// out[i][j] = sum_k C[k][i] H[j][k]  (= (H C)^T), i < nc_c, j < nr_h, k < kk: the top-left block of update_sqrt's
// stacked matrix (base/sqrt.py:59-64).  One 32x32 tile per block on the MFMA.
__global__ __launch_bounds__(256) void k_atbt(double* __restrict__ out, long ldo, const double* __restrict__ C, long ldc,
                                              const double* __restrict__ H, long ldh, int nc_c, int nr_h, int kk) {
    __shared__ double sC[32][33], sH[32][33];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4, wr = w >> 1, wc = w & 1;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = t & 31, ty = t >> 5;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < kk; k0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r;
            sC[r][tx] = (k < kk && i0 + tx < nc_c) ? C[(long)k * ldc + i0 + tx] : 0.0;   // sC[k][i]
            const int j = j0 + r;
            sH[r][tx] = (j < nr_h && k0 + tx < kk) ? H[(long)j * ldh + k0 + tx] : 0.0;   // sH[j][k]
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 8; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sC[4 * st + fk][wr * 16 + fr], sH[wc * 16 + fr][4 * st + fk], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 16 + fk + 4 * r, j = j0 + wc * 16 + fr;
        if (i < nc_c && j < nr_h) out[(long)i * ldo + j] = acc[r];
    }
}

// Block [r0, r0+nr) x [c0, c0+nc) of the factorised work matrix -> out.  tri: entries below W's diagonal read as 0 (they
// hold nothing of R); flip: rows whose diagonal entry is negative change sign (canonical factor); transpose: out is
// (nc x nr) = block^T, else (nr x nc).
__global__ void k_qr_block(const double* __restrict__ W, long ld, int r0, int c0, int nr, int nc,
                           double* __restrict__ out, int transpose, int tri, int flip) {
    const int i = blockIdx.y * blockDim.y + threadIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nr || j >= nc) return;
    const long gi = r0 + i, gj = c0 + j;
    double v = (tri && gj < gi) ? 0.0 : W[gi * ld + gj];
    if (flip && W[gi * ld + gi] < 0.0) v = -v;
    if (transpose) out[(long)j * nr + i] = v; else out[(long)i * nc + j] = v;
}

// X = R1^-1 R2 in place of R2: R1 = W[0:m, 0:m] upper triangular, R2 = W[0:m, c0:c0+nrhs] (base/sqrt.py:72 before the
// transpose).  One block per 32 right-hand sides; block rows bottom-up: X_b = R_bb^-1 (R2_b - sum_{c>b} R_bc X_c), the
// products on the MFMA, the 32x32 back substitution by one thread per right-hand side.  A zero pivot gives inf/nan like
// `solve_triangular` does.
__global__ __launch_bounds__(256) void k_trsm_upper(double* __restrict__ W, long ld, int m, int c0, int nrhs) {
    __shared__ double sR[32][33], sX[32][33], sB[32][33];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4, wr = w >> 1, wc = w & 1;
    const int tx = t & 31, ty = t >> 5;
    const int j0 = blockIdx.x * 32;                 // first right-hand side of this block
    const int mb = (m + 31) / 32;
    for (int b = mb - 1; b >= 0; --b) {
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = b * 32 + wr * 16 + fk + 4 * r, j = j0 + wc * 16 + fr;
            acc[r] = (i < m && j < nrhs) ? W[(long)i * ld + c0 + j] : 0.0;
        }
        for (int c = b + 1; c < mb; ++c) {
            for (int r = ty; r < 32; r += 8) {
                const int i = b * 32 + r, k = c * 32 + tx;
                sR[r][tx] = (i < m && k < m) ? W[(long)i * ld + k] : 0.0;                       // R_bc[i][k]
                const int kx = c * 32 + r, j = j0 + tx;
                sX[r][tx] = (kx < m && j < nrhs) ? W[(long)kx * ld + c0 + j] : 0.0;             // X_c[k][j]
            }
            __syncthreads();
#pragma unroll
            for (int st = 0; st < 8; ++st)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sR[wr * 16 + fr][4 * st + fk], sX[4 * st + fk][wc * 16 + fr], acc, 0, 0, 0);
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sB[wr * 16 + fk + 4 * r][wc * 16 + fr] = acc[r];
        for (int r = ty; r < 32; r += 8) {
            const int i = b * 32 + r, k = b * 32 + tx;
            sR[r][tx] = (i < m && k < m) ? W[(long)i * ld + k] : (i == k ? 1.0 : 0.0);          // R_bb, identity beyond m
        }
        __syncthreads();
        if (t < 32) {                                // right-hand side j0 + t: back substitution in registers
            double x[32];
#pragma unroll
            for (int i = 31; i >= 0; --i) {
                double v = sB[i][t];
#pragma unroll
                for (int k = i + 1; k < 32; ++k) v -= sR[i][k] * x[k];
                x[i] = v / sR[i][i];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (b * 32 + i < m && j0 + t < nrhs) W[(long)(b * 32 + i) * ld + c0 + j0 + t] = x[i];
        }
        __syncthreads();
    }
}

struct QrPlan {
    int rows = 0, cols = 0, Mp = 0, ld = 0, nrb = 0, ncb = 0;
    double *W = nullptr, *Vws = nullptr, *Tws = nullptr;
};

int qr_plan_alloc(pnmol_ctx* ctx, int rows, int cols, QrPlan* pl) {
    pl->rows = rows;
    pl->cols = cols;
    pl->ld = (cols + QB - 1) / QB * QB;
    pl->Mp = (std::max(rows, pl->ld) + QB - 1) / QB * QB;
    pl->nrb = pl->Mp / QB;
    pl->ncb = pl->ld / QB;
    const int maxchunks = (pl->nrb + FAN - 1) / FAN;
    QCHECK(ctx, hipMalloc(&pl->W, sizeof(double) * (size_t)pl->Mp * pl->ld));
    QCHECK(ctx, hipMalloc(&pl->Vws, sizeof(double) * (size_t)maxchunks * CR * QB));
    QCHECK(ctx, hipMalloc(&pl->Tws, sizeof(double) * (size_t)maxchunks * QB * QB));
    return 0;
}

void qr_plan_free(QrPlan* pl) {
    if (pl->W) hipFree(pl->W);
    if (pl->Vws) hipFree(pl->Vws);
    if (pl->Tws) hipFree(pl->Tws);
    *pl = QrPlan();
}

constexpr size_t kFactorLds = sizeof(FactorLds);
constexpr size_t kApplyLds = sizeof(double) * (CR * VLD + QB * VLD + 4 * QB * 17);

int qr_configure(pnmol_ctx* ctx) {
    static bool done = false;   // attributes are per function, per device context; cheap to repeat
    (void)done;
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFactorLds));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kApplyLds));
    return 0;
}

// R factor of the padded work matrix, in place (upper triangle of W's leading ld x ld block).
int qr_inplace(pnmol_ctx* ctx, const QrPlan& pl) {
    for (int p = 0; p < pl.ncb; ++p) {
        const int a = pl.nrb - p;
        const int ntrail = pl.ld - (p + 1) * QB;
        for (int s = 1;; s *= FAN) {
            const int nmem = (a + s - 1) / s, nch = (nmem + FAN - 1) / FAN;
            hipLaunchKernelGGL(k_qr_factor, dim3(nch), dim3(FT), kFactorLds, ctx->stream, pl.W, (long)pl.ld, pl.nrb, p, s,
                               pl.Vws, pl.Tws);
            if (ntrail > 0)
                hipLaunchKernelGGL(k_qr_apply, dim3(nch, (ntrail + 63) / 64), dim3(256), kApplyLds, ctx->stream, pl.W,
                                   (long)pl.ld, pl.nrb, p, s, pl.Vws, pl.Tws, (p + 1) * QB, ntrail);
            if (nch == 1) break;
        }
    }
    QCHECK(ctx, hipGetLastError());
    return 0;
}

thread_local float g_last_qr_ms = -1.f;

// qr_inplace bracketed by HIP events on the ctx stream (read back by pnmol_qr_last_ms after the caller's synchronise)
struct QrTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st;
    explicit QrTimer(hipStream_t s) : st(s) {
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, st);
    }
    void stop() { hipEventRecord(e1, st); }
    ~QrTimer() {
        if (hipEventSynchronize(e1) == hipSuccess) hipEventElapsedTime(&g_last_qr_ms, e0, e1);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
};

}  // namespace

extern "C" {

int pnmol_qr_last_ms(float* ms) {
    if (!ms) return -1;
    *ms = g_last_qr_ms;
    return 0;
}

int pnmol_qr_r(pnmol_ctx* ctx, const double* A, int rows, int cols, double* R) {
    if (!ctx || !A || !R || rows <= 0 || cols <= 0) return -1;
    QCHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = qr_configure(ctx)) return rc;
    QrPlan pl;
    if (int rc = qr_plan_alloc(ctx, rows, cols, &pl)) {
        qr_plan_free(&pl);
        return rc == -2 ? -4 : rc;
    }
    double* dR = nullptr;
    int rc = 0;
    do {
        if (hipMalloc(&dR, sizeof(double) * (size_t)cols * cols) != hipSuccess) { rc = -4; break; }
        if (hipMemsetAsync(pl.W, 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream) != hipSuccess) { rc = -2; break; }
        if (hipMemcpy2DAsync(pl.W, sizeof(double) * pl.ld, A, sizeof(double) * cols, sizeof(double) * cols, rows,
                             hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = -2; break; }
        {
            QrTimer tm(ctx->stream);
            rc = qr_inplace(ctx, pl);
            tm.stop();
        }
        if (rc) break;
        hipLaunchKernelGGL(k_qr_block, dim3((cols + 31) / 32, (cols + 7) / 8), dim3(32, 8), 0, ctx->stream, pl.W,
                           (long)pl.ld, 0, 0, cols, cols, dR, 0, 1, 1);
        if (hipMemcpyAsync(R, dR, sizeof(double) * (size_t)cols * cols, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = -2; break; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = -2; break; }
    } while (0);
    if (rc == -2 && ctx->err.empty()) ctx->err = hipGetErrorString(hipGetLastError());
    if (dR) hipFree(dR);
    qr_plan_free(&pl);
    return rc;
}

}  // extern "C"

namespace {

struct DevBuf {   // host matrix uploaded to the device for the duration of one call
    double* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int upload(pnmol_ctx* ctx, const double* h, size_t count) {
        if (hipMalloc(&p, sizeof(double) * count) != hipSuccess) return -4;
        QCHECK(ctx, hipMemcpyAsync(p, h, sizeof(double) * count, hipMemcpyHostToDevice, ctx->stream));
        return 0;
    }
    int alloc(size_t count) { return hipMalloc(&p, sizeof(double) * count) == hipSuccess ? 0 : -4; }
};

struct PlanGuard {
    QrPlan pl;
    ~PlanGuard() { qr_plan_free(&pl); }
};

inline dim3 tiles(int nc, int nr) { return dim3((nc + 31) / 32, (nr + 31) / 32); }

int run_qr_timed(pnmol_ctx* ctx, const QrPlan& pl) {
    QrTimer tm(ctx->stream);
    const int rc = qr_inplace(ctx, pl);
    tm.stop();
    return rc;
}

// update_sqrt / update_sqrt_no_meascov (base/sqrt.py:33-95): E == nullptr is the noise-free variant
int sqrt_update(pnmol_ctx* ctx, const double* H, int m, int D, const double* C, const double* E, double* C_new,
                double* gain, double* Sl) {
    if (!ctx || !H || !C || m <= 0 || D <= 0 || m > D) return -1;   // the reference pads E to (m, D): needs m <= D
    QCHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = qr_configure(ctx)) return rc;
    PlanGuard g;
    if (int rc = qr_plan_alloc(ctx, D + m, m + D, &g.pl)) return rc == -2 ? -4 : rc;
    const QrPlan& pl = g.pl;
    DevBuf dH, dC, dE, dOut;
    if (int rc = dH.upload(ctx, H, (size_t)m * D)) return rc;
    if (int rc = dC.upload(ctx, C, (size_t)D * D)) return rc;
    if (E) if (int rc = dE.upload(ctx, E, (size_t)m * m)) return rc;
    if (int rc = dOut.alloc((size_t)D * D)) return rc;
    QCHECK(ctx, hipMemsetAsync(pl.W, 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream));
    // [[C^T H^T, C^T], [E^T, 0]]
    hipLaunchKernelGGL(k_atbt, tiles(m, D), dim3(256), 0, ctx->stream, pl.W, (long)pl.ld, dC.p, (long)D, dH.p, (long)D, D, m, D);
    hipLaunchKernelGGL(k_copy_t, tiles(D, D), dim3(256), 0, ctx->stream, pl.W + m, (long)pl.ld, dC.p, (long)D, D, D);
    if (E)
        hipLaunchKernelGGL(k_copy_t, tiles(m, m), dim3(256), 0, ctx->stream, pl.W + (long)D * pl.ld, (long)pl.ld, dE.p, (long)m, m, m);
    if (int rc = run_qr_timed(ctx, pl)) return rc;
    const dim3 tb(32, 8);
    if (Sl) {   // R1^T
        hipLaunchKernelGGL(k_qr_block, dim3((m + 31) / 32, (m + 7) / 8), tb, 0, ctx->stream, pl.W, (long)pl.ld, 0, 0, m, m, dOut.p, 1, 1, 1);
        QCHECK(ctx, hipMemcpyAsync(Sl, dOut.p, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (C_new) {   // R3^T
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_qr_block, dim3((D + 31) / 32, (D + 7) / 8), tb, 0, ctx->stream, pl.W, (long)pl.ld, m, m, D, D, dOut.p, 1, 1, 1);
        QCHECK(ctx, hipMemcpyAsync(C_new, dOut.p, sizeof(double) * (size_t)D * D, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (gain) {   // (R1^-1 R2)^T
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_trsm_upper, dim3((D + 31) / 32), dim3(256), 0, ctx->stream, pl.W, (long)pl.ld, m, m, D);
        hipLaunchKernelGGL(k_qr_block, dim3((D + 31) / 32, (m + 7) / 8), tb, 0, ctx->stream, pl.W, (long)pl.ld, 0, m, m, D, dOut.p, 1, 0, 0);
        QCHECK(ctx, hipMemcpyAsync(gain, dOut.p, sizeof(double) * (size_t)D * m, hipMemcpyDeviceToHost, ctx->stream));
    }
    QCHECK(ctx, hipGetLastError());
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

}  // namespace

extern "C" {

int pnmol_sqrt_propagate_cholesky_factor(pnmol_ctx* ctx, const double* S1, int n, int k1, const double* S2, int k2,
                                         double* chol_nn) {
    if (!ctx || !S1 || !chol_nn || n <= 0 || k1 <= 0 || (S2 && k2 <= 0)) return -1;
    if (!S2) k2 = 0;
    QCHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = qr_configure(ctx)) return rc;
    PlanGuard g;
    if (int rc = qr_plan_alloc(ctx, k1 + k2, n, &g.pl)) return rc == -2 ? -4 : rc;
    const QrPlan& pl = g.pl;
    DevBuf d1, d2, dOut;
    if (int rc = d1.upload(ctx, S1, (size_t)n * k1)) return rc;
    if (S2) if (int rc = d2.upload(ctx, S2, (size_t)n * k2)) return rc;
    if (int rc = dOut.alloc((size_t)n * n)) return rc;
    QCHECK(ctx, hipMemsetAsync(pl.W, 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream));
    // vstack(S1^T, S2^T), base/sqrt.py:11
    hipLaunchKernelGGL(k_copy_t, tiles(k1, n), dim3(256), 0, ctx->stream, pl.W, (long)pl.ld, d1.p, (long)k1, n, k1);
    if (S2)
        hipLaunchKernelGGL(k_copy_t, tiles(k2, n), dim3(256), 0, ctx->stream, pl.W + (long)k1 * pl.ld, (long)pl.ld, d2.p, (long)k2, n, k2);
    if (int rc = run_qr_timed(ctx, pl)) return rc;
    hipLaunchKernelGGL(k_qr_block, dim3((n + 31) / 32, (n + 7) / 8), dim3(32, 8), 0, ctx->stream, pl.W, (long)pl.ld, 0, 0, n, n, dOut.p, 1, 1, 1);
    QCHECK(ctx, hipMemcpyAsync(chol_nn, dOut.p, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToHost, ctx->stream));
    QCHECK(ctx, hipGetLastError());
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int pnmol_sqrt_update(pnmol_ctx* ctx, const double* H, int m, int D, const double* C, const double* meascov_sqrtm,
                      double* C_new, double* gain, double* Sl) {
    if (!meascov_sqrtm) return -1;
    return sqrt_update(ctx, H, m, D, C, meascov_sqrtm, C_new, gain, Sl);
}

int pnmol_sqrt_update_no_meascov(pnmol_ctx* ctx, const double* H, int m, int D, const double* C, double* C_new,
                                 double* gain, double* Sl) {
    return sqrt_update(ctx, H, m, D, C, nullptr, C_new, gain, Sl);
}

}  // extern "C"
