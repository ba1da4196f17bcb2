// pnmol_hip.hip -- MI355X (gfx950) implementation of the PNMOL white-noise EK1 step.
//
// Reference path (schmidtjonathan/pnmol-experiments): src/pnmol/white.py:96-146
// (attempt_step), :169-186 (evaluate_ode), base/iwp.py:13-97 (Phi/Q, preconditioner),
// base/sqrt.py:8-95 (QR predict/update).  The reference is a square-root (QR) filter; its own
// tests (tests/test_base/test_sqrt.py:48-78) assert equivalence with the classic form
//     P- = A P A^T + Q,  S = H P- H^T + R,  S = Ls Ls^T,  W = P- H^T Ls^-T,  P = P- - W W^T,
// which is what runs here, in the reference's Nordsieck-preconditioned coordinates
// (white.py:97-104) so that all entries are O(1).
//
// Device layout (all fp64, row-major, zero padded):
//   state index  (a, j) -> a*dp + j      a = derivative 0..n-1, j = mesh point   ("derivative-major";
//                the C ABI converts from/to the reference's point-major j*n + a)
//   P, P-        Dp x Dp, Dp = n*dp, dp = d rounded up to 32
//   G, F         (mp + Dp + 32) x mp "tall" matrices, mp = (d + nB) rounded up to 32:
//                G = [ S ; P- H^T ; z^T ]  is factorised column-block by column-block
//                (right-looking, NB = 32) into  F = [ Ls ; W ; r^T ],  r = Ls^-1 z.
//   H            never materialised: H = c1 [I;0] (x) e1^T + c0 Hv (x) e0^T with Hv = [-L; B] in ELL
//                (stencil rows), c_i = Nordsieck scale of derivative i  (white.py:110-115, :171-186).
//   A, Q         never materialised: A = A1 (x) I_d, Q = Q1 (x) K with K = Gamma Gamma^T
//                (base/iwp.py:32-53) -> n x n block transform per (j,k) pair.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <atomic>
#include <vector>

#include "pnmol_hip.h"
#include "pnmol_internal.hpp"

namespace {

constexpr int NB = 32;        // factorisation block
constexpr int TLD = NB + 1;   // LDS leading dimension of a 32x32 tile (bank-conflict pad)
constexpr int MAXN = 4;       // max derivatives + 1
typedef double d4 __attribute__((ext_vector_type(4)));

struct IwpConsts {
    double A1[MAXN * MAXN];  // flip(pascal_lower)  base/iwp.py:24-27
    double Q1[MAXN * MAXN];  // flip(hilbert)       base/iwp.py:29
    double ts[MAXN];         // frame change  s_old[a] / s_new[a]
};

// ------------------------------------------------------------------------------------------
// H apply (stencil gather).  ELL arrays are [e*mp + i].
// ------------------------------------------------------------------------------------------
struct MeasModel {
    const int* ell_col;
    const double* ell_val;
    int w, d, m, dp, mp;
    double c0, c1;
};

// Row i of H as (index, coefficient) pairs: slot 0 is the derivative-1 entry c1 at (1,i) (PDE rows only), slots
// 1..w the ELL stencil entries c0 * Hv[i,e] at (0, col).  Loaded once per thread so that the dot products below
// issue independent loads (the loops are unrolled in groups of four; absent entries get coefficient 0 / index 0).
constexpr int HW = 4;  // fast-path stencil width; wider rows (dense Jacobians) use the generic loops

struct HRow {
    int idx[HW + 1];
    double cf[HW + 1];
};

__device__ __forceinline__ void h_row_load(const MeasModel& mm, int i, HRow& r) {
    r.idx[0] = (i < mm.d) ? mm.dp + i : 0;
    r.cf[0] = (i < mm.d) ? mm.c1 : 0.0;
#pragma unroll
    for (int e = 0; e < HW; ++e) {
        const int cidx = (e < mm.w) ? mm.ell_col[e * mm.mp + i] : -1;
        const double val = (e < mm.w) ? mm.ell_val[e * mm.mp + i] : 0.0;
        r.idx[e + 1] = cidx >= 0 ? cidx : 0;
        r.cf[e + 1] = cidx >= 0 ? mm.c0 * val : 0.0;
    }
}

// (H x)[i] for a state-indexed vector x (global or LDS).  XT = double, or float for a row of an fp32 covariance
// (pnmol_filter_desc.dtype): the stencil weights and the accumulation stay fp64 ("L assembled in fp64").
template <typename XT>
__device__ __forceinline__ double h_row_dot(const MeasModel& mm, int i, const XT* x) {
    double v = 0.0;
    if (i < mm.d) v = mm.c1 * x[mm.dp + i];
    for (int e = 0; e < mm.w; ++e) {
        const int cidx = mm.ell_col[e * mm.mp + i];
        if (cidx >= 0) v += mm.c0 * mm.ell_val[e * mm.mp + i] * x[cidx];
    }
    return v;
}

// same value (same summation order: derivative-1 term first, then the stencil entries) from a preloaded row
template <int W, typename XT>
__device__ __forceinline__ double h_row_dot_w(const HRow& r, const XT* __restrict__ x) {
    double xv[W + 1];
#pragma unroll
    for (int e = 0; e <= W; ++e) xv[e] = x[r.idx[e]];
    double v = r.cf[0] * xv[0];
#pragma unroll
    for (int e = 1; e <= W; ++e) v += r.cf[e] * xv[e];
    return v;
}

// The vector part of the predict (one workgroup of 256 threads): m- = A m, z = H m- + shift (into the z row of G and
// zbuf), reset of the dependency flags of the coming k_sweep.  `mpl` = Dp doubles of LDS.
struct RoleArgs {
    IwpConsts c;
    int dp;
    const double* min;
    double* mpred;
    const double* shift;
    double* G;
    double* zbuf;
    MeasModel mm;
    int* flags;
    int nflags;
};

template <int N>
__device__ __forceinline__ void predict_vectors(double* mpl, int tid, const IwpConsts& c, int dp, const double* __restrict__ min,
                                                double* __restrict__ mpred, const double* __restrict__ shift,
                                                double* __restrict__ G, double* __restrict__ zbuf, const MeasModel& mm,
                                                int* __restrict__ flags, int nflags) {
    const long Dp = (long)N * dp;
    for (int e = tid; e < nflags; e += 256) flags[e] = 0;  // dependency flags of this step's k_sweep
    for (int j = tid; j < dp; j += 256) {
        double x[N];
#pragma unroll
        for (int a = 0; a < N; ++a) x[a] = c.ts[a] * min[a * dp + j];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < N; ++q) sacc += c.A1[a * MAXN + q] * x[q];
            mpl[a * dp + j] = sacc;
            mpred[a * dp + j] = sacc;
        }
    }
    __syncthreads();
    for (int i = tid; i < mm.mp; i += 256) {
        const double v = (i < mm.m) ? h_row_dot(mm, i, mpl) + shift[i] : 0.0;
        G[((long)mm.mp + Dp) * mm.mp + i] = v;
        zbuf[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// K1  predict:  P-_ab = sum_ce A1[a,c] A1[b,e] ts_c ts_e P_ce + Q1[a,b] K      (HBM-bound pass)
// One extra workgroup (blockIdx.y == 0) does the vector work of the step start: m- = A m, z = H m- + shift
// (into the z row of G and zbuf) and advances the step counter: every later kernel of this step writes its
// per-step outputs to slot *ctr - 1, so all steps launch with identical arguments (hipGraph replay).
// ------------------------------------------------------------------------------------------
template <int N, typename PT>
__device__ __forceinline__ void predict_tile(const PT* __restrict__ Pin, PT* __restrict__ Pout, const double* __restrict__ Kg,
                                             const IwpConsts& c, int dp, int j, int k) {
    const long Dp = (long)N * dp;
    double X[N][N];
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) X[a][b] = c.ts[a] * c.ts[b] * Pin[((long)a * dp + j) * Dp + (long)b * dp + k];
    const double kjk = Kg[(long)j * dp + k];
    double T[N][N];
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int e = 0; e < N; ++e) {
            double sacc = 0.0;
#pragma unroll
            for (int q = 0; q < N; ++q) sacc += c.A1[a * MAXN + q] * X[q][e];
            T[a][e] = sacc;
        }
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) {
            double sacc = c.Q1[a * MAXN + b] * kjk;
#pragma unroll
            for (int e = 0; e < N; ++e) sacc += T[a][e] * c.A1[b * MAXN + e];
            Pout[((long)a * dp + j) * Dp + (long)b * dp + k] = (PT)sacc;
        }
}

// p32: the covariance is stored in fp32 (pnmol_filter_desc.dtype = 1); the n x n block transform itself is done in fp64
template <int N>
__global__ __launch_bounds__(256) void k_predict(const void* __restrict__ Pin, void* __restrict__ Pout,
                                                 const double* __restrict__ Kg, IwpConsts c, int dp,
                                                 const double* __restrict__ min, double* __restrict__ mpred,
                                                 const double* __restrict__ shift, double* __restrict__ G,
                                                 double* __restrict__ zbuf, MeasModel mm, int* __restrict__ ctr,
                                                 int* __restrict__ flags, int nflags, int p32) {
    if (blockIdx.y == 0) {  // dispatched first, so its short dependent-load chain hides behind the tile blocks
        if (blockIdx.x != 0) return;
        extern __shared__ double mpl[];  // predicted mean, Dp doubles
        const int tid = threadIdx.y * 32 + threadIdx.x;
        predict_vectors<N>(mpl, tid, c, dp, min, mpred, shift, G, zbuf, mm, flags, nflags);
        if (tid == 0) *ctr += 1;
        return;
    }
    const int k = blockIdx.x * 32 + threadIdx.x;
    const int j = (blockIdx.y - 1) * 8 + threadIdx.y;
    if (p32) predict_tile<N>(static_cast<const float*>(Pin), static_cast<float*>(Pout), Kg, c, dp, j, k);
    else predict_tile<N>(static_cast<const double*>(Pin), static_cast<double*>(Pout), Kg, c, dp, j, k);
}

// S[ip, i] = (H P- H^T)[ip, i] + R[ip, i] straight from P- (identity on the padding).  Same association order as
// H (P- H^T): inner sum over the stencil of column i, outer sum over the stencil of row ip.
template <typename PT>
__device__ __forceinline__ double s_entry(const PT* __restrict__ Ppred, long Dp, const MeasModel& mm, int ip, int i,
                                          const double* __restrict__ rdiag, const double* __restrict__ Rdense) {
    if (ip >= mm.m || i >= mm.m) return (ip == i) ? 1.0 : 0.0;
    double v = 0.0;
    if (ip < mm.d) v = mm.c1 * h_row_dot(mm, i, Ppred + (long)(mm.dp + ip) * Dp);
    for (int e = 0; e < mm.w; ++e) {
        const int cidx = mm.ell_col[e * mm.mp + ip];
        if (cidx >= 0) v += mm.c0 * mm.ell_val[e * mm.mp + ip] * h_row_dot(mm, i, Ppred + (long)cidx * Dp);
    }
    if (ip == i) v += rdiag[i];
    if (Rdense) v += Rdense[(long)ip * mm.mp + i];
    return v;
}

// stencil width <= W: all index loads, then all P- loads, are independent (pipelined) instead of 16 dependent gathers
template <int W, typename PT>
__device__ __forceinline__ double s_entry_w(const PT* __restrict__ Ppred, long Dp, const MeasModel& mm,
                                            const HRow& rrow /* row ip */, const HRow& rcol /* row i */, int ip, int i,
                                            const double* __restrict__ rdiag, const double* __restrict__ Rdense) {
    if (ip >= mm.m || i >= mm.m) return (ip == i) ? 1.0 : 0.0;
    double t[W + 1];
#pragma unroll
    for (int e = 0; e <= W; ++e) t[e] = h_row_dot_w<W>(rcol, Ppred + (long)rrow.idx[e] * Dp);
    double v = rrow.cf[0] * t[0];
#pragma unroll
    for (int e = 1; e <= W; ++e) v += rrow.cf[e] * t[e];
    if (ip == i) v += rdiag[i];
    if (Rdense) v += Rdense[(long)ip * mm.mp + i];
    return v;
}

// ------------------------------------------------------------------------------------------
// Helpers of the 32x32 diagonal-block factorisation
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double x, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

//
// Semi-definite handling: a pivot that has lost all significance against its original diagonal entry
// S_jj (p <= 1e-13 |S_jj|) is treated as zero -- column j of L and row/column j of L^-1 become 0, i.e. that
// measurement contributes nothing.  This is the noise-free Dirichlet row whose prior variance is already 0
// (exact value 0, rounding noise of either sign in covariance form; the reference's square-root form gets
// a tiny positive number and an equally negligible update).  Only a pivot that is negative at the 1e-3 level
// of a significant row (or NaN) is reported through `info` (checked for all 32 pivots at once after the loop).
//

// ------------------------------------------------------------------------------------------
// Two-wave variant of the 32x32 Cholesky + inverse (the one the step uses).
// Wave 0 factorises: lane (i, h) = (l & 31, l >> 5) holds row i's columns of parity h, u[q] = A[i][2q+h], so one
// FMA instruction updates two columns.  Column step J: the pivot is already known as a wave-uniform scalar (it was
// predicted during step J-1 by one scalar FMA), column J is scaled and published to LDS (col[J], permuted so that
// each half reads its multipliers with ds_read_b128 broadcasts), the next pivot is predicted from two
// v_readlanes, and the general update reads the row multiplier and the column multipliers back from LDS -- that
// LDS round trip overlaps the rsqrt chain of the next column instead of sitting in it.
// Wave 1 builds L^-1 one column behind: it applies the same elementary operations to the identity
// (Y = X^T, same layout), taking 1/sqrt(pivot) and the column multipliers from LDS once flag[J] is set.
// Wave 0 never waits for wave 1, so there is no deadlock; each column has its own LDS buffer (no reuse hazard).
// ------------------------------------------------------------------------------------------
#ifdef PNMOL_STAMP
__device__ unsigned long long pnmol_stamp_out[40];
#define pnmol_stamp pnmol_stamp_out
#endif
struct Diag2wLds {
    double col[NB][NB];   // col[J][pi(k)] = L[k][J]  (pi(k) = 16*(k&1) + (k>>1))
    double dump[NB];      // where the half that does not own column J writes (keeps the stream branch-free)
    double rs[NB];        // 1/sqrt(pivot J) or 0
    double piv[NB];       // pivot J
    int flag[NB];         // column J of `col`/`rs` is published
};

// value of x held by the 32-lane half HJ, delivered to both halves (v_permlane32_swap: VALU latency, no LDS)
template <int HJ>
__device__ __forceinline__ double from_half(double x) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const u2 rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);  // [0]: lanes 0..31 everywhere, [1]: lanes 32..63
    const u2 rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)rh[HJ], (int)rl[HJ]);
}

// u[q] -= m * L[2q+h][J] for the columns 2q+h > J of this lane's parity; the multiplier of column J+1 (`bnext`,
// wave-uniform, only meaningful when HAVE_NEXT) does not go through LDS: it feeds the critical path.
template <int J, bool HAVE_NEXT>
__device__ __forceinline__ void diag2w_update(double (&u)[NB / 2], const double* __restrict__ colJ, double nm, int h,
                                              double bnext) {
#pragma unroll
    for (int q = ((J + 1) >> 1) & ~1; q < NB / 2; q += 2) {
        const double2 b = *reinterpret_cast<const double2*>(colJ + h * 16 + q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int qq = q + t;
            const double bv = t ? b.y : b.x;
            if (2 * qq > J) {  // both parities of this register are right of column J
                if (HAVE_NEXT && 2 * qq == J + 1) u[qq] = fma(nm, h ? bv : bnext, u[qq]);          // (J odd) k = J+1 on h = 0
                else if (HAVE_NEXT && 2 * qq + 1 == J + 1) u[qq] = fma(nm, h ? bnext : bv, u[qq]);  // unreachable (2qq > J)
                else u[qq] = fma(nm, bv, u[qq]);
            } else if (2 * qq + 1 > J) {  // J == 2 qq: only the odd column 2qq+1 = J+1 is updated
                u[qq] = h ? fma(nm, HAVE_NEXT ? bnext : bv, u[qq]) : u[qq];
            }
        }
    }
}

template <int J>
__device__ __forceinline__ void diag2w_factor_col(double (&u)[NB / 2], double& p, Diag2wLds* L, int lane, int pi_i,
                                                  double thrv) {
    constexpr int qJ = J >> 1, hJ = J & 1;
    const int h = lane >> 5;
#ifdef PNMOL_STAMP
    pnmol_stamp[J] = __builtin_amdgcn_s_memtime();
#endif
    // Only  rsq -> refine -> (a[J+1][J] * rs) -> fma  is on the pivot-to-pivot chain; everything that does not
    // need 1/sqrt(p) (pivot test, broadcasts of the still unscaled entries) is issued beside it.
    const double y0 = __builtin_amdgcn_rsq(p);
    const bool ok = p > bcast_lane(thrv, J);
    double an = 0.0, aold = 0.0;
    if constexpr (J + 1 < NB) {
        an = bcast_lane(u[qJ], (J + 1) + 32 * hJ);                               // A[J+1][J], unscaled
        aold = bcast_lane(u[(J + 1) >> 1], (J + 1) + 32 * ((J + 1) & 1));        // A[J+1][J+1]
    }
    const double e = fma(-(p * y0), y0, 1.0);
    const double y = fma(y0 * e, fma(0.375, e, 0.5), y0);
    const double rs = ok ? y : 0.0;
    const double pj = p;
    if constexpr (J + 1 < NB) {
        const double sj = an * rs;  // = L[J+1][J]
        p = fma(-sj, sj, aold);     // next pivot (the same FMA the general update performs on that element)
        const double vj = u[qJ] * rs;
        const bool own = (h == hJ);
        u[qJ] = own ? vj : u[qJ];
        (own ? &L->col[J][0] : &L->dump[0])[pi_i] = vj;
        L->rs[J] = rs;  // wave-uniform values: every lane stores the same word (no divergent branch in the stream)
        L->piv[J] = pj;
        asm volatile("" ::: "memory");
        __hip_atomic_store(&L->flag[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const double m = from_half<hJ>(vj);  // row multiplier L[i][J] for both halves
        diag2w_update<J, true>(u, L->col[J], -m, h, sj);
    } else {
        const double vj = u[qJ] * rs;
        u[qJ] = (h == hJ) ? vj : u[qJ];
        L->rs[J] = rs;
        L->piv[J] = pj;
        asm volatile("" ::: "memory");
        __hip_atomic_store(&L->flag[J], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifndef PNMOL_NO_COLBARRIER
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) asm volatile("" : "+v"(u[q]));
#endif
}

template <int J>
__device__ __forceinline__ void diag2w_inverse_col(double (&u)[NB / 2], Diag2wLds* L, int lane, int pi_c) {
    constexpr int qJ = J >> 1, hJ = J & 1;
    const int h = lane >> 5;
    while (__hip_atomic_load(&L->flag[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    const double rs = L->rs[J];
    const double yj = u[qJ] * rs;
    const bool own = (h == hJ);
    u[qJ] = own ? yj : u[qJ];
    if constexpr (J + 1 < NB) {
        const double m = from_half<hJ>(yj);
        diag2w_update<J, false>(u, L->col[J], -m, h, 0.0);
    }
#ifndef PNMOL_NO_COLBARRIER
#pragma unroll
    for (int q = 0; q < NB / 2; ++q) asm volatile("" : "+v"(u[q]));
#endif
}

template <int... Js>
__device__ __forceinline__ void diag2w_factor_all(double (&u)[NB / 2], double& p, Diag2wLds* L, int lane, int pi_i,
                                                  double thrv, std::integer_sequence<int, Js...>) {
    (diag2w_factor_col<Js>(u, p, L, lane, pi_i, thrv), ...);
}
template <int... Js>
__device__ __forceinline__ void diag2w_inverse_all(double (&u)[NB / 2], Diag2wLds* L, int lane, int pi_c,
                                                   std::integer_sequence<int, Js...>) {
    (diag2w_inverse_col<Js>(u, L, lane, pi_c), ...);
}

// Called by waves 0 and 1 of a workgroup (wave = 0/1) after T holds the symmetric tile and a __syncthreads() that
// also covers the zeroing of L->flag.  Writes L (upper zeroed) to Fd (leading dim ld) and L^-1 to Li (32x32).
// `sdiag_blk` = the 32 original diagonal entries of this block.  WT: L^-1 is read by OTHER workgroups of the same
// launch (k_sweep), so it leaves through agent-scope (write-through) stores.
template <bool WT = false>
__device__ __forceinline__ void diag2w_from_lds(const double* T, double* __restrict__ Fd, long ld,
                                                double* __restrict__ Li, int wave, int lane, int* info, int base,
                                                const double* sdiag_blk, double smax, Diag2wLds* L) {
    const int i = lane & 31, h = lane >> 5;
    const int pi_i = 16 * (i & 1) + (i >> 1);
    double u[NB / 2];
    if (wave == 0) {
        const double sdv = fabs(sdiag_blk[i]);
        const double thrv = 1e-13 * sdv;
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) u[q] = T[i * TLD + 2 * q + h];
        double p = T[0];
        diag2w_factor_all(u, p, L, lane, pi_i, thrv, std::make_integer_sequence<int, NB>{});
#ifdef PNMOL_STAMP
        pnmol_stamp[32] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {
            double2* dst = nullptr;
            (void)dst;
            Fd[(long)i * ld + 2 * q + h] = (2 * q + h <= i) ? u[q] : 0.0;
        }
        if (lane < NB) {
            const double pv = L->piv[lane];
            const bool fatal = !(pv > thrv) && (!(pv == pv) || (pv < -1e-3 * sdv && sdv > 1e-12 * smax));
            const unsigned long long mk = __ballot(fatal);
            if (mk != 0 && lane == 0) atomicMin(info, base + __builtin_ctzll(mk));
        }
    } else {
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) u[q] = (2 * q + h == i) ? 1.0 : 0.0;
        diag2w_inverse_all(u, L, lane, pi_i, std::make_integer_sequence<int, NB>{});
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) {  // u[q] = Y[i][2q+h] = Linv[2q+h][i]
            if constexpr (WT) __hip_atomic_store(&Li[(2 * q + h) * NB + i], u[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else Li[(2 * q + h) * NB + i] = u[q];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Four-wave, 4-column-blocked variant of the 32x32 Cholesky + inverse (the one k_sweep uses).
//
// The column-at-a-time scheme above spends ~60 instructions per pivot, 43 of them fixed overhead (publication,
// broadcasts, pivot chain): 5.3 us per block, and 17 blocks in a row are the critical path of the whole filter step.
// Here a step eliminates FOUR columns:
//   * the tile lives in MFMA accumulators, NEGATED (n = -T) and symmetric, one 16x16 quadrant per wave as it falls
//     out of the sweep's own accumulation (no LDS staging).  Read "transposed", accumulator register c of a
//     quadrant IS the 4-column panel c in MFMA operand layout: lane (row, k) holds T[row][4c + k];
//   * the 4x4 pivot block is pulled into scalar registers (v_readlane) and factorised + inverted in closed form on
//     wave-uniform values: only 4 x (rsq -> refine) + two ops between them are sequential;
//   * -Linv4 is scattered into operand layout by ten FMAs with lane-constant indicators, and ONE MFMA gives the
//     panel solve  X = B Linv4^T  (as X^T = Linv4 B^T: the result lands in operand layout again), ONE more the
//     rank-4 trailing update  n += X X^T  of the quadrant that holds the next pivot block.
// Roles (w = wave): w0 factorises quadrant (0,0) [steps 0..3]; w3 follows it with the panel solve of rows 16..31 and the
// update of quadrants (0,1), (1,1), then factorises (1,1) [steps 4..7] -- the critical path changes waves once, no
// register hand-over; w1 gives its quadrant (0,1) to w3 and then builds L^-1 one step behind (same elementary block
// operations applied to the identity); w2 is free (publishes the row's last tile in k_sweep) and finally writes L.
// Waves talk through LDS buffers that are written once per block (no reuse hazard) + flags; w0 never waits.
// Semi-definite pivots: as above (rs = 0 -> zero column of L, zero row/column of L^-1).
// ------------------------------------------------------------------------------------------
struct Diag4Lds {
    double x0[4][64];   // step c < 4: X panel rows 0..15, masked to row >= column, operand layout (lane = row + 16 k)
    double x1[8][64];   // X panel rows 16..31 (steps 4..7: masked)
    double la[8][64];   // -Linv4 of step c as MFMA A operand: lane i + 16 k (i < 4, k <= i), 0 elsewhere
    double n01[4][64];  // quadrant (0,1) of n, handed from w1 to w3
    double piv[NB];     // the 32 pivots (for the info word)
    int flagA[8];       // la[c] and x0[c] (c < 4) / x1[c] (c >= 4) are published
    int flagB[4];       // x1[c], c < 4, is published
    int flagN;          // n01 is published
    int pad[3];
};

__device__ __forceinline__ double rsq_refined(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);
    return fma(y0 * e, fma(0.375, e, 0.5), y0);
}
__device__ __forceinline__ void lds_flag_set(int* f) {
    // LDS operations of one wave execute in issue order: the flag only has to be EMITTED behind the data stores, and
    // right there -- without the second barrier the compiler sinks the (relaxed) store to the end of the next step
    asm volatile("" ::: "memory");
    __hip_atomic_store(f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // ... and the scheduler would sink the whole publication behind later ALU work
}
__device__ __forceinline__ void lds_flag_wait(const int* f) {
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// lane-constant indicators of the ten entries (i >= k) of the 4x4 operand: ind[e] = (lane == i + 16 k)
__device__ __forceinline__ void diag4_indicators(double (&ind)[10], int lane) {
    int e = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k <= i; ++k) ind[e++] = (lane == i + 16 * k) ? 1.0 : 0.0;
}

// One 4-column step on the 16x16 quadrant accumulator n (negated, symmetric): local columns 4 CP .. 4 CP + 3.
// thrv: lane l holds the zero-pivot threshold of local row l & 15.  Publishes -Linv4 (la_dst), the masked panel
// (x_dst) and the pivots, then sets *flag.
template <int CP>
__device__ __forceinline__ void diag4_step(d4& n, double thrv, const double (&ind)[10], int lane, double* la_dst,
                                           double* x_dst, double* piv_dst, int* flag) {
    const int fr = lane & 15, fk = lane >> 4;
    const double src = n[CP];  // lane (i', k): -T[i'][4 CP + k]  (row view of the symmetric quadrant)
#define D4E(i, k) (-bcast_lane(src, 4 * CP + (i) + 16 * (k)))
    const double p0 = D4E(0, 0);
    const double r0 = p0 > bcast_lane(thrv, 4 * CP + 0) ? rsq_refined(p0) : 0.0;
    const double l10 = D4E(1, 0) * r0, l20 = D4E(2, 0) * r0, l30 = D4E(3, 0) * r0;
    const double p1 = fma(-l10, l10, D4E(1, 1));
    const double r1 = p1 > bcast_lane(thrv, 4 * CP + 1) ? rsq_refined(p1) : 0.0;
    const double l21 = fma(-l20, l10, D4E(2, 1)) * r1, l31 = fma(-l30, l10, D4E(3, 1)) * r1;
    const double p2 = fma(-l21, l21, fma(-l20, l20, D4E(2, 2)));
    const double r2 = p2 > bcast_lane(thrv, 4 * CP + 2) ? rsq_refined(p2) : 0.0;
    const double l32 = fma(-l31, l21, fma(-l30, l20, D4E(3, 2))) * r2;
    const double p3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, D4E(3, 3))));
    const double r3 = p3 > bcast_lane(thrv, 4 * CP + 3) ? rsq_refined(p3) : 0.0;
#undef D4E
    // Linv4 (lower): rows scaled by r_i; everything that does not need r3 is ready before it
    const double i10 = -(l10 * r0) * r1;
    const double i21 = -(l21 * r1) * r2;
    const double i20 = -fma(l21, i10, l20 * r0) * r2;
    const double s32 = l32 * r2, s31 = fma(l32, i21, l31 * r1), s30 = fma(l32, i20, fma(l31, i10, l30 * r0));
    // la = -Linv4 in operand layout (rows i < 4 of the A operand -> the product lands in accumulator register 0)
    double la = ind[0] * -r0;
    la = fma(ind[1], -i10, la);
    la = fma(ind[2], -r1, la);
    la = fma(ind[3], -i20, la);
    la = fma(ind[4], -i21, la);
    la = fma(ind[5], -r2, la);
    double lb = ind[6] * (s30 * r3);
    lb = fma(ind[7], s31 * r3, lb);
    lb = fma(ind[8], s32 * r3, lb);
    lb = fma(ind[9], -r3, lb);
    la += lb;
    const d4 zero = {0, 0, 0, 0};
    const d4 xo = __builtin_amdgcn_mfma_f64_16x16x4f64(la, src, zero, 0, 0, 0);  // [0]: lane (row, i) = X[row][i]
    const double x = (fr >= 4 * CP + fk) ? xo[0] : 0.0;                            // finished rows / upper part of L4: 0
    if constexpr (CP < 3) n = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, n, 0, 0, 0);
    la_dst[lane] = la;
    x_dst[lane] = x;
    piv_dst[0] = p0;  // wave-uniform values: every lane stores the same words (no divergent branch in the stream)
    piv_dst[1] = p1;
    piv_dst[2] = p2;
    piv_dst[3] = p3;
    lds_flag_set(flag);
}

// w3, step C < 4: panel solve of rows 16..31 and the update of quadrants (0,1) and (1,1)
template <int C>
__device__ __forceinline__ void diag4_panel_step(d4& n01, d4& n11, Diag4Lds* L, int lane) {
    lds_flag_wait(&L->flagA[C]);
    const double la = L->la[C][lane], x0 = L->x0[C][lane];
    const d4 zero = {0, 0, 0, 0};
    const d4 xo = __builtin_amdgcn_mfma_f64_16x16x4f64(la, n01[C], zero, 0, 0, 0);
    const double x1 = xo[0];
    n11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, n11, 0, 0, 0);
    if constexpr (C < 3) n01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, n01, 0, 0, 0);
    L->x1[C][lane] = x1;
    lds_flag_set(&L->flagB[C]);
}

// w1, step C: the block operations of step C applied to W (starts as the identity, ends as L^-1; natural layout,
// quadrants (0,0), (1,0), (1,1))
template <int C>
__device__ __forceinline__ void diag4_inverse_step(d4& w00, d4& w10, d4& w11, Diag4Lds* L, int lane) {
    const d4 zero = {0, 0, 0, 0};
    lds_flag_wait(&L->flagA[C]);
    const double la = L->la[C][lane];
    if constexpr (C < 4) {
        const double x0 = L->x0[C][lane];
        const double yn = __builtin_amdgcn_mfma_f64_16x16x4f64(la, w00[C], zero, 0, 0, 0)[0];  // -Linv4 W[block rows]
        w00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, yn, w00, 0, 0, 0);
        lds_flag_wait(&L->flagB[C]);
        const double x1 = L->x1[C][lane];
        w10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yn, w10, 0, 0, 0);
        w00[C] = -yn;
    } else {
        constexpr int CP = C - 4;
        const double yn0 = __builtin_amdgcn_mfma_f64_16x16x4f64(la, w10[CP], zero, 0, 0, 0)[0];
        const double yn1 = __builtin_amdgcn_mfma_f64_16x16x4f64(la, w11[CP], zero, 0, 0, 0)[0];
        if constexpr (C < 7) {
            const double x1 = L->x1[C][lane];
            w10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yn0, w10, 0, 0, 0);
            w11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yn1, w11, 0, 0, 0);
        }
        w10[CP] = -yn0;
        w11[CP] = -yn1;
    }
}

// Called by all four waves of a workgroup.  nq = this wave's quadrant (w >> 1, w & 1) of n = -T in accumulator layout
// (quadrant (1,0) of w2 is not used; quadrant (0,1) holds the transposed LOWER entries).  L->flag*: zero on entry.
// Fd (leading dim ld): L, upper part zeroed.  Li: L^-1 (32x32; its upper-right quadrant is never written: the caller
// keeps it zero).  Returns after this wave's part; w1 returns with its L^-1 stores issued, not drained.
// `lenient` codes of the sweep kernels: 0 = strict (fp64 step), 1 = a non-positive pivot is a dropped direction, never an
// error (Cholesky factor of a covariance that is PSD only up to rounding), 2 = the same for an fp32 covariance
// (pnmol_filter_desc.dtype = 1), whose innovation matrix carries errors of 6e-8 |P-|: a pivot below 3e-6 of its diagonal
// entry has lost all significance there.
__device__ __forceinline__ double pivot_tolerance(int lenient) { return lenient == 2 ? 3e-6 : 1e-13; }

// L itself (nobody inside the sweep reads a diagonal tile) and the info word, from the LDS record of a finished block
__device__ __forceinline__ void diag4_output(Diag4Lds* L, int lane, double* __restrict__ Fd, long ld, int* info, int base,
                                             const double* sd_blk, double smax, double pivtol = 1e-13) {
    const int fr = lane & 15, fk = lane >> 4;
    lds_flag_wait(&L->flagA[7]);
    lds_flag_wait(&L->flagB[3]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Fd[(long)fr * ld + 4 * c + fk] = L->x0[c][lane];
        Fd[(long)(16 + fr) * ld + 4 * c + fk] = L->x1[c][lane];
        Fd[(long)fr * ld + 16 + 4 * c + fk] = 0.0;
        Fd[(long)(16 + fr) * ld + 16 + 4 * c + fk] = L->x1[4 + c][lane];
    }
    if (lane < NB) {
        const double pv = L->piv[lane], sdv = fabs(sd_blk[lane]);
        const bool fatal = !(pv > pivtol * sdv) && (!(pv == pv) || (pv < -1e-3 * sdv && sdv > 1e-12 * smax));
        const unsigned long long mk = __ballot(fatal);
        if (mk != 0 && lane == 0) atomicMin(info, base + __builtin_ctzll(mk));
    }
}

// TO_LDS: L^-1 also goes to sLinv (row-major, leading dimension TLD; upper-right quadrant untouched).  OUT2 = false:
// wave 2 returns at once, the caller runs diag4_output later.
template <bool WT, bool TO_LDS = false, bool OUT2 = true, bool TO_GLOBAL = true>
__device__ __forceinline__ void diag4_factor(const d4& nq, Diag4Lds* L, int w, int lane, double* __restrict__ Fd, long ld,
                                             double* __restrict__ Li, int* info, int base, const double* sd_blk, double smax,
                                             double* sLinv = nullptr, double pivtol = 1e-13) {
    const int fr = lane & 15, fk = lane >> 4;
    if (w == 0) {
        d4 n = nq;
        double ind[10];
        diag4_indicators(ind, lane);
        const double thrv = pivtol * fabs(sd_blk[fr]);
        diag4_step<0>(n, thrv, ind, lane, L->la[0], L->x0[0], L->piv + 0, &L->flagA[0]);
        diag4_step<1>(n, thrv, ind, lane, L->la[1], L->x0[1], L->piv + 4, &L->flagA[1]);
        diag4_step<2>(n, thrv, ind, lane, L->la[2], L->x0[2], L->piv + 8, &L->flagA[2]);
        diag4_step<3>(n, thrv, ind, lane, L->la[3], L->x0[3], L->piv + 12, &L->flagA[3]);
    } else if (w == 3) {
        d4 n11 = nq, n01;
        double ind[10];
        diag4_indicators(ind, lane);
        const double thrv = pivtol * fabs(sd_blk[16 + fr]);
        lds_flag_wait(&L->flagN);
#pragma unroll
        for (int r = 0; r < 4; ++r) n01[r] = L->n01[r][lane];
        diag4_panel_step<0>(n01, n11, L, lane);
        diag4_panel_step<1>(n01, n11, L, lane);
        diag4_panel_step<2>(n01, n11, L, lane);
        diag4_panel_step<3>(n01, n11, L, lane);
        diag4_step<0>(n11, thrv, ind, lane, L->la[4], L->x1[4], L->piv + 16, &L->flagA[4]);
        diag4_step<1>(n11, thrv, ind, lane, L->la[5], L->x1[5], L->piv + 20, &L->flagA[5]);
        diag4_step<2>(n11, thrv, ind, lane, L->la[6], L->x1[6], L->piv + 24, &L->flagA[6]);
        diag4_step<3>(n11, thrv, ind, lane, L->la[7], L->x1[7], L->piv + 28, &L->flagA[7]);
    } else if (w == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) L->n01[r][lane] = nq[r];
        lds_flag_set(&L->flagN);
        d4 w00, w10 = {0, 0, 0, 0}, w11;
#pragma unroll
        for (int r = 0; r < 4; ++r) w00[r] = (fk + 4 * r == fr) ? 1.0 : 0.0;
        w11 = w00;
        diag4_inverse_step<0>(w00, w10, w11, L, lane);
        diag4_inverse_step<1>(w00, w10, w11, L, lane);
        diag4_inverse_step<2>(w00, w10, w11, L, lane);
        diag4_inverse_step<3>(w00, w10, w11, L, lane);
        diag4_inverse_step<4>(w00, w10, w11, L, lane);
        diag4_inverse_step<5>(w00, w10, w11, L, lane);
        diag4_inverse_step<6>(w00, w10, w11, L, lane);
        diag4_inverse_step<7>(w00, w10, w11, L, lane);
        if constexpr (TO_LDS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sLinv[(fk + 4 * r) * TLD + fr] = w00[r];
                sLinv[(16 + fk + 4 * r) * TLD + fr] = w10[r];
                sLinv[(16 + fk + 4 * r) * TLD + 16 + fr] = w11[r];
            }
        }
        if constexpr (TO_GLOBAL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double* p00 = &Li[(fk + 4 * r) * NB + fr];
                double* p10 = &Li[(16 + fk + 4 * r) * NB + fr];
                if constexpr (WT) {
                    __hip_atomic_store(p00, w00[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p10, w10[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p10 + 16, w11[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *p00 = w00[r];
                    *p10 = w10[r];
                    p10[16] = w11[r];
                }
            }
        }
    } else {  // w == 2
        if constexpr (OUT2) diag4_output(L, lane, Fd, ld, info, base, sd_blk, smax, pivtol);
    }
}

// this wave's quadrant (w >> 1, w & 1) of -T from a symmetric tile in LDS (leading dimension TLD), lower entries only
__device__ __forceinline__ d4 diag4_quadrant_from_lds(const double* T, int w, int lane) {
    const int fr = lane & 15, fk = lane >> 4, qi = w >> 1, qj = w & 1;
    d4 n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * qi + fk + 4 * r, col = 16 * qj + fr;
        n[r] = -(row >= col ? T[row * TLD + col] : T[col * TLD + row]);
    }
    return n;
}

__device__ __forceinline__ void tile_g2s(const double* __restrict__ g, long ld, double* s, int tid) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + 256 * q, r = e >> 5, cc = e & 31;
        s[r * TLD + cc] = g[(long)r * ld + cc];
    }
}

// ------------------------------------------------------------------------------------------
// K2  build the tall matrix G = [S; P- H^T; z^T; I].  Roles by blockIdx.y:
//   [0, Dp)             row of P- H^T                       (stencil gather along the row of P-)
//   [Dp, Dp+mp)         row of S = H P- H^T + R             (straight from P-, see s_entry)
//   [Dp+mp, Dp+2mp)     row of the trailing identity block  (the sweep turns it into Ls^-T)
// ------------------------------------------------------------------------------------------
template <typename PT>
__device__ __forceinline__ void front_block(const PT* __restrict__ Ppred, double* __restrict__ G,
                                            const double* __restrict__ rdiag, const double* __restrict__ Rdense,
                                            const MeasModel& mm, long Dp, int bx, long y) {
    const int tid = threadIdx.x;
    const int i = bx * 256 + tid;
    const int mp = mm.mp;
    const bool narrow = mm.w <= HW;
    if (y < Dp) {
        if (i < mp) {
            double v = 0.0;
            if (i < mm.m) {
                if (narrow) {
                    HRow rc;
                    h_row_load(mm, i, rc);
                    v = h_row_dot_w<HW>(rc, Ppred + y * Dp);
                } else {
                    v = h_row_dot(mm, i, Ppred + y * Dp);
                }
            }
            G[((long)mp + y) * mp + i] = v;
        }
        return;
    }
    if (y < Dp + mp) {
        const int ip = (int)(y - Dp);
        // Only the block-lower part of S is ever read (a row block of the sweep uses its tiles up to the diagonal one,
        // and inside a diagonal tile the entries right of the diagonal do not influence the factor): skip the rest.
        if (i < mp && i < (ip / NB + 1) * NB) {
            double v;
            if (narrow && ip < mm.m && i < mm.m) {
                HRow rr, rc;
                h_row_load(mm, ip, rr);
                h_row_load(mm, i, rc);
                v = s_entry_w<HW>(Ppred, Dp, mm, rr, rc, ip, i, rdiag, Rdense);
            } else {
                v = s_entry(Ppred, Dp, mm, ip, i, rdiag, Rdense);
            }
            G[(long)ip * mp + i] = v;
        }
        return;
    }
    const long q = y - Dp - mp;
    if (i < mp) G[((long)mp + Dp + NB + q) * mp + i] = (q == i) ? 1.0 : 0.0;
}

// The same rows of G by FEW, FAT workgroups (the constant-step loop's k_readout launch, where this gather is serial with the
// sweep): front_block's one-row blocks need (mp/256 rounded up) x (Dp + mp) = 6240 workgroups at N = 512 -- three rounds of
// resident blocks, each a chain of dependent loads (stencil row, then the gather) with one entry per thread: 12.7 us.  Here
//   blocks [0, Dp/8)                 8 rows of P- H^T: a thread keeps the stencil rows of ITS columns (i = tid, tid+256, ..)
//                                    and runs them over the 8 rows with all gathers of a row group in flight,
//   blocks [Dp/8, Dp/8 + CB(CB+1)/2) one block-lower 32x32 tile of S: four entries per thread.
// Same device functions, same association order: the entries are bit-identical to front_block's.
__host__ __device__ inline int front_tiled_blocks(long Dp, int mp) { return (int)(Dp / 8) + (mp / NB) * (mp / NB + 1) / 2; }
template <typename PT>
__device__ __forceinline__ void front_block_tiled(const PT* __restrict__ Ppred, double* __restrict__ G,
                                                  const double* __restrict__ rdiag, const double* __restrict__ Rdense,
                                                  const MeasModel& mm, long Dp, int fb) {
    const int tid = threadIdx.x, mp = mm.mp;
    const bool narrow = mm.w <= HW;
    const int nP = (int)(Dp / 8);
    if (fb < nP) {
        const long y0 = (long)fb * 8;
        for (int i = tid; i < mp; i += 256) {
            if (i >= mm.m) {
#pragma unroll
                for (int q = 0; q < 8; ++q) G[((long)mp + y0 + q) * mp + i] = 0.0;
                continue;
            }
            double v[8];
            if (narrow) {
                HRow rc;
                h_row_load(mm, i, rc);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = h_row_dot_w<HW>(rc, Ppred + (y0 + q) * Dp);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = h_row_dot(mm, i, Ppred + (y0 + q) * Dp);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) G[((long)mp + y0 + q) * mp + i] = v[q];
        }
        return;
    }
    int t = fb - nP;  // lower tile (I, J), t = I (I + 1) / 2 + J
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    const int J = t - I * (I + 1) / 2;
    const int i = J * NB + (tid & 31);
    HRow rc;
    if (narrow && i < mm.m) h_row_load(mm, i, rc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ip = I * NB + (tid >> 5) + 8 * q;
        double v;
        if (narrow && ip < mm.m && i < mm.m) {
            HRow rr;
            h_row_load(mm, ip, rr);
            v = s_entry_w<HW>(Ppred, Dp, mm, rr, rc, ip, i, rdiag, Rdense);
        } else {
            v = s_entry(Ppred, Dp, mm, ip, i, rdiag, Rdense);
        }
        G[(long)ip * mp + i] = v;
    }
}

// the identity block of G never changes (k_sweep only reads G): written once, at filter creation
__global__ void k_set_identity(double* __restrict__ Gi, int mp) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < mp) Gi[(long)q * mp + q] = 1.0;
}

__global__ __launch_bounds__(256) void k_front(const void* __restrict__ Ppred, double* __restrict__ G,
                                               const double* __restrict__ rdiag, const double* __restrict__ Rdense,
                                               MeasModel mm, long Dp, int p32) {
    if (p32) front_block(static_cast<const float*>(Ppred), G, rdiag, Rdense, mm, Dp, blockIdx.x, blockIdx.y);
    else front_block(static_cast<const double*>(Ppred), G, rdiag, Rdense, mm, Dp, blockIdx.x, blockIdx.y);
}

// S = H (P- H^T) + R from the rows of P- H^T that k_front has just written into G (a launch of its own behind it).  For
// wide stencils (2-d meshes: 5 neighbours + the derivative entry) s_entry's (w + 1)^2 = 36 dependent gathers per entry of S
// become w + 1 = 6 coalesced loads: G[mp + y][i] IS h_row_dot(i, row y of P-), the inner sum s_entry forms, so the entries
// are the same numbers in the same association order.  Block-lower part only, identity on the padding (as front_block).
__global__ __launch_bounds__(256) void k_front_s(double* __restrict__ G, const double* __restrict__ rdiag,
                                                 const double* __restrict__ Rdense, MeasModel mm) {
    const int i = blockIdx.x * 256 + threadIdx.x, ip = blockIdx.y, mp = mm.mp;
    if (i >= mp || i >= (ip / NB + 1) * NB) return;
    double v;
    if (ip >= mm.m || i >= mm.m) {
        v = (ip == i) ? 1.0 : 0.0;
    } else {
        const double* PHt = G + (long)mp * mp;
        v = 0.0;
        if (ip < mm.d) v = mm.c1 * PHt[(long)(mm.dp + ip) * mp + i];
        for (int e = 0; e < mm.w; ++e) {
            const int cidx = mm.ell_col[e * mp + ip];
            if (cidx >= 0) v += mm.c0 * mm.ell_val[e * mp + ip] * PHt[(long)cidx * mp + i];
        }
        if (ip == i) v += rdiag[i];
        if (Rdense) v += Rdense[(long)ip * mp + i];
    }
    G[(long)ip * mp + i] = v;
}

// first diagonal block: diag(S) -> sdiag, its max -> sdiag[mp]; F[0,0] = chol(G[0,0]), Linv[0] = its inverse
__global__ __launch_bounds__(128) void k_diag0(const double* __restrict__ G, double* __restrict__ F,
                                               double* __restrict__ Linv, int ld, int* info_base,
                                               double* __restrict__ sdiag, const int* __restrict__ ctr) {
    __shared__ double sT[NB * TLD];
    __shared__ __attribute__((aligned(16))) Diag2wLds dl;
    __shared__ double smx[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double smax = 0.0;
    for (int e = tid; e < ld; e += 128) {
        const double x = G[(long)e * ld + e];
        sdiag[e] = x;
        smax = fmax(smax, fabs(x));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o));
    if (lane == 0) smx[wave] = smax;
    if (tid < NB) dl.flag[tid] = 0;
    for (int e = tid; e < NB * NB; e += 128) sT[(e >> 5) * TLD + (e & 31)] = G[(long)(e >> 5) * ld + (e & 31)];
    __syncthreads();
    smax = fmax(smx[0], smx[1]);
    if (tid == 0) sdiag[ld] = smax;
    diag2w_from_lds(sT, F, ld, Linv, wave, lane, info_base + (*ctr - 1), 0, sdiag, smax, &dl);
}

// panel j:  L_Ij = G_Ij Linv_j^T  for all row blocks I > j  (written to F by the c == 0 column),
//           G_IK -= L_Ij L_Kj^T   for trailing column blocks K = j+1+c,
//           and the workgroup owning (j+1, j+1) factorises it for the next panel.
__global__ __launch_bounds__(256) void k_panel(double* __restrict__ G, double* __restrict__ F,
                                               double* __restrict__ Linv, int ld, int j, int CB, int RBS,
                                               int* info_base, const double* __restrict__ sdiag,
                                               const int* __restrict__ ctr) {
    // +2200 doubles of padding: A/B on one MI355X shows the latency-bound diagonal waves run ~2 % faster with at most
    // three of these workgroups per CU (42.9 KB each) than with six (25 KB)
    __shared__ __attribute__((aligned(16))) double smem_p[3 * NB * TLD + 2200];
    double* sA = smem_p;
    double* sB = smem_p + NB * TLD;
    double* sI = smem_p + 2 * NB * TLD;
    static_assert(sizeof(Diag2wLds) <= 2 * NB * TLD * sizeof(double), "Diag2wLds must fit in sB + sI");
    Diag2wLds* dlp = reinterpret_cast<Diag2wLds*>(sB);  // sB and sI are free once the diagonal tile sits in sA
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int I = j + 1 + blockIdx.x;
    const int c = blockIdx.y;
    const int Kc = j + 1 + c;
    const bool trailing = Kc < CB;
    if (trailing && I < RBS && I < Kc) return;  // strictly-upper tile of the symmetric part

    const int wr = w >> 1, wc = w & 1;
    const int fr = l & 15, fk = l >> 4;
    // issue the loads of the tile to be updated first: their latency hides behind the panel product
    double* gt = G + (long)I * NB * ld + (long)Kc * NB;
    d4 acc = {0, 0, 0, 0};
    if (trailing) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = gt[(long)(wr * 16 + fk + 4 * r) * ld + wc * 16 + fr];
    }
    tile_g2s(Linv + (long)j * NB * NB, NB, sI, tid);
    tile_g2s(G + (long)I * NB * ld + (long)j * NB, ld, sA, tid);
    if (trailing) tile_g2s(G + (long)Kc * NB * ld + (long)j * NB, ld, sB, tid);
    __syncthreads();

    d4 li = {0, 0, 0, 0}, lk = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < NB / 4; ++s) {
        const double b = sI[(wc * 16 + fr) * TLD + 4 * s + fk];
        li = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[(wr * 16 + fr) * TLD + 4 * s + fk], b, li, 0, 0, 0);
        if (trailing) lk = __builtin_amdgcn_mfma_f64_16x16x4f64(sB[(wr * 16 + fr) * TLD + 4 * s + fk], b, lk, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = wr * 16 + fk + 4 * r, col = wc * 16 + fr;
        sA[row * TLD + col] = li[r];
        if (trailing) sB[row * TLD + col] = lk[r];
        if (c == 0) F[((long)I * NB + row) * ld + (long)j * NB + col] = li[r];
    }
    if (!trailing) return;
    __syncthreads();

#pragma unroll
    for (int s = 0; s < NB / 4; ++s)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sA[(wr * 16 + fr) * TLD + 4 * s + fk],
                                                   sB[(wc * 16 + fr) * TLD + 4 * s + fk], acc, 0, 0, 0);
    if (!(blockIdx.x == 0 && c == 0)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[(long)(wr * 16 + fk + 4 * r) * ld + wc * 16 + fr] = acc[r];
        return;
    }
    // next diagonal block (j+1, j+1): wave 0 factorises it for the next panel
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) sA[(wr * 16 + fk + 4 * r) * TLD + wc * 16 + fr] = acc[r];
    if (tid < NB) dlp->flag[tid] = 0;
    __syncthreads();
    if (w < 2)  // waves 0 and 1: factor / inverse (two-wave scheme above)
        diag2w_from_lds(sA, F + ((long)(j + 1) * NB) * ld + (long)(j + 1) * NB, ld, Linv + (long)(j + 1) * NB * NB, w, l,
                        info_base + (*ctr - 1), (j + 1) * NB, sdiag + (j + 1) * NB, sdiag[ld], dlp);
}

// ------------------------------------------------------------------------------------------
// K3'  The whole sweep  G = [S; P-H^T; z^T; I]  ->  F = [Ls; W; r^T; Ls^-T]  in ONE launch (dataflow, left-looking).
//
// One workgroup per 32-row block I of the tall matrix (grid = RT), dependencies pass through global memory:
//   row[I]  = number of tiles L_I0 .. of row block I published (chain rows only),
//   diag[I] = 1 once L_II^-1 is published (and row I is complete).
// Step j of row block I:   S_j = G_Ij - sum_{k<j} X_k L_jk^T   (needs row j of L;  X_k = F_Ik, own row)
//                          X_j = S_j L_jj^-T                    (needs diag[j])
// Chain rows (I < CB, the rows of S) stop at j = I-1, keep D = G_II - sum X_k X_k^T in registers and then factorise
// D with the two-wave scheme above; the other rows (P-H^T, z, identity) run all CB steps.  G is only read.
// Only  diag[j] -> (load L_jj^-1, two 8-MFMA products) -> factor(j+1) -> diag[j+1]  is on the critical path:
// everything that needs only row j of L (published BEFORE block j is factorised) happens while block j is being
// factorised, so a panel costs the 32x32 factorisation + one flag hop (~1.2 us, tools/flag_hop_bench.hip) instead of
// a kernel boundary + a cold reload of the tiles.
// Coherence protocol (measured in tools/flag_hop_bench.hip; the XCDs' L2s are not coherent with each other):
// producers write shared tiles with agent-scope (write-through, sc1) stores, wait for them (s_waitcnt) and then set
// the flag; consumers poll with agent-scope loads and execute an agent-scope acquire fence (buffer_inv sc1) before
// touching the data.
// Forward progress.  In k_sweep every dependency points to a LOWER block index, and each XCD dispatches its workgroups in
// index order (block b goes to XCD b mod 8) -- but the eight queues advance independently, so "the producer was dispatched
// before me" holds per XCD only.  In k_sweep_rl one dependency even points upwards: the chain workgroup (block 0) waits for
// the feed of row block J+1.  With ONE sweep on the device every workgroup is resident from the start (RT + pairs <= 256
// CUs) and none of this matters.  With SEVERAL sweeps in flight a workgroup can spin on a producer that has no CU yet; it
// then gets one whenever any workgroup on that XCD retires, which the workgroups that do not wait (or whose producers are
// resident) keep doing -- in practice sixteen sweeps in flight run through (tests/test_gpu_parity.py::
// test_sixteen_problems_in_flight), but it is not a proof: every spin is therefore bounded (SWEEP_SPIN_LIMIT polls, about
// 0.2 s), a sweep that exceeds it reports `info` = -2 (return code -2, the step is lost, nothing hangs), and a caller that
// wants a guarantee keeps at most one sweep per 256 / (RT + pairs) of the device in flight.
// ------------------------------------------------------------------------------------------
constexpr int SWEEP_SPIN_LIMIT = 1 << 18;
#ifdef PNMOL_SWEEP_STAMP
__device__ long long pnmol_sweep_stamp[512][8];
// (L.stamp_id: the workgroup's index in the layout  0 = chain workgroup, 1 + I = row block I / down-date pair I - RT,
//  whatever the block mapping of the launch)
#define SWEEP_STAMP(slot) do { if (tid == 0 && L.stamp_id < 256) pnmol_sweep_stamp[L.stamp_id][slot] = wall_clock64(); } while (0)
#define SWEEP_STAMP_L(slot) do { if (l == 0 && L.stamp_id < 256) pnmol_sweep_stamp[L.stamp_id][slot] = wall_clock64(); } while (0)
// per-step trace of ONE workgroup (block PNMOL_SWEEP_TRACE_WG): rows 256 + j
#ifndef PNMOL_SWEEP_TRACE_WG
#define PNMOL_SWEEP_TRACE_WG 16
#endif
#define SWEEP_TRACE(j, slot) do { if (tid == 0 && L.stamp_id == PNMOL_SWEEP_TRACE_WG) pnmol_sweep_stamp[256 + (j)][slot] = wall_clock64(); } while (0)
// the publishing wave (w == 2) of the traced workgroup: rows 320 + j
#define SWEEP_TRACE_W2(j, slot) do { if (w == 2 && l == 0 && L.stamp_id == PNMOL_SWEEP_TRACE_WG) pnmol_sweep_stamp[320 + (j)][slot] = wall_clock64(); } while (0)
// the chain workgroup of k_sweep_rl (block 0): rows 384 + J
// (stamped builds only) wait for this wave's outstanding loads, then stamp: separates load latency from what follows
#define SWEEP_TRACE_VM(j, slot) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SWEEP_TRACE(j, slot); } while (0)
#define CHAIN_TRACE(J, slot) do { if (tid == 0) pnmol_sweep_stamp[384 + (J)][slot] = wall_clock64(); } while (0)
#define CHAIN_TRACE_W(J, slot, wave) do { if (l == 0 && w == (wave)) pnmol_sweep_stamp[384 + (J)][slot] = wall_clock64(); } while (0)
// shader-clock counter (s_memtime) next to the constant 100 MHz one: their ratio is the clock the chain workgroup's CU ran at
#define CHAIN_TRACE_CLK(J, slot) do { if (tid == 0) pnmol_sweep_stamp[384 + (J)][slot] = clock64(); } while (0)
#define PNMOL_DD_DBG 1
// one down-date workgroup (pair 60): rows 448 + j
#define DD_TRACE(j, slot) do { if (tid == 0 && pair == 60) pnmol_sweep_stamp[448 + (j)][slot] = wall_clock64(); } while (0)
#else
#define DD_TRACE(j, slot) do {} while (0)
#define CHAIN_TRACE_CLK(J, slot) do {} while (0)
#define SWEEP_TRACE_VM(j, slot) do {} while (0)
#define SWEEP_TRACE_W2(j, slot) do {} while (0)
#define CHAIN_TRACE(J, slot) do {} while (0)
#define CHAIN_TRACE_W(J, slot, wave) do {} while (0)
#define SWEEP_TRACE(j, slot) do {} while (0)
#define SWEEP_STAMP(slot) do {} while (0)
#define SWEEP_STAMP_L(slot) do {} while (0)
#endif

struct SweepLds {
    double sS[2][NB * TLD];  // S_j (ping-pong by step parity), C layout; sS[0] is the diagonal tile at the end
    double sX[NB * TLD];     // X_j of the last step
    double sP[4][NB * TLD];  // per-wave partial sums of the k-split products; the factorisation's scratch at the end
    double sd[NB];           // |S_ii| of this block
    double red[4];
    int seen[3];             // row[j], row[j+1], diag[j] as last polled
    int dead, pub;
    int pubcnt;              // k_sweep_rl: waves whose stores of the tiles so far have drained (4 per step)
    int rdiag, rnew, rcol, rzb;  // k_sweep_rl: what wave 0 has seen in global memory (relay_wait_ge)
    int stamp_id;            // (timeline builds, -DPNMOL_SWEEP_STAMP) index of this workgroup's stamps
    Diag4Lds d4;             // the diagonal block's four-wave factorisation (flags zeroed at kernel start)
};
// the chain workgroup of k_sweep_rl (it factorises every diagonal block)
struct ChainLds {
    double sX[NB * TLD];      // X_{J,J-1}
    double sLinv[NB * TLD];   // L_{J-1,J-1}^-1, row-major (B operand of the next TRSM)
    double sFeedS[NB * TLD];  // S_{J,J-1} as fed by row block J
    double sFeedD[NB * TLD];  // -D'_J as fed by row block J (natural positions of the four quadrants)
    double sd[2][NB];         // |G_ii| of the block being factorised (by parity of J)
    double red[4];
    Diag4Lds d4[2];           // by parity of J: wave 2 writes L_JJ out of d4[J & 1] while block J+1 is being prepared
};

__device__ __forceinline__ int flag_ld(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_st(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wt_st(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Write-through (sc1) hand-over traffic of k_sweep_rl, 16 bytes per lane:
//  * an 8-byte sc1 store is a fabric write of its own per lane (2.7x the time per byte of the 16-byte form, MI355X guide),
//    and a store instruction with every other lane masked off is not coalesced either (measured: 114 cycles per 128-byte
//    line against <= 37 for full 1-KB instructions) -- so tiles leave through LDS, as whole rows, all lanes active;
//  * 8-byte sc1 LOADS are not safe for data another workgroup has published: they were served stale lines from this XCD's
//    L2 (the bytes of the previous launch's hand-over through the same buffer; the guide's table of validated hand-offs
//    lists dword / dwordx4 loads, "not dwordx2") -- 16-byte sc1 loads are;
//  * buffer intrinsics rather than inline asm: the compiler then counts these operations in its s_waitcnt vmcnt(N)
//    bookkeeping (an asm store it does not know of makes every later counted wait stricter than meant) and inserts the
//    wait states a write to the data registers of a >64-bit store needs.
typedef int v4i_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffffc, 0x00020000);
}
template <int AUX = 16 /* sc1: write-through */>
__device__ __forceinline__ void wt_st2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double a, double b) {
    const v4i_t v = {__double2loint(a), __double2hiint(a), __double2loint(b), __double2hiint(b)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, AUX);
}
// rows [row0, row0 + 4 NE) of a 32x32 tile from LDS (leading dimension TLD) to global memory (tile origin at byte offset
// `org` of the buffer, leading dimension ld), by one wave: NE full 1-KB store instructions.  AUX = 0: plain stores (the
// data stops in this XCD's L2: readers on the same XCD only)
template <int NE, int AUX = 16>
__device__ __forceinline__ void wt_rows_from_lds(__amdgpu_buffer_rsrc_t r, unsigned org, long ld, const double* s, int row0, int l) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = l + 64 * e, row = row0 + (idx >> 4), c2 = 2 * (idx & 15);
        wt_st2<AUX>(r, org + (unsigned)((row * ld + c2) * 8), s[row * TLD + c2], s[row * TLD + c2 + 1]);
    }
}
// A whole tile held by one wave (8 x 2 doubles per lane, taken from LDS once): stored twice by the chain workgroup, first
// into its XCD's L2, later written through for everybody else.
struct TileRegs {
    double v[16];
};
__device__ __forceinline__ void tile_regs_from_lds(TileRegs& t, const double* s, int l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = l + 64 * e, row = idx >> 4, c2 = 2 * (idx & 15);
        t.v[2 * e] = s[row * TLD + c2];
        t.v[2 * e + 1] = s[row * TLD + c2 + 1];
    }
}
template <int AUX>
__device__ __forceinline__ void tile_regs_store(const TileRegs& t, __amdgpu_buffer_rsrc_t r, unsigned org, long ld, int l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = l + 64 * e, row = idx >> 4, c2 = 2 * (idx & 15);
        wt_st2<AUX>(r, org + (unsigned)((row * ld + c2) * 8), t.v[2 * e], t.v[2 * e + 1]);
    }
}
// a flag word for pollers on the SAME XCD: plain store, it stops in the L2 the pollers' sc1 loads are served from
__device__ __forceinline__ void l2_flag_st(__amdgpu_buffer_rsrc_t rflags, int word, int v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, rflags, (unsigned)word * 4u, 0, 0);
}
// a 32x32 tile (row-major, leading dimension NB, at byte offset `org`) that ANOTHER workgroup has published, into LDS
// (leading dimension TLD), by one wave
__device__ __forceinline__ void sc1_tile_to_lds(double* s, __amdgpu_buffer_rsrc_t r, unsigned org, int l) {
    v4i_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = l + 64 * e;
        v[e] = __builtin_amdgcn_raw_buffer_load_b128(r, org + (unsigned)(((idx >> 4) * NB + 2 * (idx & 15)) * 8), 0, 16 /* sc1 */);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int idx = l + 64 * e, row = idx >> 4, c2 = 2 * (idx & 15);
        s[row * TLD + c2] = __hiloint2double(v[e][1], v[e][0]);
        s[row * TLD + c2 + 1] = __hiloint2double(v[e][3], v[e][2]);
    }
}
// all of this wave's outstanding memory operations (in particular its write-through stores) are complete
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// What a workgroup knows about the three flags step j depends on (uniform per-thread copies of SweepLds::seen).
struct SweepSeen {
    int rowj, rown, diag;
};

// Wait until  row[j] >= need_rowj,  row[j+1] >= need_rown  and  (need_diag ? diag[j] : true).  All three flags are
// polled together, so a workgroup that lags behind the chain pays one memory round trip per step, not three.
// Called by all threads.
__device__ __forceinline__ void sweep_wait(SweepLds* L, const int* frow, const int* fdiag, int* fabort, int j, int jn_ok,
                                           int need_rowj, int need_rown, bool need_diag, SweepSeen& sn, int tid) {
    if (sn.rowj >= need_rowj && sn.rown >= need_rown && (!need_diag || sn.diag)) return;
    __syncthreads();  // every thread has consumed the previous values
    if (tid == 0) {
        int r = sn.rowj, rn = sn.rown, d = sn.diag;
        if (!L->dead) {
            for (int spins = 0;; ++spins) {
                r = flag_ld(frow + j);
                rn = jn_ok ? flag_ld(frow + j + 1) : (1 << 30);
                d = flag_ld(fdiag + j);
                if (r >= need_rowj && rn >= need_rown && (!need_diag || d)) break;
                if (spins > SWEEP_SPIN_LIMIT || flag_ld(fabort)) {
                    L->dead = 1;
                    flag_st(fabort, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (L->dead) r = rn = 1 << 30, d = 1;  // stop waiting; the step is reported as failed through `info`
#ifdef PNMOL_SWEEP_INV_EVERY_POLL
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        L->seen[0] = r;
        L->seen[1] = rn;
        L->seen[2] = d;
    }
    __syncthreads();
    sn.rowj = L->seen[0];
    sn.rown = L->seen[1];
    sn.diag = L->seen[2];
}

// MFMA operand fragments straight from global memory.  The inner (k) index of the 16x16x4 MFMA is permuted: step s,
// lane group g = lane >> 4 takes column 8 g + s, so a lane reads 8 CONTIGUOUS doubles of its row (four 16-byte
// loads) instead of 8 strided ones; A and B use the same permutation, so the sum is the same set of products.
struct Frag8 {
    double v[8];
};
__device__ __forceinline__ void frag_ld(Frag8& f, const double* __restrict__ p /* &T[row][8 g], 16-byte aligned */) {
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double2 t = q[e];
        f.v[2 * e] = t.x;
        f.v[2 * e + 1] = t.y;
    }
}
// one 32x32 tile as two row-halves of fragments: rows (l & 15) and 16 + (l & 15)
struct FragTile {
    Frag8 lo, hi;
};
__device__ __forceinline__ void tile_ld(FragTile& t, const double* __restrict__ p, long ld) {
    frag_ld(t.lo, p);
    frag_ld(t.hi, p + 16 * ld);
}

// One wave per row of four matrix-vector products that all need the finished sweep (role of the extra
// blockIdx.y rows of k_downdate):
//   rows [0, Dp)          m = m- - W r                               (mean update, white.py:123)
//   rows [Dp, Dp+mp)      part[0][i] = r_i^2                         (whitened residual)
//                         part[1][i] = (Ls^-T z)_i^2                 (white.py:125 with the Cholesky factor)
//                         part[2][i] = z_i (Sq^-1 z)_i               (estimate_error, white.py:153-162)
struct VecArgs {
    const double* mpred;
    const double* r;
    const double* LinvT;
    const double* z;
    const double* Sqinv;
    double* mout;
    double* part;
};

__device__ __forceinline__ void vecops_rows(const VecArgs& va, const double* __restrict__ W, int mp, long Dp,
                                            long wave_row, int l) {
    const long row = wave_row;
    if (row < Dp) {
        double sacc = 0.0;
        for (int i = l; i < mp; i += 64) sacc += W[row * mp + i] * va.r[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
        if (l == 0) va.mout[row] = va.mpred[row] - sacc;
    } else if (row < Dp + mp) {
        const long q = row - Dp;
        double x = 0.0, y = 0.0;
        for (int i = l; i < mp; i += 64) {
            const double zi = va.z[i];
            x += va.LinvT[q * mp + i] * zi;
            if (va.Sqinv) y += va.Sqinv[q * mp + i] * zi;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            x += __shfl_xor(x, o);
            y += __shfl_xor(y, o);
        }
        if (l == 0) {
            va.part[q] = va.r[q] * va.r[q];
            va.part[mp + q] = x * x;
            va.part[2 * mp + q] = va.z[q] * y;
        }
    }
}

// rows [r0, r1), r1 - r0 <= R, of vecops_rows by one wave; same per-row lane-strided summation, but the loads of up
// to R rows x 4 column slices are issued before the first is used
template <int R>
__device__ __forceinline__ void vecops_rowsR(const VecArgs& va, const double* __restrict__ W, int mp, long Dp, long r0,
                                             long r1, int l) {
    double a[R], b[R];
#pragma unroll
    for (int q = 0; q < R; ++q) a[q] = b[q] = 0.0;
    // All loads are unconditional (clamped index, weight 0 outside) so that they are issued back to back: with
    // predicated loads the compiler emitted one exec-masked region per load and this latency-bound tail took 9-13 us.
    const double* src[R];
    const double* src2[R];
    bool isw[R];
    double w2[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        const long row = min(r0 + q, r1 - 1);
        isw[q] = row < Dp;
        src[q] = isw[q] ? W + row * mp : va.LinvT + (row - Dp) * mp;
        const bool has2 = !isw[q] && va.Sqinv != nullptr;
        src2[q] = has2 ? va.Sqinv + (row - Dp) * mp : src[q];  // (dummy: same row, weight 0)
        w2[q] = has2 ? 1.0 : 0.0;
    }
    for (int i0 = l; i0 < mp; i0 += 256) {
        double wv[R][4], sv[R][4], xv[4], zv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = i0 + 64 * t;
            const int ii = i < mp ? i : mp - 1;
            const double in = i < mp ? 1.0 : 0.0;
            xv[t] = in * va.r[ii];
            zv[t] = in * va.z[ii];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                wv[q][t] = src[q][ii];
                sv[q][t] = src2[q][ii];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < R; ++q) {
                a[q] += wv[q][t] * (isw[q] ? xv[t] : zv[t]);
                b[q] += w2[q] * sv[q][t] * zv[t];
            }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a[q] += __shfl_xor(a[q], o);
            b[q] += __shfl_xor(b[q], o);
        }
        const long row = r0 + q;
        if (l == 0 && row < r1) {
            if (row < Dp) {
                va.mout[row] = va.mpred[row] - a[q];
            } else if (row < Dp + mp) {
                const long k = row - Dp;
                va.part[k] = va.r[k] * va.r[k];
                va.part[mp + k] = a[q] * a[q];
                va.part[2 * mp + k] = va.z[k] * b[q];
            }
        }
    }
}

// This wave's share of  sum_{k0 <= k < k1} A_k B_k^T  (32x32 tiles; A_k at Abase + k NB, B_k at Bbase + k NB, both
// already offset to this lane's row and column slice): tiles k0 + w, k0 + w + 4, ...  The full 32x32 partial sum goes
// to `sp` (C layout, row stride TLD).  Loads are unconditional (clamped index) and ping-pong between two register
// sets, so the wait counts are static and the loads of the next tile stay in flight behind the 32 MFMAs of this one.
__device__ __forceinline__ void ksplit_partial(double* sp, const double* __restrict__ Abase, const double* __restrict__ Bbase,
                                               int k0, int k1, long ld, int w, int l) {
    const int fr = l & 15, fk = l >> 4;
    d4 p00 = {0, 0, 0, 0}, p01 = p00, p10 = p00, p11 = p00;
    FragTile a0, b0, a1, b1;
    auto mfma32 = [&](const FragTile& a, const FragTile& b) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            p00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.lo.v[s], b.lo.v[s], p00, 0, 0, 0);
            p01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.lo.v[s], b.hi.v[s], p01, 0, 0, 0);
            p10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.hi.v[s], b.lo.v[s], p10, 0, 0, 0);
            p11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.hi.v[s], b.hi.v[s], p11, 0, 0, 0);
        }
    };
    const int nk = (k0 + w < k1) ? (k1 - k0 - w + 3) >> 2 : 0;  // my tiles: k_t = k0 + w + 4 t, t < nk
    if (nk > 0) {
        tile_ld(a0, Abase + (long)(k0 + w) * NB, ld);
        tile_ld(b0, Bbase + (long)(k0 + w) * NB, ld);
    }
    int t = 0;
    for (; t + 2 <= nk; t += 2) {
        const long ka = k0 + w + 4 * (t + 1), kb = k0 + w + 4 * (t + 2 < nk ? t + 2 : nk - 1);
        tile_ld(a1, Abase + ka * NB, ld);
        tile_ld(b1, Bbase + ka * NB, ld);
        mfma32(a0, b0);
        tile_ld(a0, Abase + kb * NB, ld);
        tile_ld(b0, Bbase + kb * NB, ld);
        mfma32(a1, b1);
    }
    if (t < nk) mfma32(a0, b0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        sp[(fk + 4 * r) * TLD + fr] = p00[r];
        sp[(fk + 4 * r) * TLD + 16 + fr] = p01[r];
        sp[(16 + fk + 4 * r) * TLD + fr] = p10[r];
        sp[(16 + fk + 4 * r) * TLD + 16 + fr] = p11[r];
    }
}

// ------------------------------------------------------------------------------------------
// Helpers.  A row of S with a large index H falls behind the chain near the end: the one-step-ahead partial sum of its
// target step t runs over t-1 tile pairs (0.5 us each), more than a panel period allows.  Its OLD part (tiles
// k < kh = t-1-KMAX, published long ago) is therefore done by the workgroup of row CB-1-H, which finished its own
// row early (dependencies keep pointing to lower block indices).  Nobody ever blocks on an unscheduled helper: a target
// is claimed (CAS on claim[H][t]: 0 free, 1 helper computing, 2 helper done, 3 owner took it) by the helper only once
// its inputs are there, and an owner that finds it unclaimed simply does the old part itself.
// ------------------------------------------------------------------------------------------
constexpr int HELP_KMAX = 8;
__device__ __forceinline__ int helper_share(int H, int t, int CB) {
    const int h = CB - 1 - H;
    if (h >= H - 1 || t < h + 3) return 0;  // no helper for this row / helper not free that early
    const int kh = t - 1 - HELP_KMAX;
    return kh > 0 ? kh : 0;
}

__device__ __forceinline__ void vecops_rows4(const VecArgs& va, const double* __restrict__ W, int mp, long Dp, long r0,
                                             long r1, int l) {
    vecops_rowsR<4>(va, W, mp, Dp, r0, r1, l);
}

// Arguments of the covariance down-date role (workgroups RT .. RT + pairs - 1 of a fused launch)
struct DowndateArgs {
    const void* Ppred;    // P-  (Dp x Dp), fp64 or (p32) fp32
    void* Pout;           // P = P- - W W^T
    double* var;          // diag(P)
    int dp;               // padded points per derivative (multiple of 32)
    int vrows;            // rows of the vector ops (vecops_rows) each down-date workgroup does at its end
    VecArgs va;
    // constant-step loop: the epilogue also predicts the NEXT step's covariance, P-' = A P A^T + Q, in place of P-
    // (a lane holds all n x n derivative entries of its point pairs), so the loop's steps need no k_predict pass over
    // P; P itself is then only written by the step whose counter equals *last_ctr.
    void* Pnext;          // = Ppred (in place: a workgroup only reads its own tile of P-, at its start) or nullptr
    const double* Kg;     // K = Gamma Gamma^T (dp x dp)
    const int* last_ctr;  // step counter value of the last step of the call
    double A1[MAXN * MAXN], Q1[MAXN * MAXN];
    int p32;              // the covariance is fp32 (pnmol_filter_desc.dtype = 1): fp32 accumulators on v_mfma_f32_16x16x4_f32
    int last_ksteps;      // MFMA k-steps (of 8) of the LAST column block of W that hold real columns; 0 = all
                          // (m = 514 = 16 * 32 + 2: columns 2 .. 31 of block 16 are padding, exactly zero in W)
};

// accumulator / MFMA of the down-date in the covariance's element type
template <typename PT>
struct AccOf;
template <>
struct AccOf<double> {
    typedef d4 type;
    static __device__ __forceinline__ d4 mfma_neg(double a, double b, d4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(-a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int fk, int r) { return fk + 4 * r; }  // C/D row of register r, lane group fk
};
template <>
struct AccOf<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ type mfma_neg(double a, double b, type c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(-(float)a, (float)b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int fk, int r) { return 4 * fk + r; }  // (the f64 form is the odd one out)
};

// Down-date role: one workgroup per lower pair (J >= K) of 32-point tiles, all N x N derivative blocks.  Wave
// (qr, qc) owns the 16x16 quadrant (qr, qc) of each block: its accumulators start as the P- tile, every 32-column
// block j of W is subtracted as soon as the 2 N row blocks of W it needs have published step j (row[] counters), so
// the down-date rides along with the sweep on CUs the sweep does not use instead of following it.
// SHORT_LAST: honour dd.last_ksteps (k_sweep_rl only: in k_sweep the guarded MFMA loop costs the large problems 20 %,
// N = 1024 797 us against 660)
#ifdef PNMOL_DD_DBG
__device__ __forceinline__ void pnmol_dd_avail_dbg(int j, int avail, bool loaded) { pnmol_sweep_stamp[448 + j][3] = avail; pnmol_sweep_stamp[448 + j][4] = loaded; }
#else
__device__ __forceinline__ void pnmol_dd_avail_dbg(int, int, bool) {}
#endif
template <int N, typename PT, bool SHORT_LAST>
__device__ __forceinline__ void sweep_downdate_body(SweepLds& L, const DowndateArgs& dd, const double* F, int ld, int CB,
                                                    int RBS, const int* frow, int* fabort, int* info, int pair, int tid,
                                                    int l, int w, int slot) {
    const int fr = l & 15, fk = l >> 4, qr = w >> 1, qc = w & 1;
    const int T32 = dd.dp / NB;
    int J = (int)((sqrt(8.0 * pair + 1.0) - 1.0) * 0.5);  // pair = J (J + 1) / 2 + K
    while ((J + 1) * (J + 2) / 2 <= pair) ++J;
    while (J * (J + 1) / 2 > pair) --J;
    const int K = pair - J * (J + 1) / 2;
    const long Dp = (long)N * dd.dp;
    const double* W = F + (long)ld * ld;  // rows of W follow the ld (= mp) rows of Ls
    typedef typename AccOf<PT>::type acc_t;
    const PT* Ppred = static_cast<const PT*>(dd.Ppred);
    acc_t acc[N][N];
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[a][b][r] = Ppred[((long)a * dd.dp + J * NB + qr * 16 + AccOf<PT>::row(fk, r)) * Dp + (long)b * dd.dp + K * NB + qc * 16 + fr];
    // K[j,k] of the fused predict in the epilogue, requested NOW: asked for behind the last column block, its two levels of
    // cache misses were 1-2 us of the sweep's tail in every pair (the epilogue is on the step's critical path)
    acc_t kjk = {0, 0, 0, 0};
    if (dd.Pnext) {
#pragma unroll
        for (int r = 0; r < 4; ++r) kjk[r] = (PT)dd.Kg[(long)(J * NB + qr * 16 + AccOf<PT>::row(fk, r)) * dd.dp + K * NB + qc * 16 + fr];
    }
    int avail = 0;  // column blocks of W known to be complete for all 2 N row blocks
    // wait until column block j is there (all threads)
    auto wait_for = [&](int j) {
        if (j < avail) return;
        __syncthreads();
        if (w == 0) {
            int mn = 0;
            if (!L.dead) {
                for (int spins = 0;; ++spins) {
                    int v = 1 << 30;
                    if (l < 2 * N) {
                        const int a = l >> 1, tile = (l & 1) ? K : J;
                        v = flag_ld(frow + RBS + a * T32 + tile);
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
                    mn = v;
                    if (mn > j) break;
                    if (spins > SWEEP_SPIN_LIMIT || flag_ld(fabort)) {
                        if (l == 0) {
                            L.dead = 1;
                            flag_st(fabort, 1);
                        }
                        mn = 1 << 30;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            } else {
                mn = 1 << 30;
            }
            if (l == 0) L.seen[0] = mn;
        }
        __syncthreads();
        avail = L.seen[0];
    };
    auto load_panel = [&](int j, Frag8 (&fa)[N], Frag8 (&fb)[N]) {
#pragma unroll
        for (int a = 0; a < N; ++a) {
            frag_ld(fa[a], W + ((long)a * dd.dp + J * NB + qr * 16 + fr) * ld + (long)j * NB + 8 * fk);
            frag_ld(fb[a], W + ((long)a * dd.dp + K * NB + qc * 16 + fr) * ld + (long)j * NB + 8 * fk);
        }
    };
    // One column block: its fragments are in (fa, fb) or are loaded now; the NEXT block's are requested before this block's
    // 72 MFMAs whenever it is known to be complete already (a workgroup that has fallen behind the sweep -- they all
    // do at the end: the last panels arrive every 6 us and took 6 us each with the load latency exposed -- catches up at the
    // MFMA rate).  Returns whether (na, nb) hold block j+1.
    auto panel = [&](int j, bool loaded, Frag8 (&fa)[N], Frag8 (&fb)[N], Frag8 (&na)[N], Frag8 (&nb)[N]) {
        DD_TRACE(j, 0);
        if (!loaded) {
            wait_for(j);
            DD_TRACE(j, 1);
            load_panel(j, fa, fb);
        }
        const bool next = j + 1 < CB && j + 1 < avail;
        if (next) load_panel(j + 1, na, nb);
        // (k-step s of the permuted inner index covers columns s, 8 + s, 16 + s, 24 + s of the block)
        const int smax = (SHORT_LAST && j + 1 == CB && dd.last_ksteps > 0) ? dd.last_ksteps : 8;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            if (!SHORT_LAST || s < smax) {
#pragma unroll
                for (int a = 0; a < N; ++a)
#pragma unroll
                    for (int b = 0; b < N; ++b)
                        acc[a][b] = AccOf<PT>::mfma_neg(fa[a].v[s], fb[b].v[s], acc[a][b]);
            }
        DD_TRACE(j, 2);
#ifdef PNMOL_SWEEP_STAMP
        if (tid == 0 && pair == 60) pnmol_dd_avail_dbg(j, avail, loaded);
#endif
        return next;
    };
    if constexpr (SHORT_LAST) {  // (k_sweep_rl; the large problems' k_sweep keeps the plain loop below)
        // Round 3: every WAVE follows the row flags by itself (its fragments come straight from global memory: nothing in
        // this loop is shared between the waves), so a column block costs no workgroup barrier; and the flags of block
        // j+1 are polled -- one non-blocking load -- at the top of block j and looked at two k-steps into its MFMAs, so the
        // fragments of block j+1 are in flight under the other six.  Round 2's loop refreshed its knowledge of the flags
        // only when it had to wait, i.e. it re-synchronised (two barriers and a poll round trip, 0.56 us) and then loaded
        // with nothing to hide the latency behind on every block once it lagged behind the sweep -- which, 72 MFMAs per
        // block at the one-wave-per-SIMD rate against a chain period of 4.9 us, is always (profiles/r03_microbench).
        (void)panel;
        (void)wait_for;
        // (Only wave 0 touches the flags in memory -- a flag word is served at the memory side, ~12 ns per poll per address,
        // and 64 waves polling each word made the rows of W themselves 10 us slower -- and relays what it has seen through
        // one LDS word, L.seen[0]: the number of column blocks known to be complete, monotone.)
        bool wdead = false;
        int* relay = &L.seen[0];
        auto poll_min = [&]() {  // wave 0: smallest published column count over the 2 N row blocks of W this pair reads
            int v = 1 << 30;
            if (l < 2 * N) {
                const int a = l >> 1, tile = (l & 1) ? K : J;
                v = flag_ld(frow + RBS + a * T32 + tile);
            }
            return v;
        };
        auto reduce_min = [&](int v) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
            return v;
        };
        auto publish = [&](int v) {  // wave 0
            if (l == 0) __hip_atomic_store(relay, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto known = [&]() { return __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
        auto wave_wait = [&](int j) {
            if (wdead) return;
            if (w == 0) {
                for (int spins = 0;; ++spins) {
                    const int v = reduce_min(poll_min());
                    if (v > j) {
                        publish(v);
                        return;
                    }
                    if (spins > SWEEP_SPIN_LIMIT || ((spins & 63) == 63 && flag_ld(fabort))) {
                        wdead = true;
                        flag_st(fabort, 1);
                        publish(1 << 30);
                        return;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
            } else {
                int v;
                while ((v = known()) <= j) __builtin_amdgcn_s_sleep(2);
                if (v == (1 << 30)) wdead = true;
            }
        };
        // one column block: fragments in (fa, fb); those of block j+1 go to (na, nb) if its flags are up
        auto block = [&](int j, bool have, Frag8 (&fa)[N], Frag8 (&fb)[N], Frag8 (&na)[N], Frag8 (&nb)[N]) {
            DD_TRACE(j, 0);
            if (!have) {
                wave_wait(j);
                DD_TRACE(j, 1);
                load_panel(j, fa, fb);
            }
            const int pv = (w == 0 && j + 1 < CB) ? poll_min() : 0;  // (wave 0: requested now, looked at below)
            const int smax = (j + 1 == CB && dd.last_ksteps > 0) ? dd.last_ksteps : 8;
            bool next = false;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s == 2) {
                    if (w == 0 && j + 1 < CB && !wdead) {
                        const int v = reduce_min(pv);
                        if (v > known()) publish(v);
                    }
                    const int kv = known();
                    next = (j + 1 < CB) && !wdead && kv > j + 1 && kv != (1 << 30);
                    if (next) load_panel(j + 1, na, nb);
                }
                if (s < smax) {
#pragma unroll
                    for (int a = 0; a < N; ++a)
#pragma unroll
                        for (int b = 0; b < N; ++b)
                            acc[a][b] = AccOf<PT>::mfma_neg(fa[a].v[s], fb[b].v[s], acc[a][b]);
                }
            }
            DD_TRACE(j, 2);
            return next;
        };
        Frag8 fa0[N], fb0[N], fa1[N], fb1[N];
        bool have = false;
        for (int j = 0; j < CB; j += 2) {
            have = block(j, have, fa0, fb0, fa1, fb1);
            if (j + 1 < CB) have = block(j + 1, have, fa1, fb1, fa0, fb0);
        }
        if (wdead && l == 0) atomicMin(info, -2);
        __syncthreads();  // (the epilogue's staging buffers are per wave, but L.dead / the vector-op tail below are shared)
    } else {
        for (int j = 0; j < CB; ++j) {
            wait_for(j);
            Frag8 fa[N], fb[N];
            load_panel(j, fa, fb);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int a = 0; a < N; ++a)
#pragma unroll
                    for (int b = 0; b < N; ++b)
                        acc[a][b] = AccOf<PT>::mfma_neg(fa[a].v[s], fb[b].v[s], acc[a][b]);
        }
    }
    SWEEP_STAMP(1);
    // epilogue: the tile, diag(P), and (J != K) the mirror image, transposed through wave-private LDS so that it
    // leaves as 128-byte rows; in the constant-step loop also (or, except for the last step, instead) the next
    // step's predicted covariance
    PT* stg = reinterpret_cast<PT*>(L.sP[w]);
    const bool write_p = dd.Pnext == nullptr || (slot + 1 == *dd.last_ctr);
    auto put_block = [&](PT* dst, const acc_t& v, long row0, long col0, bool mirror) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dst[(row0 + AccOf<PT>::row(fk, r)) * Dp + col0 + fr] = v[r];
            if (J != K && mirror) stg[AccOf<PT>::row(fk, r) * 17 + fr] = v[r];
        }
        if (J != K && mirror) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r)  // stg[i][c] = tile(i, c); mirror row c (= fk + 4 r) holds tile(:, c)
                dst[(col0 + fk + 4 * r) * Dp + row0 + fr] = stg[fr * 17 + fk + 4 * r];
            __builtin_amdgcn_wave_barrier();
        }
    };
#pragma unroll
    for (int a = 0; a < N; ++a) {
        const long row0 = (long)a * dd.dp + J * NB + qr * 16;
        if (J == K && qr == qc) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (AccOf<PT>::row(fk, r) == fr) dd.var[row0 + fr] = acc[a][a][r];
        }
        if (write_p) {
#pragma unroll
            for (int b = 0; b < N; ++b) put_block(static_cast<PT*>(dd.Pout), acc[a][b], row0, (long)b * dd.dp + K * NB + qc * 16, true);
        }
    }
    if (dd.Pnext) {  // P-' = A1 X A1^T + Q1 K[j,k] per point pair, X = the n x n entries this lane holds
        acc_t T[N][N];
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int e = 0; e < N; ++e) {
                acc_t t = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < N; ++q) t += (PT)dd.A1[a * MAXN + q] * acc[q][e];
                T[a][e] = t;
            }
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) {
                acc_t pn = (PT)dd.Q1[a * MAXN + b] * kjk;
#pragma unroll
                for (int e = 0; e < N; ++e) pn += T[a][e] * (PT)dd.A1[b * MAXN + e];
                // The mirror image of block (a, b) lands in COLUMNS of derivative a of an upper tile.  P-' has two readers:
                // the pairs of the next step (lower tiles only) and the gather of the next G = [H P- H^T + R; P- H^T],
                // whose H touches the columns of derivatives 0 and 1 (h_row_load: (1, i) and the stencil (0, col)).  Nobody
                // reads the upper tiles' columns of derivative >= 2: a sixth of this epilogue's 18.9 MB (n = 3) stays home.
                put_block(static_cast<PT*>(dd.Pnext), pn, (long)a * dd.dp + J * NB + qr * 16, (long)b * dd.dp + K * NB + qc * 16, a < 2);
            }
    }
    SWEEP_STAMP(2);
    // The vector ops of the step ride at the end of this role (rows [pair * vrows, +vrows) of vecops_rows): by now the
    // whole sweep is finished or about to be; a row still waits for the row block it reads and for the r^T block.
    {
        const int RBW = (int)(Dp / NB);
        const long row_lo = (long)pair * dd.vrows, row_hi = min(row_lo + dd.vrows, Dp + (long)ld);
        __syncthreads();
        if (tid == 0 && !L.dead && row_lo < row_hi) {
            for (long rb = row_lo / NB; rb <= (row_hi - 1) / NB; ++rb) {  // row blocks of [W; Ls^-T] this range touches
                const int need = rb < RBW ? RBS + (int)rb : RBS + RBW + 1 + (int)(rb - RBW);
                for (int spins = 0;; ++spins) {
                    if (flag_ld(frow + need) >= CB && flag_ld(frow + RBS + RBW) >= CB) break;
                    if (spins > SWEEP_SPIN_LIMIT || flag_ld(fabort)) {
                        L.dead = 1;
                        flag_st(fabort, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (L.dead) break;
            }
        }
        __syncthreads();
        // four rows per wave at a time with all loads of a slice of columns in flight (this tail is latency-bound:
        // one workgroup per CU, nothing else to switch to)
        for (long rg = row_lo + 4 * w; rg < row_hi; rg += 16) vecops_rows4(dd.va, W, ld, Dp, rg, min(rg + 4, row_hi), l);
    }
    if (tid == 0 && L.dead) atomicMin(info, -2);
    SWEEP_STAMP(5);
}

template <int N, bool SHORT_LAST = false>
__device__ __forceinline__ void sweep_downdate_role(SweepLds& L, const DowndateArgs& dd, const double* F, int ld, int CB,
                                                    int RBS, const int* frow, int* fabort, int* info, int pair, int tid,
                                                    int l, int w, int slot) {
    if (dd.p32) sweep_downdate_body<N, float, SHORT_LAST>(L, dd, F, ld, CB, RBS, frow, fabort, info, pair, tid, l, w, slot);
    else sweep_downdate_body<N, double, SHORT_LAST>(L, dd, F, ld, CB, RBS, frow, fabort, info, pair, tid, l, w, slot);
}

template <int N, bool FUSED, bool CHAINHELP = FUSED>
__global__ __launch_bounds__(256) void k_sweep(const double* __restrict__ G, double* F, double* Linv, int ld, int CB,
                                               int RT, int* flags, int* info_base, const int* __restrict__ ctr,
                                               DowndateArgs dd, int* claim, double* hs_scratch, int lenient, int wskip) {
    __shared__ __attribute__((aligned(16))) SweepLds L;
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // w: wave-uniform (SGPR)
    const int wr = w >> 1, wc = w & 1, fr = l & 15, fk = l >> 4;
    // Block -> row block.  The r^T block (row block zb of the tall matrix) is dealt right behind the rows of S, before the
    // rows of W and of Ls^-T, which wait for it at their end (vector ops): dependencies keep pointing to lower blocks.
    // wskip (large problems, !FUSED): the launch has no workgroups for the rows of W -- W = (P- H^T) Ls^-T is one GEMM behind
    // this launch (k_w_gemm) -- : blocks [0, CB) rows of S, CB the r^T block, CB + 1 .. 2 CB the rows of Ls^-T.
    int I = blockIdx.x;
#ifdef PNMOL_SWEEP_STAMP
    if (tid == 0) L.stamp_id = blockIdx.x;
#endif
    {
        const int zb = RT - CB - 1;
        if (wskip) {
            if (I >= CB) I = zb + (I - CB);
        } else if (I >= CB && I <= zb) {
            I = (I == CB) ? zb : I - 1;
        }
    }
    const bool chain = I < CB;
    int* frow = flags;            // [RT]  chain rows: tiles of the row published; other rows: steps completed
    int* fdiag = flags + RT;      // [CB]
    int* fabort = flags + RT + CB;
    int* info = info_base + (*ctr - 1);
    if constexpr (FUSED) {
        if (I >= RT) {
            if (tid == 0) {
                L.dead = 0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // see the note on the single invalidate below
            }
            __syncthreads();
            SWEEP_STAMP(0);
            sweep_downdate_role<N>(L, dd, F, ld, CB, CB, frow, fabort, info, I - RT, tid, l, w, *ctr - 1);
            return;
        }
    }

    double smax = 0.0;
    if (chain) {
        for (int e = tid; e < ld; e += 256) smax = fmax(smax, fabs(G[(long)e * ld + e]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o));
        if (l == 0) L.red[w] = smax;
        if (tid < NB) L.sd[tid] = G[(long)(I * NB + tid) * ld + I * NB + tid];
    }
    if (tid < 16) L.d4.flagA[tid] = 0;  // flagA[8], flagB[4], flagN, pad: contiguous
    if (tid == 0) {
        L.dead = 0, L.pub = 0;
        // One invalidate per workgroup: no copy of F / Linv from before this launch survives in this CU's L1 or this
        // XCD's L2.  Afterwards a tile is only ever read AFTER its producer's flag (first touch is fresh: tiles are
        // written once per launch, cache-line aligned, never read speculatively), so no further invalidates are needed
        // and the shared L tiles stay L2-resident per XCD instead of being re-fetched from memory by every workgroup.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    SWEEP_STAMP(0);
    if (chain) smax = fmax(fmax(L.red[0], L.red[1]), fmax(L.red[2], L.red[3]));
    // lenient (factor of a covariance that is PSD only up to the rounding of the filter recursion): a non-positive pivot
    // is a dropped direction, never an error -- only NaN is
    if (lenient) smax = __builtin_inf();

    const long rowC = (long)I * NB + wr * 16 + fk;  // first of my four C-layout rows (stride 4)
    const int colC = wc * 16 + fr;
    const int offC = (wr * 16 + fk) * TLD + colC;   // the same position in an LDS tile (rows stride 4 * TLD)
    const double* Xown = F + ((long)I * NB + fr) * ld + 8 * fk;  // A fragments of my own row block: X_k = F[I][k]
    // accD = this wave's quadrant (wr, wc) of -D, D = G_II - sum_k X_k X_k^T (negated: the updates are plain
    // accumulations, and it is the form diag4_factor takes).  Only the lower entries of G_II are read: quadrant
    // (0, 1) starts as the transpose of (1, 0) and stays it (its updates are the same products, transposed).
    d4 accD = {0, 0, 0, 0};
    if (chain) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = wr * 16 + fk + 4 * r, cl = colC;
            accD[r] = -G[((long)I * NB + (rl >= cl ? rl : cl)) * ld + (long)I * NB + (rl >= cl ? cl : rl)];
        }
    }

    const int nsteps = chain ? I : CB;
    d4 gnext = {0, 0, 0, 0};  // G_{I,j} for the coming step
    if (nsteps > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) gnext[r] = G[(rowC + 4 * r) * ld + colC];
    }
    SweepSeen sn{0, 0, 0};
    for (int j = 0; j < nsteps; ++j) {
        const int jn_ok = (j + 1 < CB);
        // (1) S_j = G_Ij - [k-split partial sums over k < j-1, made during step j-1] - X_{j-1} L_{j,j-1}^T
        d4 acc = gnext;
        SWEEP_TRACE(j, 0);
        if (j >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[r] -= (L.sP[0][offC + 4 * r * TLD] + L.sP[1][offC + 4 * r * TLD]) +
                          (L.sP[2][offC + 4 * r * TLD] + L.sP[3][offC + 4 * r * TLD]);
        }
        if (CHAINHELP && chain) {
            const int kh = helper_share(I, j, CB);
            if (kh > 0) {  // the old part of this step's sum: from the helper, or done here if it never got to it
                int* cl = claim + (long)I * CB + j;
                __syncthreads();
                if (tid == 0) {
                    int c = atomicCAS(cl, 0, 3);
                    for (int spins = 0; c == 1 && !L.dead; ++spins) {
                        c = flag_ld(cl);
                        if (spins > SWEEP_SPIN_LIMIT || flag_ld(fabort)) {
                            L.dead = 1;
                            flag_st(fabort, 1);
                        }
                    }
                    L.seen[0] = c;
                }
                __syncthreads();
                const int c = L.seen[0];
                if (c == 2) {
                    const double* hs = hs_scratch + ((long)I * CB + j) * NB * NB;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] -= hs[(wr * 16 + fk + 4 * r) * NB + colC];
                } else if (c == 0) {
                    ksplit_partial(L.sP[w], Xown, F + ((long)j * NB + fr) * ld + 8 * fk, 0, kh, ld, w, l);
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[r] -= (L.sP[0][offC + 4 * r * TLD] + L.sP[1][offC + 4 * r * TLD]) +
                                  (L.sP[2][offC + 4 * r * TLD] + L.sP[3][offC + 4 * r * TLD]);
                }
            }
        }
        if (j >= 1) {  // newest tile L_{j,j-1}; X_{j-1} is still in LDS
            sweep_wait(&L, frow, fdiag, fabort, j, jn_ok, j, 0, false, sn, tid);
            SWEEP_TRACE(j, 1);
            Frag8 b0;
            frag_ld(b0, F + ((long)j * NB + wc * 16 + fr) * ld + (long)(j - 1) * NB + 8 * fk);
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-L.sX[(wr * 16 + fr) * TLD + 8 * fk + s], b0.v[s], acc, 0, 0, 0);
        }
        double* sS = L.sS[j & 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) sS[offC + 4 * r * TLD] = acc[r];
        drain_vmem();  // my X stores of the previous step are complete (for the other waves' reads)
        __syncthreads();
        SWEEP_TRACE(j, 2);
        // (2) while the chain catches up: the part of step j+1 that needs only rows published so far,
        //     sum_{k<j} X_k L_{j+1,k}^T, k split over the four waves (each wave: full 32x32 tile, every 4th k)
        if (j + 1 < nsteps) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gnext[r] = G[(rowC + 4 * r) * ld + (long)(j + 1) * NB + colC];
            if (j >= 1) {
                sweep_wait(&L, frow, fdiag, fabort, j, jn_ok, 0, j, false, sn, tid);
                SWEEP_TRACE(j, 3);
                const int kh = CHAINHELP && chain ? helper_share(I, j + 1, CB) : 0;  // tiles [0, kh): a helper's
                ksplit_partial(L.sP[w], Xown, F + ((long)(j + 1) * NB + fr) * ld + 8 * fk, kh, j, ld, w, l);
            }
        }
        if (j + 1 == nsteps) SWEEP_STAMP(1);
        SWEEP_TRACE(j, 4);
        // (3) X_j = S_j L_jj^-T
        sweep_wait(&L, frow, fdiag, fabort, j, jn_ok, 0, 0, true, sn, tid);
        SWEEP_TRACE(j, 5);
        if (j + 1 == nsteps) SWEEP_STAMP(2);
        Frag8 bl;
        frag_ld(bl, Linv + (long)j * NB * NB + (wc * 16 + fr) * NB + 8 * fk);
        d4 x = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; ++s)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(sS[(wr * 16 + fr) * TLD + 8 * fk + s], bl.v[s], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) L.sX[offC + 4 * r * TLD] = x[r];
        sn = SweepSeen{sn.rown, 0, 0};  // flags of step j+1: row[j+1] is already known this far
        __syncthreads();                // sX (and sP) complete
        if (w == 2) {  // one wave publishes the tile; the stores complete behind the other waves' next instructions
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int idx = l + 64 * e, row = idx >> 5, col = idx & 31;
                wt_st(&F[((long)I * NB + row) * ld + (long)j * NB + col], L.sX[row * TLD + col]);
            }
        }
        if (chain) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(L.sX[(wr * 16 + fr) * TLD + 8 * fk + s],
                                                            L.sX[(wc * 16 + fr) * TLD + 8 * fk + s], accD, 0, 0, 0);
        }
        if ((!chain || j + 1 < nsteps) && w == 2) {  // (a chain row's last tile is published during its factorisation)
            drain_vmem();
            if (l == 0) flag_st(frow + I, j + 1);
        }
        SWEEP_TRACE(j, 6);
    }
    if (!chain) {
        if constexpr (FUSED) {
            // The vector ops of the step for the rows this workgroup has just finished (m = m- - W r for a block of W,
            // the |r|^2 / |Ls^-T z|^2 / z^T Sq^-1 z terms for a block of Ls^-T): they only wait for the r^T block.
            const int RBW = RT - 2 * CB - 1, zb = CB + RBW;
            if (I != zb) {
                SWEEP_STAMP(1);
                __syncthreads();  // the publishing wave has drained this row block's last tile
                if (tid == 0 && !L.dead) {
                    for (int spins = 0; flag_ld(frow + zb) < CB; ++spins) {
                        if (spins > SWEEP_SPIN_LIMIT || flag_ld(fabort)) {
                            L.dead = 1;
                            flag_st(fabort, 1);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __syncthreads();
                const long Dp = (long)RBW * NB;
                const long row0 = (I < zb ? (long)(I - CB) * NB : Dp + (long)(I - zb - 1) * NB) + 8 * w;
                const double* W = F + (long)ld * ld;
                SWEEP_STAMP(2);
                vecops_rowsR<8>(dd.va, W, ld, Dp, row0, row0 + 8, l);
            }
        }
        if (tid == 0 && L.dead) atomicMin(info, -2);
        SWEEP_STAMP(5);
        return;
    }

    // block (I, I): factorise, publish L_II^-1.  No barrier: every wave goes straight into its role of the four-wave
    // scheme (diag4_factor) with its quadrant of -D in registers.
    SWEEP_STAMP(3);
    if (w == 2 && I > 0) {
        drain_vmem();  // the last tile of row I has reached memory
        if (l == 0) flag_st(frow + I, I);
        __hip_atomic_store(&L.pub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    diag4_factor<true>(accD, &L.d4, w, l, F + ((long)I * NB) * ld + (long)I * NB, ld, Linv + (long)I * NB * NB, info, I * NB,
                       L.sd, smax, nullptr, pivot_tolerance(lenient));
    if (w == 1) {
        drain_vmem();  // L^-1 has reached memory
        if (I > 0) {
            int spins = 0;
            while (__hip_atomic_load(&L.pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0 && ++spins < (1 << 24))
                __builtin_amdgcn_s_sleep(1);
        }
        if (l == 0) flag_st(fdiag + I, 1);
        SWEEP_STAMP_L(4);
    }
    if (tid == 0 && L.dead) atomicMin(info, -2);
    if constexpr (CHAINHELP) {
        // this row is done: help row H = CB-1-I with the old part of its partial sums (see helper_share)
        const int H = CB - 1 - I;
        if (I < H - 1) {
            const double* Xh = F + ((long)H * NB + fr) * ld + 8 * fk;
            for (int t = I + 3; t < H; ++t) {
                const int kh = helper_share(H, t, CB);
                if (kh <= 0) continue;
                int* cl = claim + (long)H * CB + t;
                __syncthreads();
                if (tid == 0) {
                    int got = 0;  // 1: claimed, 0: the owner was faster, -1: inputs never came, stop helping
                    for (int spins = 0;; ++spins) {
                        if (flag_ld(cl) != 0) break;
                        if (flag_ld(frow + H) >= kh && flag_ld(frow + t) >= kh) {
                            got = atomicCAS(cl, 0, 1) == 0;
                            break;
                        }
                        if (spins > 4096 || flag_ld(fabort)) {
                            got = -1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    L.seen[0] = got;
                }
                __syncthreads();
                const int got = L.seen[0];
                if (got < 0) break;
                if (got == 0) continue;
                ksplit_partial(L.sP[w], Xh, F + ((long)t * NB + fr) * ld + 8 * fk, 0, kh, ld, w, l);
                __syncthreads();
                double* hs = hs_scratch + ((long)H * CB + t) * NB * NB;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wt_st(&hs[(wr * 16 + fk + 4 * r) * NB + colC],
                          (L.sP[0][offC + 4 * r * TLD] + L.sP[1][offC + 4 * r * TLD]) +
                              (L.sP[2][offC + 4 * r * TLD] + L.sP[3][offC + 4 * r * TLD]));
                drain_vmem();
                __syncthreads();
                if (tid == 0) flag_st(cl, 2);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K3''  The sweep, RIGHT-looking with the row block resident in registers (CB <= MAXT; the left-looking k_sweep above
// stays for wider matrices: 2-d meshes, the Cholesky factor of a whole covariance).
//
// Same roles, flags and coherence protocol as k_sweep.  What changes is the row-block workgroup: in k_sweep it makes,
// at step j, the whole sum  sum_{k<j} X_k L_jk^T  (j tile products, both operands from global memory: the late rows of S
// need 5.4 us per step and fall behind the chain).  Here the workgroup keeps ALL its tiles in MFMA accumulators
// (wave (wr, wc) holds the 16x16 quadrant of every tile, negated: n_t = -S_t) and, at step j, applies the rank-32 update
// of column panel j-1 to the tiles that are still to come:
//     n_t += X_{j-1} L_{t,j-1}^T,  t = j+1 .. last        (X_{j-1}: own tile, one LDS fragment for all t;
//                                                           L_{t,j-1}: tile (t, j-1) of chain row t, 4 KB per wave)
// so the work per step SHRINKS with j, every tile product is 8 MFMAs per wave with one operand stream, and nothing but
//     tile j:  n_j += X_{j-1} L_{j,j-1}^T  ->  X_j = S_j L_jj^-T  [ -> D -= X_j X_j^T -> factor ]
// waits for the newest data.  The slots rotate (slot u holds tile j + u: the update of slot u is written to slot u-1), so
// the step loop is a plain run-time loop over fixed registers -- at the price of a register move per accumulator at the
// loop's back-edge (0.5 us per step), which is why the feed of a row block of S is NOT a pass of that loop.  Two
// workgroup barriers per step (S_j complete in LDS, X_j complete in LDS), none around the polls: wave 0 polls and relays
// through LDS.  The update's operands stream through a per-wave LDS ring filled by LDS-DMA; with XL the chain workgroup
// and the row blocks of S share an XCD and hand over through its L2.  DESIGN.md section 3a'' has the measurements.
// ------------------------------------------------------------------------------------------
// Flags of k_sweep_rl.  A poll is an uncached (sc1) load served at the memory side, ~12 ns each PER ADDRESS: with every
// wave of 83 workgroups polling the same line (and the abort word beside it) a flag that had been set was seen 2-5 us
// late.  So: ONE wave per workgroup polls global memory and relays through LDS to the other three; the flags everybody
// waits for at the same time exist in RL_REP copies on lines of their own (a consumer polls copy blockIdx % RL_REP); the
// update of step j waits for ONE counter per column panel instead of up to 16 row flags (a row block of S for exactly the
// rows it reads); one poll reads a whole flag array and wave 0 remembers the finished prefix (relay_progress); the abort
// word is read every 64th poll only.
//   frow[RT]          row block I has published its own tiles 0 .. frow[I]-1       (down-date role; the r^T block)
//   fabort            somebody timed out
//   ffeed[CB]         row block J has fed S_{J,J-1} and -D'_J to the chain workgroup
//   fdiag[RL_REP][CBp]  L_JJ^-1 is published                                       (chain workgroup)
//   fnew[RL_REP][CBp]   tile (J+1, J) of F is published                            (chain workgroup, ~1 us later)
//   fcol[RL_REP][CBp]   number of chain rows that have published their own tile p  (atomic adds)
//   fxcc              1 + XCC id of the chain workgroup (XL)
//   fdiagL[CBp], fnewL[CBp]  copies of fdiag / fnew that are stored plainly, for the pollers on the chain workgroup's XCD (XL)
//   frowL[CBp]        frow of the row blocks of S for the same pollers; ahead of frow for a row's last ordinary tile (see there)
constexpr int RL_REP = 8;
// The late row blocks of S are split over two workgroups (XL layout only): row block I >= S_SPLIT_FROM is "part A", tiles
// 0 .. S_SPLIT_H-1 (an ordinary row block that stops there and publishes all its tiles), and "part B", tiles S_SPLIT_H .. I-1
// and the feed.  Why: a row block's step costs ~3.5 us + 0.45 us per tile it still has to update, so row block I needs
// 3.5 (I-1) + 0.45 I^2 / 2 us in all and has 5.5 I us until the chain workgroup wants its feed -- from I = 10 on it falls
// behind, and the chain's period grew from 5.3 us (J <= 2) to 7-12 us (J >= 10): profiles/r03_microbench/sweep_timeline_a_early_publication.log.
constexpr int S_SPLIT_FROM = 11, S_SPLIT_H = 9;
__host__ __device__ inline int s_split_count(int CB, bool xl) { return (xl && CB > S_SPLIT_FROM) ? CB - S_SPLIT_FROM : 0; }
// The row blocks of W are split the same way (both layouts): part A = tiles 0 .. W_SPLIT_H-1, part B = the rest, the vector
// ops.  A row block of W needs 17 x ~4.2 us of step overhead + 136 tile updates x 0.42 us = ~125 us for its 17 steps and
// was the last thing the sweep waited for once the chain had become faster; the halves need ~63 us each.  The parts B
// sit behind all other row blocks in the block order (they wait for their parts A), the down-date pairs behind them.
// MEASURED AND SWITCHED OFF (PNMOL_W_SPLIT=1 at build time turns it on): the halves need ~63 us each as predicted (parts A
// done at 78 us), but 48 more workgroups make 273 for 256 CUs -- 250 of them on the seven XCDs that are not the chain's --
// and the 26 down-date pairs that get a CU only when the parts A retire finish at 149 us: 160 us per step against 144
// (gpurun r3u).  The rows of W are made faster instead (eager bulk update, below).
constexpr int W_SPLIT_H = 11;
#ifndef PNMOL_W_SPLIT
#define PNMOL_W_SPLIT 0
#endif
__host__ __device__ inline int w_split_count(int RT, int CB) {
    const int RBW = RT - 2 * CB - 1;
    return (PNMOL_W_SPLIT && RBW > 0 && CB > W_SPLIT_H + 1) ? RBW : 0;
}
struct RlFlags {
    int frow, fabort, fxcc, ffeed, fdiag, fnew, fcol, fdiagL, fnewL, frowL, CBp, total;
};
__host__ __device__ inline RlFlags rl_flags(int RT, int CB) {
    const int RTp = (RT + 31) / 32 * 32, CBp = (CB + 31) / 32 * 32;
    RlFlags f;
    f.CBp = CBp;
    f.frow = 0;
    f.fabort = RTp;
    f.fxcc = RTp + 16;  // 1 + XCC id of the chain workgroup (half a line away from the abort word: read once per workgroup)
    f.ffeed = RTp + 32;
    f.fdiag = f.ffeed + CBp;
    f.fnew = f.fdiag + RL_REP * CBp;
    f.fcol = f.fnew + RL_REP * CBp;
    f.fdiagL = f.fcol + RL_REP * CBp;  // copies of fdiag / fnew that live in the chain workgroup's XCD (see XL below)
    f.fnewL = f.fdiagL + CBp;
    f.frowL = f.fnewL + CBp;  // frow of the row blocks of S, for the pollers on the chain workgroup's XCD (XL)
    f.total = f.frowL + CBp;
    return f;
}
// The XCC (XCD) this wave runs on.
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }
struct WaveWait {
    bool dead;
};
// every lane polls the same word (one request per poll); returns once *p >= need (or the launch is being aborted)
__device__ __forceinline__ void wave_wait_ge(const int* p, int need, int* fabort, WaveWait& ww) {
    if (ww.dead) return;
    for (int spins = 0;; ++spins) {
        if (flag_ld(p) >= need) return;
        if (spins > SWEEP_SPIN_LIMIT || ((spins & 63) == 63 && flag_ld(fabort))) {
            ww.dead = true;
            flag_st(fabort, 1);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
// Wave 0 waits for the global word and relays `tag` through the LDS word `relay` (monotone); the other waves wait for the
// relay.  An abort is relayed as a huge tag: nobody keeps waiting.
__device__ __forceinline__ void relay_wait_ge(const int* p, int need, int* relay, int tag, int* fabort, WaveWait& ww, int w) {
    if (w == 0) {
        wave_wait_ge(p, need, fabort, ww);
        __hip_atomic_store(relay, ww.dead ? (1 << 30) : tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        int v;
        while ((v = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < tag) __builtin_amdgcn_s_sleep(1);
        if (v == (1 << 30)) ww.dead = true;
    }
    asm volatile("" ::: "memory");
}

// Progress polls.  A row block that lags behind the chain workgroup would pay a memory round trip (0.3-1 us) per wait for
// flags that were set long ago.  One poll reads ALL the words of a flag array (lane k reads p[k]) and yields the length of
// the leading run of finished entries (entry k is finished when p[k] >= need(k)); wave 0 remembers it (`known`) and only
// polls again when a step needs more than it knows.  The relay word carries `known` to the other waves.
template <class Need>
__device__ __forceinline__ void relay_progress(const int* p, int cnt, Need need, int want, int& known, int* relay, int* fabort,
                                               WaveWait& ww, int w, int l) {
    if (w == 0) {
        if (known < want) {
            for (int spins = 0; !ww.dead; ++spins) {
                const int v = l < cnt ? flag_ld(p + l) : 0;
                const unsigned long long ok = __ballot(l < cnt && v >= need(l));
                known = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(~ok));
                if (known >= want) break;
                if (spins > SWEEP_SPIN_LIMIT || ((spins & 63) == 63 && flag_ld(fabort))) {
                    ww.dead = true;
                    flag_st(fabort, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(relay, ww.dead ? (1 << 30) : known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        int v;
        while ((v = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) __builtin_amdgcn_s_sleep(1);
        if (v == (1 << 30)) ww.dead = true;
    }
    asm volatile("" ::: "memory");
}
// The same for "every word of p[0 .. cnt-1] is >= want": wave 0 remembers the minimum it has seen (valid for later, shorter
// ranges that end at the same word and for smaller `want`).
__device__ __forceinline__ void relay_progress_min(const int* p, int cnt, int want, int& known, int* relay, int* fabort,
                                                   WaveWait& ww, int w, int l) {
    if (w == 0) {
        if (known < want) {
            for (int spins = 0; !ww.dead; ++spins) {
                int v = l < cnt ? flag_ld(p + l) : (1 << 29);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
                known = __builtin_amdgcn_readfirstlane(v);
                if (known >= want) break;
                if (spins > SWEEP_SPIN_LIMIT || ((spins & 63) == 63 && flag_ld(fabort))) {
                    ww.dead = true;
                    flag_st(fabort, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(relay, ww.dead ? (1 << 30) : known, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        int v;
        while ((v = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) __builtin_amdgcn_s_sleep(1);
        if (v == (1 << 30)) ww.dead = true;
    }
    asm volatile("" ::: "memory");
}
// The same for a RANGE of words: p[0 .. cnt-1] >= need, cnt <= 64 (one load instruction per poll: lane k reads p[k]).
__device__ __forceinline__ void relay_wait_range_ge(const int* p, int cnt, int need, int* relay, int tag, int* fabort,
                                                    WaveWait& ww, int w, int l) {
    if (w == 0) {
        if (!ww.dead) {
            for (int spins = 0;; ++spins) {
                const int v = l < cnt ? flag_ld(p + l) : need;
                if (__all(v >= need)) break;
                if (spins > SWEEP_SPIN_LIMIT || ((spins & 63) == 63 && flag_ld(fabort))) {
                    ww.dead = true;
                    flag_st(fabort, 1);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __hip_atomic_store(relay, ww.dead ? (1 << 30) : tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        int v;
        while ((v = __hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < tag) __builtin_amdgcn_s_sleep(1);
        if (v == (1 << 30)) ww.dead = true;
    }
    asm volatile("" ::: "memory");
}

// out[q] = sum_i base[(row0 + q) * mp + i] * vec[i], q = 0 .. 7, by one wave (every lane gets the sums).  All loads of a
// 256-column slice are issued back to back (see vecops_rowsR).
__device__ __forceinline__ void rows8_dot(const double* __restrict__ base, long row0, int mp, const double* __restrict__ vec,
                                          double (&out)[8], int l) {
#pragma unroll
    for (int q = 0; q < 8; ++q) out[q] = 0.0;
    for (int i0 = l; i0 < mp; i0 += 256) {
        double wv[8][4], xv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int i = i0 + 64 * t;
            const int ii = i < mp ? i : mp - 1;
            xv[t] = i < mp ? vec[ii] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q][t] = base[(row0 + q) * mp + ii];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] += wv[q][t] * xv[t];
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) out[q] += __shfl_xor(out[q], o);
}
// The vector ops of eight rows of [W; Ls^-T] (see vecops_rows) with one matrix row read per output row: rows of W need W r
// only, and for the rows of Ls^-T the error-model product Sq^-1 z -- a constant matrix times a vector that is known when
// the sweep starts -- comes in as bq (made at the start of the launch by the same wave, off the tail).
__device__ __forceinline__ void vecops_rows8(const VecArgs& va, const double* __restrict__ W, int mp, long Dp, long r0,
                                             const double* bq /* LDS, 8 values */, int l) {
    double a[8];
    if (r0 < Dp) {
        rows8_dot(W, r0, mp, va.r, a, l);
        if (l < 8) {
            double mine = a[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) mine = (l == q) ? a[q] : mine;
            va.mout[r0 + l] = va.mpred[r0 + l] - mine;
        }
    } else {
        const long k0 = r0 - Dp;
        rows8_dot(va.LinvT, k0, mp, va.z, a, l);
        if (l < 8) {
            double mine = a[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) mine = (l == q) ? a[q] : mine;
            const double bm = bq[l];
            const long k = k0 + l;
            va.part[k] = va.r[k] * va.r[k];
            va.part[mp + k] = mine * mine;
            va.part[2 * mp + k] = va.z[k] * bm;
        }
    }
}
// hipcc (ROCm 7.2) hazard: where a chain of MFMAs ends a conditional block, the s_nop that must separate the last MFMA
// from a v_accvgpr_read / v_accvgpr_mov of its result can end up BEHIND the first reads in the block the branch joins
// (seen twice in k_sweep_rl: accumulator element 3 -- rows fk + 12 of a tile -- stale, only in builds without the timeline
// stamps, whose LDS reads hide it).  Wait states by hand behind such chains; nothing may be scheduled across them.
__device__ __forceinline__ void mfma_result_guard() {
#ifdef PNMOL_NO_MFMA_GUARD  // (tools/mfma_hazard_scan.py must then find the reads in the ISA)
    return;
#endif
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// ---- bulk operand ring of k_sweep_rl ------------------------------------------------------------------------------
// The rank-32 update of a row block reads one 32x32 tile of L per tile it updates; as fragment loads to registers with one
// tile of look-ahead (all the register file allows) a tile took 0.45 us against 0.21 us of MFMA: one miss latency each.
// Each wave now streams ITS half tiles (16 rows x 256 B) through a private LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: no registers, RING tiles ahead, no workgroup barrier), and reads its B fragments with
// ds_read_b128.  An LDS-DMA instruction writes 64 x 16 B contiguously, so the image is row-major without padding and the
// bank conflicts of 16 lanes reading 16 rows at one column are avoided by XOR-swizzling the 16-byte column index with the
// row on the SOURCE address: slot (row, g ^ row) holds columns 2g, 2g+1 of the row.
constexpr int RING = 3;
constexpr int RING_LINV = RING, RING_NEWEST = RING + 1;  // two more slots per wave: prefetched L_jj^-1 and tile (j+1, j)
// one 16-byte LDS-DMA per lane: LDS[lds_dst + 16 lane] = *gsrc   (M0 is written and restored in the same statement)
__device__ __forceinline__ void lds_dma16(const double* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// half tile (rows wc*16 .. +15 of the tile at `tile`, leading dimension ld) -> ring slot at LDS byte address `slot`
__device__ __forceinline__ void ring_fill(const double* tile, const int (&off)[4], unsigned slot) {
#pragma unroll
    for (int c = 0; c < 4; ++c) lds_dma16(tile + off[c], slot + 1024u * c);
}
// at most `tiles_behind` later half tiles (4 DMAs each) may still be in flight
__device__ __forceinline__ void ring_wait(int tiles_behind) {
    switch (tiles_behind) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}
// B fragment of lane (fr, fk) from a ring slot: row fr, columns 8 fk .. 8 fk + 7
__device__ __forceinline__ void ring_frag(Frag8& f, const double* slot, int fr, int fk) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double2 t = *reinterpret_cast<const double2*>(slot + (fr * 16 + ((4 * fk + e) ^ fr)) * 2);
        f.v[2 * e] = t.x;
        f.v[2 * e + 1] = t.y;
    }
}

// The chain workgroup (block 0 of k_sweep_rl): factorises ALL diagonal blocks one after the other, so no hand-over between
// workgroups sits on the critical path.  Block J needs  D_J = D'_J - X X^T,  X = X_{J,J-1} = S_{J,J-1} L_{J-1,J-1}^-T:
// row block J feeds S_{J,J-1} and -D'_J (everything that does not depend on block J-1's factor: ready about one
// factorisation earlier) through `feed`; wave 0, idle during the second half of factorisation J-1, brings them into LDS;
// L_{J-1,J-1}^-1 goes from wave 1's registers through LDS; X is published as tile (J, J-1) of F by wave 2, which also
// writes the previous L_JJ and checks its pivots while the others go on.
//
// XL ("XCD-local"): the launch has placed the chain workgroup and the row blocks of S -- everybody on the critical loop
// L^-1, X -> row J+1 -> feed -> next factorisation -- on blockIdx = 0 mod 8, i.e. on ONE XCD (round-robin dispatch; each row
// block checks it against the XCC id published here and falls back to the write-through flags if not).  A hand-over inside
// an XCD needs no write-through: plain stores stop in the shared L2, where the readers' L1-bypassing loads find them
// (tools/hop_load_bench.hip: 0.7-1.1 us per hop against 1.2-1.6 us, and the write-through latency grows with the fabric
// traffic of the down-date workgroups).  L^-1 and X are therefore stored twice: into the L2 first (flags fdiagL / fnewL),
// written through afterwards for the readers on the other XCDs (W rows, identity rows, down-date: fdiag / fnew as before).
template <bool XL>
__device__ __forceinline__ void sweep_chain_role(ChainLds& C, const double* __restrict__ G, double* F, double* Linv, int ld,
                                                 int CB, int* flags, const RlFlags& fl, int* info,
                                                 double* feed, int lenient, int tid, int l, int w) {
    int* fdiag_all = flags + fl.fdiag;
    int* fnew_all = flags + fl.fnew;
    const int CBp = fl.CBp;
    int* ffeed = flags + fl.ffeed;
    int* fabort = flags + fl.fabort;
    const int wr = w >> 1, wc = w & 1, fr = l & 15, fk = l >> 4;
    const int offC = (wr * 16 + fk) * TLD + wc * 16 + fr;
    const __amdgpu_buffer_rsrc_t rF = wt_rsrc(F), rlinv = wt_rsrc(Linv), rfeed = wt_rsrc(feed), rflags = wt_rsrc(flags);
    if (XL && tid == 0) flag_st(flags + fl.fxcc, 1 + xcc_id());
    WaveWait ww{false};
    double smax = 0.0;
    for (int e = tid; e < ld; e += 256) smax = fmax(smax, fabs(G[(long)e * ld + e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o));
    if (l == 0) C.red[w] = smax;
    if (tid < NB) C.sd[0][tid] = G[(long)tid * ld + tid];
    if (tid < 16) C.d4[0].flagA[tid] = 0, C.d4[1].flagA[tid] = 0;
    for (int e = tid; e < 16 * 16; e += 256) C.sLinv[(e >> 4) * TLD + 16 + (e & 15)] = 0.0;  // upper-right quadrant of L^-1
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    smax = lenient ? __builtin_inf() : fmax(fmax(C.red[0], C.red[1]), fmax(C.red[2], C.red[3]));
    const double pivtol = pivot_tolerance(lenient);
    d4 accD;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // quadrant of -G_00, lower entries only
        const int rl = wr * 16 + fk + 4 * r, cl = wc * 16 + fr;
        accD[r] = -G[(long)(rl >= cl ? rl : cl) * ld + (rl >= cl ? cl : rl)];
    }
    for (int J = 0; J < CB; ++J) {
        Diag4Lds* dl = &C.d4[J & 1];
        CHAIN_TRACE(J, 0);
        CHAIN_TRACE_CLK(J, 4);
        if (w == 2 && J > 0)  // L_{J-1,J-1} and its pivots, off the critical path
            diag4_output(&C.d4[(J - 1) & 1], l, F + ((long)(J - 1) * NB) * ld + (long)(J - 1) * NB, ld, info, (J - 1) * NB,
                         C.sd[(J - 1) & 1], smax, pivtol);
        diag4_factor<true, true, false, false>(accD, dl, w, l, F + ((long)J * NB) * ld + (long)J * NB, ld,
                                               Linv + (long)J * NB * NB, info, J * NB, C.sd[J & 1], smax, C.sLinv, pivtol);
        CHAIN_TRACE_W(J, 1, 3);
        CHAIN_TRACE_W(J, 6, 1);
        // XL: wave 1, which has just finished L_JJ^-1, sends it to the row blocks of S on this XCD NOW -- before this
        // workgroup waits for the feed of row block J+1.  (Round 2 published it behind that wait: row block J+2's
        // TRSM with L_JJ^-1, hence its feed, hence the next wait, started only when the feed of row block J+1 had
        // arrived -- a cycle  feed(J+1) -> L_JJ^-1 out -> TRSM, feed(J+2)  of 6.2 us around a chain that needs 4.5.)
        if (XL && w == 1 && J + 1 < CB) {
            TileRegs t1;
            tile_regs_from_lds(t1, C.sLinv, l);  // (this wave's own LDS writes + the zero quadrant set at kernel start)
            tile_regs_store<0>(t1, rlinv, (unsigned)(J * NB * NB * 8), NB, l);
            drain_vmem();
            if (l == 0) l2_flag_st(rflags, fl.fdiagL + J, 1);
        }
        // w1 has left L^-1 in C.sLinv.  w2 publishes it behind the next block's TRSM: the other row blocks need it, the
        // next diagonal block does not.
        if (J + 1 == CB) {
            __syncthreads();
            if (w == 2) {
                wt_rows_from_lds<8>(rlinv, (unsigned)(J * NB * NB * 8), NB, C.sLinv, 0, l);
                drain_vmem();
                if (l < RL_REP) flag_st(fdiag_all + l * CBp + J, 1);
                if (XL && l == 0) l2_flag_st(rflags, fl.fdiagL + J, 1);
            }
            break;
        }
        if (w == 0 || w == 2) {  // these two are free early: fetch what row block J+1 has fed (w0: S, w2: -D')
            wave_wait_ge(ffeed + J + 1, 1, fabort, ww);
            sc1_tile_to_lds((w == 2) ? C.sFeedD : C.sFeedS, rfeed, (unsigned)((2 * (J + 1) + (w == 2 ? 1 : 0)) * NB * NB * 8), l);
            // (w2, behind its use of sd[(J-1) & 1] in diag4_output above)
            if (w == 2 && l < NB) C.sd[(J + 1) & 1][l] = G[(long)((J + 1) * NB + l) * ld + (J + 1) * NB + l];
        }
        __syncthreads();  // sLinv, sFeedS, sFeedD, sd complete; everybody is done with d4[(J+1) & 1]'s previous use
        CHAIN_TRACE(J, 2);
        if (tid < 16) C.d4[(J + 1) & 1].flagA[tid] = 0;
        // w2 sends L_JJ^-1 on its way NOW (everybody else's TRSM of step J waits for it); the stores drain behind this
        // block's TRSM
        TileRegs tl;
        if (w == 2) {
            if constexpr (XL) {
                tile_regs_from_lds(tl, C.sLinv, l);  // (for the write-through copy below; wave 1 has served this XCD)
            } else {
                wt_rows_from_lds<8>(rlinv, (unsigned)(J * NB * NB * 8), NB, C.sLinv, 0, l);
            }
        }
        // X = S L^-T  (A: rows of S from LDS, B: rows of L^-1 from LDS)
        d4 x = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; ++s)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(C.sFeedS[(wr * 16 + fr) * TLD + 8 * fk + s],
                                                     C.sLinv[(wc * 16 + fr) * TLD + 8 * fk + s], x, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) C.sX[offC + 4 * r * TLD] = x[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) accD[r] = C.sFeedD[offC + 4 * r * TLD];
        __syncthreads();  // X complete, flags of the next factorisation zeroed
        CHAIN_TRACE(J, 3);
        if (w == 2) {
            if constexpr (!XL) {
                drain_vmem();
                if (l < RL_REP) flag_st(fdiag_all + l * CBp + J, 1);  // all copies, one store instruction
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
            accD = __builtin_amdgcn_mfma_f64_16x16x4f64(C.sX[(wr * 16 + fr) * TLD + 8 * fk + s],
                                                        C.sX[(wc * 16 + fr) * TLD + 8 * fk + s], accD, 0, 0, 0);
        if (w == 2) {  // tile (J+1, J) of F: the other row blocks' "newest" operand of step J+1
            const unsigned orgX = (unsigned)((((long)(J + 1) * NB) * ld + (long)J * NB) * 8);
            if constexpr (XL) {
                TileRegs tx;
                tile_regs_from_lds(tx, C.sX, l);
                tile_regs_store<0>(tx, rF, orgX, ld, l);
                drain_vmem();
                if (l == 0) l2_flag_st(rflags, fl.fnewL + J, 1);
                // ... and now both tiles once more, written through, for the row blocks on the other XCDs
                tile_regs_store<16>(tl, rlinv, (unsigned)(J * NB * NB * 8), NB, l);
                tile_regs_store<16>(tx, rF, orgX, ld, l);
                drain_vmem();
                if (l < RL_REP) flag_st(fdiag_all + l * CBp + J, 1);
                if (l < RL_REP) flag_st(fnew_all + l * CBp + J, 1);
            } else {
                wt_rows_from_lds<8>(rF, orgX, ld, C.sX, 0, l);
                drain_vmem();
                if (l < RL_REP) flag_st(fnew_all + l * CBp + J, 1);
            }
        }
    }
    if (w == 2)
        diag4_output(&C.d4[(CB - 1) & 1], l, F + ((long)(CB - 1) * NB) * ld + (long)(CB - 1) * NB, ld, info, (CB - 1) * NB,
                     C.sd[(CB - 1) & 1], smax, pivtol);
    if (l == 0 && ww.dead) atomicMin(info, -2);
}

template <int N, bool FUSED, int MAXT, bool XL>
__global__ __launch_bounds__(256) void k_sweep_rl(const double* __restrict__ G, double* F, double* Linv, int ld, int CB,
                                                  int RT, int* flags, int* info_base, const int* __restrict__ ctr,
                                                  DowndateArgs dd, double* feed, int lenient, int home) {
    __shared__ __attribute__((aligned(16))) union {
        SweepLds L;
        ChainLds C;
    } lds;
    SweepLds& L = lds.L;
    __shared__ __attribute__((aligned(16))) double bulk_ring[4][RING + 2][16 * NB];  // per wave: half tiles (4 KB each)
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, fr = l & 15, fk = l >> 4;
    const RlFlags fl = rl_flags(RT, CB);
    int* frow = flags + fl.frow;
    int* fabort = flags + fl.fabort;
    int* ffeed = flags + fl.ffeed;
    const int* fdiag = flags + fl.fdiag + (blockIdx.x % RL_REP) * fl.CBp;  // the copies this workgroup polls
    const int* fnew = flags + fl.fnew + (blockIdx.x % RL_REP) * fl.CBp;
    const int* fcol = flags + fl.fcol + (blockIdx.x % RL_REP) * fl.CBp;
    int* info = info_base + (*ctr - 1);
    // XL: the launch starts `home` (0 .. 7) empty blocks early, so that block 0 of the layout below -- the chain workgroup --
    // and the row blocks of S land on XCD `home`: sweeps of different filters that share the device get different XCDs
    // (with all of them on XCD 0, eight problems in flight fought over its 32 CUs: 7.0 k steps/s against 8.7 k)
    const int bx = XL ? (int)blockIdx.x - home : (int)blockIdx.x;
    if (bx < 0) return;
    if (bx == 0) {
        sweep_chain_role<XL>(lds.C, G, F, Linv, ld, CB, flags, fl, info, feed, lenient, tid, l, w);
        return;
    }
    // !XL: block 1 + I is row block I.  XL: blocks 8 s, s = 1 .. CB-1, are the row blocks of S (I = s: the chain workgroup's
    // XCD) and the blocks between them are empty -- the row blocks of W / Ls^-T and the down-date pairs wait for these
    // (lower) blocks only, and an XCD dispatches in index order.  (The chain workgroup itself waits for the feed of row
    // block J+1, a HIGHER index on the same XCD: see "Forward progress" at SWEEP_SPIN_LIMIT.)  The remaining row blocks and
    // the down-date pairs follow behind block 8 (CB-1).
    int I, part = 0;  // part: 0 whole row block, 1 / 2 = parts A / B of a split row block of S (S_SPLIT_FROM)
    if constexpr (XL) {
        const int nsplit = s_split_count(CB, true);
        const int b = bx, nslot = 8 * (CB - 1 + nsplit) + 1;
        if (b < nslot) {
            if (b % 8 != 0) return;
            const int sidx = b / 8;
            if (sidx < CB) {
                I = sidx;
                part = (nsplit > 0 && I >= S_SPLIT_FROM) ? 1 : 0;
            } else {  // (the parts B follow the rows of S on the same XCD: they wait for their parts A, lower block indices)
                I = S_SPLIT_FROM + (sidx - CB);
                part = 2;
            }
        } else {
            // ... on the seven other XCDs: the chain workgroup's XCD (32 CUs) keeps its CUs for the CB workgroups above --
            // with every eighth of the remaining workgroups on it as well, some of them found no CU for most of the sweep
            if (b % 8 == 0) return;
            const int q = b - nslot;  // (nslot = 1 mod 8: block nslot + q is empty when (q + 1) % 8 == 0)
            I = CB + q - (q + 1) / 8;
        }
    } else {
        I = bx - 1;
    }
#ifdef PNMOL_SWEEP_STAMP
    if (tid == 0) L.stamp_id = part == 1 ? 200 + I : 1 + I;  // (visible to everybody behind the workgroup's first barrier; only thread 0 and
                                       //  lane-0 threads behind later barriers stamp)
#endif
    {
        const int zb = RT - CB - 1;  // the r^T block is dealt right behind the rows of S (see k_sweep)
        if (I >= CB && I <= zb) I = (I == CB) ? zb : I - 1;
    }
    const int nwsplit = w_split_count(RT, CB);
    if (I >= RT && I < RT + nwsplit) {  // part B of row block CB + (I - RT) of W
        I = CB + (I - RT);
        part = 2;
    } else if (I >= RT) {
        I -= nwsplit;                   // (the down-date pairs: logical indices RT ..)
        if constexpr (FUSED) {
            if (tid == 0) {
                L.dead = 0, L.seen[0] = 0;  // (seen[0]: column blocks of W wave 0 has seen complete, see the down-date loop)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            SWEEP_STAMP(0);
            sweep_downdate_role<N, true>(L, dd, F, ld, CB, CB, frow, fabort, info, I - RT, tid, l, w, *ctr - 1);
        }
        return;
    } else if (nwsplit > 0 && I >= CB && I < RT - CB - 1) {
        part = 1;                       // part A of a row block of W
    }
    const bool chain = I < CB;
    if (chain && I == 0) return;  // (block (0,0) is the chain workgroup's own)
    if (tid == 0) {
        L.dead = 0, L.pub = 0, L.pubcnt = part == 2 ? 4 * (I < CB ? S_SPLIT_H : W_SPLIT_H) : 0, L.rdiag = 0, L.rnew = 0, L.rcol = 0, L.rzb = 0;
        L.seen[0] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one invalidate per workgroup (see k_sweep)
    }
    __syncthreads();
    SWEEP_STAMP(0);

    const long rowC = (long)I * NB + wr * 16 + fk;
    const int colC = wc * 16 + fr;
    const int offC = (wr * 16 + fk) * TLD + colC;
    // Steps 0 .. last end with X_j; a chain row runs one more, incomplete step (j = I - 1: tile I-1 with all panels but
    // the last, then fed to the chain workgroup with -D') and holds tiles 0 .. I-1.
    const bool feeds = chain && part != 1;           // (part A of a split row block stops at tile S_SPLIT_H - 1 and feeds nothing)
    const int hsplit = chain ? S_SPLIT_H : W_SPLIT_H;
    const int t0 = part == 2 ? hsplit : 0;           // first tile this workgroup owns
    const int last = part == 1 ? hsplit - 1 : (chain ? I - 2 : CB - 1);
    const int ntiles = part == 1 ? hsplit : (chain ? I : CB);   // (one past its last tile)
    d4 n[MAXT];  // slot u: quadrant of -(tile j + u) at step j >= t0 (part B before its first step: tile t0 + u)
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
        if (t0 + t < ntiles) {
#pragma unroll
            for (int r = 0; r < 4; ++r) n[t][r] = -G[(rowC + 4 * r) * ld + (long)(t0 + t) * NB + colC];
        }
    d4 accD = {0, 0, 0, 0};  // quadrant of -D' (see k_sweep)
    if (feeds) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rl = wr * 16 + fk + 4 * r, cl = colC;
            accD[r] = -G[((long)I * NB + (rl >= cl ? rl : cl)) * ld + (long)I * NB + (rl >= cl ? cl : rl)];
        }
    }
    WaveWait ww{false};
    int known_diag = 0, known_new = 0, known_col = 0;  // wave 0: progress seen so far (relay_progress)
    const __amdgpu_buffer_rsrc_t rF = wt_rsrc(F), rfeed = wt_rsrc(feed), rflags = wt_rsrc(flags);
    // this wave's bulk operand ring and its lanes' source offsets (element units) inside a tile, per DMA instruction
    const double* ringw = &bulk_ring[w][0][0];
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)ringw);
    int roff[4], roffL[4];  // (roffL: the same inside a tile of Linv, leading dimension NB)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int row = 4 * c + (l >> 4);
        roff[c] = (wc * 16 + row) * ld + 2 * ((l & 15) ^ row);
        roffL[c] = (wc * 16 + row) * NB + 2 * ((l & 15) ^ row);
    }
    // A row block that lags behind the chain workgroup (all rows of W and of the identity block, the late rows of S) finds
    // L_jj^-1 and the newest tile (j+1, j) published long before it needs them: each wave then fetches its half of them by
    // LDS-DMA a phase early (L_jj^-1 before the bulk update of step j, the newest tile at the end of step j), instead of
    // loading them when the step gets there (0.3-0.5 us of exposed latency each, 17 times).
    // tiles of a row of W / Ls^-T that are flagged right behind their stores instead of at the next step's first load: where
    // the row block has slack (little left to update), so that the down-date workgroups -- MFMA-bound, 5.5 us per column
    // block -- get the last blocks of W a chain period earlier and finish closer behind the sweep
#ifndef PNMOL_EAGER_TILES
#define PNMOL_EAGER_TILES 6
#endif
    constexpr int EAGER_TILES = PNMOL_EAGER_TILES;
    bool pre_b0 = false;
    // identity rows (rows of Ls^-T): the error-model product (Sq^-1 z)_k of this wave's eight rows, now (vecops_rows8)
    // (parked in L.sd[8 w + q], which the row blocks of this kernel do not use otherwise: no registers across the sweep)
    if constexpr (FUSED) {
        const int RBW = RT - 2 * CB - 1, zb = CB + RBW;
        if (!chain && I > zb) {
            double bq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (dd.va.Sqinv != nullptr) rows8_dot(dd.va.Sqinv, (long)(I - zb - 1) * NB + 8 * w, ld, dd.va.z, bq, l);
            if (l == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) L.sd[8 * w + q] = bq[q];
            }
        }
    }
    // XL: a row block of S that does share the chain workgroup's XCD takes L^-1 and the newest tile from the L2 (flags
    // fdiagL / fnewL) and feeds through it
    bool local = false;
    if constexpr (XL) {
        if (chain) {
            if (tid == 0) {
                int v = 0;
                for (int spins = 0; spins < 4096 && (v = flag_ld(flags + fl.fxcc)) == 0; ++spins) __builtin_amdgcn_s_sleep(1);
                L.seen[0] = (v == 1 + xcc_id());
            }
            __syncthreads();
            local = L.seen[0] != 0;
            if (local) fdiag = flags + fl.fdiagL, fnew = flags + fl.fnewL;
        }
    }
    Frag8 ax;  // rows wr*16 + fr of X_{j-1}, columns 8 fk .. 8 fk + 7 (A operand of both updates of step j)
    // Tile t of this row block is published once the stores of all four waves have completed; the last wave to say so
    // sets the flag(s).  Called right behind the first wait for a LOAD issued after those stores (memory operations of a
    // wave complete in issue order: no waiting here) -- a blocking drain at the end of step t cost the rows with little
    // to update, the ones that feed the chain workgroup, 0.8-1.5 us per step (write-through latency).
    auto flag_tile = [&](int t) {
        drain_vmem();
        int lastw = 0;
        if (l == 0) lastw = __hip_atomic_fetch_add(&L.pubcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 4 * t + 3;
        if (__builtin_amdgcn_readfirstlane(lastw)) {
            if (l == 0) flag_st(frow + I, t + 1);
            if (XL && chain && l == 0) {  // (a row block that is not on the chain workgroup's XCD after all: written through)
                if (local) l2_flag_st(rflags, fl.frowL + I, t + 1);
                else flag_st(flags + fl.frowL + I, t + 1);
            }
            if (chain && l < RL_REP)
                __hip_atomic_fetch_add(flags + fl.fcol + l * fl.CBp + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (part == 2) {
        // Part B of a split row block.  Columns k = 0 .. t0-1: X_{I,k} comes from part A (tile (I, k) of F), panel k goes onto
        // ALL own tiles t0 .. ntiles-1 (the slots stay put) and, for a row block of S, onto D': the first step below (j = t0)
        // finds its tiles as if this workgroup had run the steps before it.
        const int* fsrc = (XL && local) ? flags + fl.frowL : frow;
        for (int k = 0; k < t0; ++k) {
            if (chain) {
                if (k + 1 < t0) {
                    relay_progress_min(fsrc + t0, I - t0 + 1, k + 1, known_col, &L.rcol, fabort, ww, w, l);  // rows t0 .. I: tile k out
                } else {  // (k = t0-1: tile (t0, t0-1) is the chain workgroup's; rows t0+1 .. I, part A among them, have the rest)
                    relay_progress(fnew, CB - 1, [](int) { return 1; }, t0, known_new, &L.rnew, fabort, ww, w, l);
                    relay_progress_min(fsrc + t0 + 1, I - t0, t0, known_col, &L.rcol, fabort, ww, w, l);
                }
            } else {  // (a row block of W: all rows of S have published tile k; part A has published tile (I, k))
                relay_progress(fnew, CB - 1, [](int) { return 1; }, k + 1, known_new, &L.rnew, fabort, ww, w, l);
                relay_progress(fcol, CB - 1, [CB](int q) { return CB - 2 - q; }, k + 1, known_col, &L.rcol, fabort, ww, w, l);
                relay_wait_ge(frow + I, k + 1, &L.rzb, k + 1, fabort, ww, w);
            }
            Frag8 bx;
            frag_ld(ax, F + ((long)I * NB + wr * 16 + fr) * ld + (long)k * NB + 8 * fk);
            frag_ld(bx, F + ((long)I * NB + wc * 16 + fr) * ld + (long)k * NB + 8 * fk);
#pragma unroll
            for (int u0 = 0; u0 < 8; u0 += 4) {
                Frag8 b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t0 + u0 + q < ntiles)
                        frag_ld(b[q], F + ((long)(t0 + u0 + q) * NB + wc * 16 + fr) * ld + (long)k * NB + 8 * fk);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t0 + u0 + q < ntiles) {
#pragma unroll
                        for (int s = 0; s < 8; ++s)
                            n[u0 + q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ax.v[s], b[q].v[s], n[u0 + q], 0, 0, 0);
                    }
            }
            if (chain) {
#pragma unroll
                for (int s = 0; s < 8; ++s) accD = __builtin_amdgcn_mfma_f64_16x16x4f64(ax.v[s], bx.v[s], accD, 0, 0, 0);
            }
            mfma_result_guard();
        }
    }
    // EAGER right-looking steps (all row blocks).  When X_j exists, the whole column j of L -- tile (j+1, j) from the chain
    // workgroup, tiles (t, j) of the later rows of S -- is out or about to be: panel j goes onto ALL remaining tiles at once,
    // tile j+1 included, and the next step starts with a finished S_{j+1}.  Round 2's step (panel j-1 onto tile j "newest
    // panel", panel j-1 onto the rest, then X_j) had one more poll, one more dependent tile load and one more 8-MFMA product
    // with its LDS round trip and barrier per step: ~4.2 us beside the tile updates for a row of W, 5.5-6.8 us for a row of S
    // whatever it had left to update -- and the rows of S, each one step of that length per chain block, set the period of
    // the chain workgroup (6.6 us against the 4.5 it needs; profiles/r03_microbench/sweep_timeline_e_s_row7_trace.log).
    // Same operations on the same operands in the same order per tile: the results are bit-identical to round 2's.
    // A row block of S that feeds ends with its feed INSIDE its last step: panel I-2 onto tile I-1 is that step's bulk.
    const bool lastlocal = feeds && XL && local;
#pragma unroll
    for (int r = 0; r < 4; ++r) L.sS[t0 & 1][offC + 4 * r * TLD] = -n[0][r];
    __syncthreads();
    bool pre_l = false;
    for (int j = t0; j <= last; ++j) {
        const double* sS = L.sS[j & 1];
        SWEEP_TRACE(j, 0);
        relay_progress(fdiag, CB, [](int) { return 1; }, j + 1, known_diag, &L.rdiag, fabort, ww, w, l);
        SWEEP_TRACE(j, 5);
        Frag8 bl;
        if (pre_l) {
            ring_wait(0);
            ring_frag(bl, ringw + 512 * RING_LINV, fr, fk);
        } else {
            frag_ld(bl, Linv + (long)j * NB * NB + (wc * 16 + fr) * NB + 8 * fk);
        }
        d4 x = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; ++s)
            x = __builtin_amdgcn_mfma_f64_16x16x4f64(sS[(wr * 16 + fr) * TLD + 8 * fk + s], bl.v[s], x, 0, 0, 0);
        double* sXj = (j & 1) ? L.sP[0] : L.sX;
#pragma unroll
        for (int r = 0; r < 4; ++r) sXj[offC + 4 * r * TLD] = x[r];
        __syncthreads();  // X_j complete
        SWEEP_TRACE(j, 3);
        // every wave publishes a quarter of the tile (two full 1-KB store instructions).  The last ordinary tile of a feeding
        // row block on the chain workgroup's XCD goes into that XCD's L2 only (the next row block of S needs it for ITS last
        // step); its written-through copy follows behind the feed
        const bool llast = lastlocal && j == last;
        const unsigned orgX = (unsigned)((((long)I * NB) * ld + (long)j * NB) * 8);
        if (llast) wt_rows_from_lds<2, 0>(rF, orgX, ld, sXj, 8 * w, l);
        else wt_rows_from_lds<2>(rF, orgX, ld, sXj, 8 * w, l);
        if (feeds) {  // D' -= X_j X_j^T
#pragma unroll
            for (int s = 0; s < 8; ++s)
                accD = __builtin_amdgcn_mfma_f64_16x16x4f64(sXj[(wr * 16 + fr) * TLD + 8 * fk + s],
                                                            sXj[(wc * 16 + fr) * TLD + 8 * fk + s], accD, 0, 0, 0);
            mfma_result_guard();
        }
        const int cnt = ntiles - 1 - j;  // tiles j+1 .. ntiles-1
        pre_l = false;
        if (cnt > 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) ax.v[s] = sXj[(wr * 16 + fr) * TLD + 8 * fk + s];
            // column j of L: tile (j+1, j) from the chain workgroup ...
            relay_progress(fnew, CB - 1, [](int) { return 1; }, j + 1, known_new, &L.rnew, fabort, ww, w, l);
            // X_j is flagged HERE, behind one poll round trip (its stores are ~0.5 us old: a short wait), not behind the wait
            // for the other rows of S
            if (llast) {
                drain_vmem();
                int lastw = 0;
                if (l == 0) lastw = __hip_atomic_fetch_add(&L.pub, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 3;
                if (__builtin_amdgcn_readfirstlane(lastw) && l == 0) l2_flag_st(rflags, fl.frowL + I, j + 1);
            } else if (chain) {
                flag_tile(j);
            }
            // ... tiles (t, j), t >= j+2, from the rows of S: a row block of S reads those of the rows j+2 .. ntiles-1 only and
            // waits for exactly them, the others for the column's counter
            if (chain) {
                if (cnt > 1)
                    relay_progress_min((XL && local ? flags + fl.frowL : frow) + j + 2, cnt - 1, j + 1, known_col, &L.rcol, fabort, ww, w, l);
            } else {
                relay_progress(fcol, CB - 1, [CB](int k) { return CB - 2 - k; }, j + 1, known_col, &L.rcol, fabort, ww, w, l);
            }
            SWEEP_TRACE(j, 1);
            const double* Lt = F + (long)j * NB + (long)(j + 1) * NB * ld;  // tile (j+1, j); + k * NB * ld
#pragma unroll
            for (int r = 0; r < RING; ++r)
                if (r < cnt) ring_fill(Lt + (long)r * NB * ld, roff, ring0 + 4096u * r);
            if (!chain) {  // L_{j+1,j+1}^-1 for the next step, if it is out already (the rows that lag behind the chain)
                const int seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&L.rdiag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (seen >= j + 2 && seen != (1 << 30)) {
                    ring_fill(Linv + (long)(j + 1) * NB * NB, roffL, ring0 + 4096u * RING_LINV);
                    pre_l = true;
                }
            }
            // Everything issued so far has to land before the counted waits of the loop below are meaningful (they count
            // this panel's LDS-DMA loads only): one full wait.  A row block of W / Ls^-T flags X_j on THIS wait -- the
            // write-through drain of its stores (~0.5 us) and the latency of the first operand tiles overlap instead of
            // following each other, 17 times per launch in the workgroups the sweep waits for (the column's tiles are
            // nearly always out when these rows get here, so the flag is not held up by the wait above).
            if (!chain) flag_tile(j);
            else ring_wait(0);
            Frag8 b[2];
            ring_frag(b[0], ringw, fr, fk);
#pragma unroll
            for (int u = 1; u < MAXT; ++u) {
                if (u <= cnt) {
                    const int k = u - 1;
                    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): b[k & 1] is here, slot k % RING is free again
                    asm volatile("" ::: "memory");
                    if (k + RING < cnt) ring_fill(Lt + (long)(k + RING) * NB * ld, roff, ring0 + 4096u * (k % RING));
                    d4 a = __builtin_amdgcn_mfma_f64_16x16x4f64(ax.v[0], b[k & 1].v[0], n[u], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 1 < cnt) {
                        // tiles issued after tile k+1: those still in the ring window behind it
                        const int behind = (k + RING < cnt ? k + RING : cnt - 1) - (k + 1);
                        ring_wait(behind < RING - 1 ? behind : RING - 1);
                        ring_frag(b[(k + 1) & 1], ringw + 512 * ((k + 1) % RING), fr, fk);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 1; s < 8; ++s)
                        a = __builtin_amdgcn_mfma_f64_16x16x4f64(ax.v[s], b[k & 1].v[s], a, 0, 0, 0);
                    n[u - 1] = a;
                } else {
                    // (the exit edge of the unrolled loop: the wait states behind the last tile's chain sit HERE -- behind
                    //  the loop they came behind the accumulator moves of the join, whose first read followed the chain's
                    //  last MFMA by six wait states: tools/mfma_hazard_scan.py)
                    mfma_result_guard();
                    break;
                }
            }
            mfma_result_guard();
            SWEEP_TRACE(j, 4);
            if (!(feeds && j == last)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.sS[(j + 1) & 1][offC + 4 * r * TLD] = -n[0][r];
                __syncthreads();  // S_{j+1} complete
            }
        }
        SWEEP_TRACE(j, 6);
    }
    if (feeds) {
        // slot 0 is tile I-1 with every panel but the last: it goes to the chain workgroup together with -D'
        const int j = I - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            L.sS[0][offC + 4 * r * TLD] = -n[0][r];
            L.sS[1][offC + 4 * r * TLD] = accD[r];
        }
        __syncthreads();
        if (XL && local) {
            wt_rows_from_lds<2, 0>(rfeed, (unsigned)((2 * I) * NB * NB * 8), NB, L.sS[0], 8 * w, l);
            wt_rows_from_lds<2, 0>(rfeed, (unsigned)((2 * I + 1) * NB * NB * 8), NB, L.sS[1], 8 * w, l);
        } else {
            wt_rows_from_lds<2>(rfeed, (unsigned)((2 * I) * NB * NB * 8), NB, L.sS[0], 8 * w, l);
            wt_rows_from_lds<2>(rfeed, (unsigned)((2 * I + 1) * NB * NB * 8), NB, L.sS[1], 8 * w, l);
        }
        SWEEP_TRACE(j, 2);
        drain_vmem();
        SWEEP_TRACE(j, 3);
        __syncthreads();
        if (tid == 0) {
            if (XL && local) l2_flag_st(rflags, fl.ffeed + I, 1);
            else flag_st(ffeed + I, 1);
        }
        SWEEP_TRACE(j, 4);
        SWEEP_STAMP(1);
        // (on the chain workgroup's XCD) the written-through copy of tile (I, I-2) for everybody else, and its flags
        if (last >= 0 && lastlocal) {
            const double* sXl = (last & 1) ? L.sP[0] : L.sX;
            wt_rows_from_lds<2>(rF, (unsigned)((((long)I * NB) * ld + (long)last * NB) * 8), ld, sXl, 8 * w, l);
            flag_tile(last);
        }
    }
    if (!feeds) flag_tile(ntiles - 1);  // (a feeding row block of S has flagged its last own tile in or behind its feed)
    if (!chain) {
        if constexpr (FUSED) {
            const int RBW = RT - 2 * CB - 1, zb = CB + RBW;
            if (I != zb && part != 1) {  // the vector ops of the step for the rows this workgroup has just finished (see k_sweep)
                __syncthreads();  // every wave has drained this row block's last tile
                relay_wait_ge(frow + zb, CB, &L.rzb, 1 << 24, fabort, ww, w);  // (tag above those of a part B's prologue)
                const long Dp = (long)RBW * NB;
                const long row0 = (I < zb ? (long)(I - CB) * NB : Dp + (long)(I - zb - 1) * NB) + 8 * w;
                const double* W = F + (long)ld * ld;
                SWEEP_STAMP(2);
                vecops_rows8(dd.va, W, ld, Dp, row0, L.sd + 8 * w, l);
            }
        }
    }
    if (l == 0 && ww.dead) atomicMin(info, -2);
    SWEEP_STAMP(5);
}

// ------------------------------------------------------------------------------------------
// covariance down-date  P = P- - W W^T  on 16x16 point tiles x all n*n derivative blocks.
// One workgroup (4 waves) per lower tile pair (J >= K).  The waves split the inner (measurement) dimension
// in interleaved 8-column chunks; each wave streams its chunks global -> registers -> wave-private LDS ->
// MFMA fragments with no workgroup barrier inside the loop; latency is hidden by occupancy (4 waves/SIMD).
// Partial sums are combined through LDS in two rounds; waves 0/1 then write the tile and, transposed through
// LDS so that it leaves as full 128-byte rows, its mirror image.  W = F + mp*ld (Dp x mp).
// This launch subtracts the columns [8*c_begin, 8*c_end) of W; Ppred may alias Pout (a workgroup only reads
// the tile it writes), which is how later column groups accumulate in place.
// ------------------------------------------------------------------------------------------
template <int N>
struct DowndateCfg {
    static constexpr int CW = 8;                    // chunk width (columns of W)
    static constexpr int SLD = CW + 1;              // staging leading dim
    static constexpr int STG = 2 * N * 16 * SLD;    // doubles of staging per wave
    static constexpr int RED = N * N * 4 * 64;      // doubles of partial sums per wave
    static constexpr int TLD16 = 17;                // transpose tile leading dim
    static constexpr int LDS_D = (4 * STG > 2 * RED) ? 4 * STG : 2 * RED;
};

// second reduction round + epilogue of k_downdate for wave WV in {0,1}
template <int N, int WV>
__device__ __forceinline__ void downdate_finish(d4 (&acc)[N][N], double* smem, const double* Ppred, double* Pout,
                                                double* __restrict__ var, int dp, long Dp, int J, int K, int l) {
    using C = DowndateCfg<N>;
    const int fr = l & 15, fk = l >> 4;
    double* wr = smem + (1 - WV) * C::RED;  // my contribution to the OTHER wave's registers
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) wr[((a * N + b) * 2 + q) * 64 + l] = acc[a][b][(1 - WV) * 2 + q];
    __syncthreads();
    const double* rdo = smem + WV * C::RED;
    double x[N][N][2];
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int row = fk + 4 * (WV * 2 + q);
                const long gi = ((long)a * dp + J * 16 + row) * Dp + (long)b * dp + K * 16 + fr;
                x[a][b][q] = Ppred[gi] - (acc[a][b][WV * 2 + q] + rdo[((a * N + b) * 2 + q) * 64 + l]);
                Pout[gi] = x[a][b][q];
                if (J == K && a == b && row == fr) var[a * dp + J * 16 + row] = x[a][b][q];
            }
    if (J == K) return;  // a diagonal tile pair is its own mirror image (workgroup-uniform)
    // mirror image: block (b,a) of tile (K,J) = transpose.  This wave holds rows {fk + 4(2WV+q)} of each block;
    // stage them in LDS as [col][row] and write rows of 8 consecutive doubles (the two waves fill the other half).
    __syncthreads();  // both waves are done reading the reduction buffer
    // one shared staging image [block][col][row]: wave WV contributes its rows 8WV..8WV+7 of every block
    double* tp = smem;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) tp[((a * N + b) * 16 + fr) * C::TLD16 + fk + 4 * (WV * 2 + q)] = x[a][b][q];
    __syncthreads();
    // wave WV writes the mirrored rows cc in [8 WV, 8 WV + 8): lane -> (cc, rr0 = 2 (l & 7)), one double2 per block:
    // 8 full 128-byte rows per store instruction
    const int cc = 8 * WV + (l >> 3), rr0 = (l & 7) * 2;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) {
            double2 v2;
            v2.x = tp[((a * N + b) * 16 + cc) * C::TLD16 + rr0];
            v2.y = tp[((a * N + b) * 16 + cc) * C::TLD16 + rr0 + 1];
            *reinterpret_cast<double2*>(Pout + ((long)b * dp + K * 16 + cc) * Dp + (long)a * dp + J * 16 + rr0) = v2;
        }
}

template <int N>
__global__ __launch_bounds__(256, (N <= 3 ? 4 : 1)) void k_downdate(const double* Ppred, const double* __restrict__ W, double* Pout,
                                                     double* __restrict__ var, int dp, int mp, int c_begin,
                                                     int c_end, VecArgs va) {
    using C = DowndateCfg<N>;
    __shared__ __attribute__((aligned(16))) double smem[C::LDS_D];
    static_assert(N * N * 16 * C::TLD16 <= C::LDS_D, "transpose staging must fit");
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const long Dp = (long)N * dp;
    if ((int)blockIdx.y >= dp / 16) {  // vector-op rows ride in the same launch (4 rows per workgroup)
        vecops_rows(va, W, mp, Dp, ((long)(blockIdx.y - dp / 16) * (dp / 16) + blockIdx.x) * 4 + w, l);
        return;
    }
    const int J = blockIdx.y, K = blockIdx.x;
    if (K > J) return;
    double* st = smem + w * C::STG;
    const int fr = l & 15, fk = l >> 4;
    const int lr = l >> 2, lc = (l & 3) * 2;  // staging load: 16 rows x 8 cols per instruction

    d4 acc[N][N];
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
        for (int b = 0; b < N; ++b) acc[a][b] = (d4){0, 0, 0, 0};

    const double* wa = W + ((long)J * 16 + lr) * mp + lc;
    const double* wb = W + ((long)K * 16 + lr) * mp + lc;
    const long astride = (long)dp * mp;
    for (int chunk = c_begin + w; chunk < c_end; chunk += 4) {
        double2 pa[N], pb[N];
#pragma unroll
        for (int a = 0; a < N; ++a) {
            pa[a] = *reinterpret_cast<const double2*>(wa + a * astride + chunk * C::CW);
            pb[a] = *reinterpret_cast<const double2*>(wb + a * astride + chunk * C::CW);
        }
        // registers -> wave-private LDS (the previous chunk's fragment reads are older in this wave's LDS queue)
#pragma unroll
        for (int a = 0; a < N; ++a) {
            st[(a * 16 + lr) * C::SLD + lc] = pa[a].x;
            st[(a * 16 + lr) * C::SLD + lc + 1] = pa[a].y;
            st[((N + a) * 16 + lr) * C::SLD + lc] = pb[a].x;
            st[((N + a) * 16 + lr) * C::SLD + lc + 1] = pb[a].y;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < C::CW / 4; ++kk) {
            double fa[N], fb[N];
#pragma unroll
            for (int a = 0; a < N; ++a) {
                fa[a] = st[(a * 16 + fr) * C::SLD + kk * 4 + fk];
                fb[a] = st[((N + a) * 16 + fr) * C::SLD + kk * 4 + fk];
            }
#pragma unroll
            for (int a = 0; a < N; ++a)
#pragma unroll
                for (int b = 0; b < N; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    // round 1: waves 2,3 -> LDS, waves 0,1 accumulate
    __syncthreads();
    if (w >= 2) {
        double* rd = smem + (w - 2) * C::RED;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) rd[((a * N + b) * 4 + r) * 64 + l] = acc[a][b][r];
    }
    __syncthreads();
    if (w >= 2) return;
    {
        const double* rd = smem + w * C::RED;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] += rd[((a * N + b) * 4 + r) * 64 + l];
    }
    // round 2 (waves 0,1 only; exited waves do not take part in the barrier): wave 0 finalises accumulator
    // registers 0,1 and wave 1 registers 2,3 -- two instantiations so that every register index is static
    __syncthreads();
    if (w == 0)
        downdate_finish<N, 0>(acc, smem, Ppred, Pout, var, dp, Dp, J, K, l);
    else
        downdate_finish<N, 1>(acc, smem, Ppred, Pout, var, dp, Dp, J, K, l);
}

// ------------------------------------------------------------------------------------------
// On-device error model (estimate_error, white.py:153-162).  Sq = H (Ql Ql^T) H^T + E E^T is the innovation matrix of
// a filter whose predicted covariance is Q = Q1 (x) K, so the same k_front / k_sweep pair factorises it:
// [Sq; I] -> [Lq; Lq^-T], and Sq^-1 = Lq^-T Lq^-1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_q(void* __restrict__ Qf, const double* __restrict__ Kg, IwpConsts c, int n,
                                                int dp, int p32) {
    const long Dp = (long)n * dp;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= Dp * Dp) return;
    const long row = e / Dp, col = e % Dp;
    const int a = (int)(row / dp), j = (int)(row % dp), b = (int)(col / dp), k = (int)(col % dp);
    const double v = c.Q1[a * MAXN + b] * Kg[(long)j * dp + k];
    if (p32) static_cast<float*>(Qf)[e] = (float)v;
    else static_cast<double*>(Qf)[e] = v;
}

// C = T T^T for a row-major (mp x mp) T (rows dot rows), 16x16 outputs per workgroup
__global__ __launch_bounds__(256) void k_ttt(const double* __restrict__ T, double* __restrict__ C, int mp) {
    __shared__ double sa[16][17], sb[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, k = blockIdx.x * 16 + tx;
    double acc = 0.0;
    for (int q0 = 0; q0 < mp; q0 += 16) {
        sa[ty][tx] = T[(long)(blockIdx.y * 16 + ty) * mp + q0 + tx];
        sb[ty][tx] = T[(long)(blockIdx.x * 16 + ty) * mp + q0 + tx];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sa[ty][q] * sb[tx][q];
        __syncthreads();
    }
    C[(long)i * mp + k] = acc;
}

// per-step read-out of the parity quantities (experiments/figure1.py:76-80), raw coordinates;
// block 0 also reduces the partial sums of the vector-op rows into rec[0..2] (fixed order: deterministic)
template <int N, bool ROLE>
__global__ __launch_bounds__(256) void k_readout(const double* __restrict__ mean, const double* __restrict__ var,
                                                 double* __restrict__ means_base, double* __restrict__ stds_base,
                                                 double s0, int d, const double* __restrict__ part,
                                                 double* __restrict__ rec_base, int mp, int* __restrict__ ctr,
                                                 RoleArgs ra, int* __restrict__ tickets, int rblocks,
                                                 const void* __restrict__ Ppred, const double* __restrict__ rdiag,
                                                 const double* __restrict__ Rdense, int p32) {
    if constexpr (ROLE) {
        // blocks behind the read-out and role blocks: the NEXT step's k_front (G from the P- the down-date epilogue
        // has just written) -- independent of the rest of this launch, which hides behind it
        if ((int)blockIdx.x > rblocks) {
            const int fb = blockIdx.x - rblocks - 1;
            if (p32) front_block_tiled(static_cast<const float*>(Ppred), ra.G, rdiag, Rdense, ra.mm, (long)N * ra.dp, fb);
            else front_block_tiled(static_cast<const double*>(Ppred), ra.G, rdiag, Rdense, ra.mm, (long)N * ra.dp, fb);
            return;
        }
    }
    const int slot = __builtin_amdgcn_readfirstlane(*ctr) - 1;  // (the load has completed before the ticket below)
    if constexpr (ROLE) {
        // Constant-step loop: the last block prepares the NEXT step (the vector part of its predict, whose covariance
        // part the down-date epilogue has already done) beside this step's read-out blocks.  It moves the step counter
        // only after every read-out block has taken its slot (ticket counter; those blocks have lower indices).
        if ((int)blockIdx.x == rblocks) {
            extern __shared__ double mpl[];
            predict_vectors<N>(mpl, threadIdx.x, ra.c, ra.dp, ra.min, ra.mpred, ra.shift, ra.G, ra.zbuf, ra.mm, ra.flags,
                               ra.nflags);
            if (threadIdx.x == 0) {
                int spins = 0;
                while (__hip_atomic_load(tickets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rblocks &&
                       ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                *ctr = slot + 2;
                __hip_atomic_store(tickets, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        __syncthreads();  // (slot read by all threads of this block)
        if (threadIdx.x == 0) atomicAdd(tickets, 1);
    }
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    double* means = means_base ? means_base + (size_t)slot * d : nullptr;
    double* stds = stds_base ? stds_base + (size_t)slot * d : nullptr;
    double* rec = rec_base + 4 * (size_t)slot;
    if (j < d) {
        if (means) means[j] = s0 * mean[j];
        if (stds) stds[j] = s0 * sqrt(fmax(var[j], 0.0));
    }
    if (blockIdx.x == 0) {
        __shared__ double sred[3][4];
        double a[3] = {0.0, 0.0, 0.0};
        for (int i = threadIdx.x; i < mp; i += 256)
#pragma unroll
            for (int q = 0; q < 3; ++q) a[q] += part[q * mp + i];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a[q] += __shfl_xor(a[q], o);
        if ((threadIdx.x & 63) == 0)
            for (int q = 0; q < 3; ++q) sred[q][threadIdx.x >> 6] = a[q];
        __syncthreads();
        if (threadIdx.x < 3) rec[threadIdx.x] = sred[threadIdx.x][0] + sred[threadIdx.x][1] + sred[threadIdx.x][2] + sred[threadIdx.x][3];
    }
}

// Test hook (pnmol_filter_debug_poison): NaN into every buffer the sweep hands data over through -- F, the lower parts of
// the L_jj^-1 tiles (their upper-right quadrants are zero by contract and never written), the feed / scratch tiles -- by
// plain stores from workgroups on all XCDs, i.e. as a previous launch would have left them in the L2s and in memory.
__global__ __launch_bounds__(256) void k_poison(double* __restrict__ F, long nF, double* __restrict__ Linv, long nL,
                                                double* __restrict__ hs, long nH) {
    const double bad = __longlong_as_double(0x7ff8dead0000beefLL);
    const long stride = (long)gridDim.x * 256;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nF; e += stride) F[e] = bad;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nL; e += stride) {
        const int r = (int)((e / NB) % NB), c = (int)(e % NB);
        if (!(r < 16 && c >= 16)) Linv[e] = bad;
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nH; e += stride) hs[e] = bad;
}

// M = L + diag(j) on the device (semilinear EK1 with a pointwise nonlinearity, white.py:192-208: J_x = diag(df/du)): the
// ELL image of Hv = [-M; B] is the base image with -j_i added to the diagonal entry of PDE row i; the shift comes along.
__global__ void k_operator_diagonal(double* __restrict__ ell_val, const double* __restrict__ base_val,
                                    const int* __restrict__ slot, const double* __restrict__ op /* [jdiag d | shift mp] */,
                                    double* __restrict__ shift, int d, int mp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= mp) return;
    if (i < d) {
        const int e = slot[i];
        ell_val[(long)e * mp + i] = base_val[(long)e * mp + i] - op[i];
    }
    shift[i] = op[d + i];
}

// Batched kernel finite-difference stencils (discretize.py:177-201): per mesh point p an s x s system
//     w_p = (k(X_p, X_p) + eta I)^-1 Lk(x_p, X_p),     u_p = LLk(x_p, x_p) - w_p . Lk(x_p, X_p)
// (X_p = the s stencil neighbours; the reference vmaps jnp.linalg.solve over the points).  One thread per point: LU with
// partial pivoting in registers, like LAPACK's getf2 on the s x s matrix (same pivot rule, same update order).
constexpr int FD_MAXS = 16;
__global__ __launch_bounds__(128) void k_fd_solve(const double* __restrict__ gram, const double* __restrict__ dk,
                                                  const double* __restrict__ llk, int N, int s, double* __restrict__ w,
                                                  double* __restrict__ unc) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    double A[FD_MAXS][FD_MAXS], b[FD_MAXS];
    for (int i = 0; i < s; ++i) {
        for (int k = 0; k < s; ++k) A[i][k] = gram[((long)p * s + i) * s + k];
        b[i] = dk[(long)p * s + i];
    }
    for (int c = 0; c < s; ++c) {
        int piv = c;
        double best = fabs(A[c][c]);
        for (int i = c + 1; i < s; ++i)
            if (fabs(A[i][c]) > best) best = fabs(A[i][c]), piv = i;
        if (piv != c) {
            for (int k = 0; k < s; ++k) {
                const double t = A[c][k];
                A[c][k] = A[piv][k], A[piv][k] = t;
            }
            const double t = b[c];
            b[c] = b[piv], b[piv] = t;
        }
        const double inv = 1.0 / A[c][c];
        for (int i = c + 1; i < s; ++i) {
            const double f = A[i][c] * inv;
            A[i][c] = f;
            for (int k = c + 1; k < s; ++k) A[i][k] -= f * A[c][k];
        }
    }
    for (int i = 1; i < s; ++i)          // L y = P b
        for (int k = 0; k < i; ++k) b[i] -= A[i][k] * b[k];
    for (int i = s - 1; i >= 0; --i) {   // U x = y
        for (int k = i + 1; k < s; ++k) b[i] -= A[i][k] * b[k];
        b[i] /= A[i][i];
    }
    double dot = 0.0;
    for (int i = 0; i < s; ++i) {
        w[(long)p * s + i] = b[i];
        dot += b[i] * dk[(long)p * s + i];
    }
    unc[p] = llk[p] - dot;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#define HIPCHK(ctx, call)                                                                        \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                     \
            return -2;                                                                           \
        }                                                                                        \
    } while (0)

static inline int round_up(int x, int q) { return (x + q - 1) / q * q; }

static double nordsieck_scale(int nu, int a, double dt) {  // base/iwp.py:55-62
    double fact = 1.0;
    for (int q = 2; q <= nu - a; ++q) fact *= q;
    return std::pow(std::fabs(dt), nu - a + 0.5) / fact;
}

}  // namespace

// struct pnmol_ctx: pnmol_internal.hpp

// ------------------------------------------------------------------------------------------
// Large problems (the 64x64 mesh of BASELINE config 5: D = 8192, m = 4348): P = P- - W W^T as ONE plain SYRK over the D x D
// matrix -- the derivative-block structure plays no role in it -- on 128x128 tiles.  The pairs of 32-point tiles that ride
// along in k_sweep read both operands per 32x32 tile (36 GB per step at that size, on top of the 33 GB of the sweep's own
// re-reads, all L2 misses: the fused launch ran at the fabric's rate, 18.2 ms), and the stand-alone k_downdate
// (16-point tiles) needs 11.6 ms.  Here a workgroup (4 waves, each a 64x64 quadrant = 4x4 MFMA tiles, 128 accumulator
// registers) streams 16-column slabs of its two 128-row panels of W through LDS, K-major (conflict-free fragment reads:
// the 16 lanes of a fragment row read 16 consecutive doubles), double-buffered, ONE barrier per slab of 64 MFMAs per wave,
// two workgroups per CU.  Lower tiles only; the mirror image leaves through a wave-private LDS transpose as 128-byte rows.
constexpr int BDT = 128;  // tile
constexpr int BDK = 16;   // columns of W per slab
// PT = float: the covariance is fp32 (pnmol_filter_desc.dtype = 1): W is rounded to fp32 on its way into LDS, the products run on
// v_mfma_f32_16x16x4_f32 with fp32 accumulators (the arithmetic of the fused pairs' fp32 form), half the LDS, half the registers.
template <typename PT>
__global__ __launch_bounds__(256, 2) void k_downdate_big(const PT* Ppred, const double* __restrict__ W, PT* Pout,
                                                         double* __restrict__ var, long Dp, int mp, int nt) {
    typedef typename AccOf<PT>::type acc_t;
    __shared__ __attribute__((aligned(16))) PT sA[2][BDK][BDT];
    __shared__ __attribute__((aligned(16))) PT sB[2][BDK][BDT];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, fr = l & 15, fk = l >> 4;
    // tile (I, J), J <= I.  Consecutive tiles of the row-major order go to the SAME XCD (blockIdx.x round-robins over the
    // eight XCDs): the workgroups that share an L2 share the panel I and walk neighbouring panels J.
    // (the launch has 8 * ceil(nt / 8) workgroups)
    const int per = (int)gridDim.x / 8;
    const int t = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (t >= nt) return;
    int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    while (I * (I + 1) / 2 > t) --I;
    const int J = t - I * (I + 1) / 2;
    const long row0 = (long)I * BDT + wr * 64, col0 = (long)J * BDT + wc * 64;
    acc_t acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ti][tj][r] = Ppred[(row0 + 16 * ti + AccOf<PT>::row(fk, r)) * Dp + col0 + 16 * tj + fr];
    // loader: thread -> (row tid / 2 of the panel, columns 8 (tid & 1) .. +7 of the slab)
    const int lrow = tid >> 1, lk = (tid & 1) * 8;
    const double* ga = W + ((long)I * BDT + lrow) * mp + lk;
    const double* gb = W + ((long)J * BDT + lrow) * mp + lk;
    double2 ra[4], rb[4];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const double2*>(ga + (long)slab * BDK + 2 * q);
            rb[q] = *reinterpret_cast<const double2*>(gb + (long)slab * BDK + 2 * q);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sA[buf][lk + 2 * q][lrow] = (PT)ra[q].x;
            sA[buf][lk + 2 * q + 1][lrow] = (PT)ra[q].y;
            sB[buf][lk + 2 * q][lrow] = (PT)rb[q].x;
            sB[buf][lk + 2 * q + 1][lrow] = (PT)rb[q].y;
        }
    };
    const int nslab = mp / BDK;
    fetch(0);
    park(0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        if (sl + 1 < nslab) fetch(sl + 1);  // (in flight under this slab's MFMAs)
#pragma unroll
        for (int ks = 0; ks < BDK / 4; ++ks) {
            PT a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = -sA[buf][4 * ks + fk][wr * 64 + 16 * q + fr];
                b[q] = sB[buf][4 * ks + fk][wc * 64 + 16 * q + fr];
            }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
                    if constexpr (sizeof(PT) == 8) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
                    else acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
                }
        }
        // (buffer buf ^ 1 was read in the previous iteration: everybody is past that iteration's barrier)
        if (sl + 1 < nslab) park(buf ^ 1);
        __syncthreads();
    }
    // the tile, diag(P)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + 16 * ti + AccOf<PT>::row(fk, r), col = col0 + 16 * tj + fr;
                Pout[row * Dp + col] = acc[ti][tj][r];
                if (row == col) var[row] = acc[ti][tj][r];
            }
    if (I == J) return;  // (a diagonal tile holds both of its triangles)
    // mirror image: strips of 16 rows x 64 columns, transposed through this wave's quarter of the slab buffers
    PT* stg = (w < 2 ? &sA[0][0][0] : &sB[0][0][0]) + (w & 1) * (BDK * BDT);  // 2048 elements of LDS per wave
#pragma unroll  // (a run-time index into acc would put all accumulator registers into scratch memory)
    for (int ti = 0; ti < 4; ++ti) {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) stg[(16 * tj + fr) * 17 + AccOf<PT>::row(fk, r)] = acc[ti][tj][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = fk + 4 * q;  // column of the tile = row of the mirror image
            Pout[(col0 + c) * Dp + row0 + 16 * ti + fr] = stg[c * 17 + fr];
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// W = (P- H^T) T,  T = Ls^-T as the sweep leaves it in the identity rows of F (every row block of the tall matrix is
// right-multiplied by the same operator, so the rows of W need not go through the sweep at all: on the 64x64 mesh they were
// 256 of its 529 workgroups and 155 of its 210 GFLOP, at the left-looking kernel's 21 TF).  One GEMM on 128x128 tiles with
// k_downdate_big's pipeline: A = 16-column slabs of 128 rows of P- H^T (K-major in LDS), B = 16 ROWS of T x 128 columns (K-major
// as they lie in memory); T is upper triangular: column tile J needs k < 128 (J + 1) only, the tiles are dealt by falling J.
__global__ __launch_bounds__(256, 2) void k_w_gemm(const double* __restrict__ A, const double* __restrict__ T, double* __restrict__ C,
                                                   int ld, int nI, int nt) {
    __shared__ __attribute__((aligned(16))) double sA[2][BDK][BDT];
    __shared__ __attribute__((aligned(16))) double sB[2][BDK][BDT];
    const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 1, wc = w & 1, fr = l & 15, fk = l >> 4;
    // (tiles in dispatch order = by falling work, round-robin over the XCDs: with consecutive tiles on ONE XCD -- as in
    //  k_downdate_big, where all tiles cost the same -- XCD 0 got the 272 longest tiles and the launch took 5.3 ms for 2.9 ms of work)
    const int t = (int)blockIdx.x;
    if (t >= nt) return;
    const int nJ = nt / nI, J = nJ - 1 - t / nI, I = t % nI;
    d4 acc[4][4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = (d4){0, 0, 0, 0};
    const int lrow = tid >> 1, lk = (tid & 1) * 8;        // A: row of the panel, 8 columns of the slab
    const int bk = tid >> 4, bj = (tid & 15) * 8;         // B: row of the slab, 8 columns of the tile
    const double* ga = A + ((long)I * BDT + lrow) * ld + lk;
    const double* gb = T + (long)bk * ld + (long)J * BDT + bj;
    double2 ra[4], rb[4];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const double2*>(ga + (long)slab * BDK + 2 * q);
            rb[q] = *reinterpret_cast<const double2*>(gb + (long)slab * BDK * ld + 2 * q);
        }
    };
    auto park = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sA[buf][lk + 2 * q][lrow] = ra[q].x;
            sA[buf][lk + 2 * q + 1][lrow] = ra[q].y;
            *reinterpret_cast<double2*>(&sB[buf][bk][bj + 2 * q]) = rb[q];
        }
    };
    const int nslab = (J + 1) * BDT / BDK;
    fetch(0);
    park(0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        if (sl + 1 < nslab) fetch(sl + 1);
#pragma unroll
        for (int ks = 0; ks < BDK / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = sA[buf][4 * ks + fk][wr * 64 + 16 * q + fr];
                b[q] = sB[buf][4 * ks + fk][wc * 64 + 16 * q + fr];
            }
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
        if (sl + 1 < nslab) park(buf ^ 1);
        __syncthreads();
    }
    const long row0 = (long)I * BDT + wr * 64, col0 = (long)J * BDT + wc * 64;
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(row0 + 16 * ti + fk + 4 * r) * ld + col0 + 16 * tj + fr] = acc[ti][tj][r];
}
// the vector ops of the step as a launch of their own (k_downdate carries them in extra blockIdx.y rows)
__global__ __launch_bounds__(256) void k_vecops(VecArgs va, const double* __restrict__ W, int mp, long Dp, int per_row) {
    vecops_rows(va, W, mp, Dp, ((long)blockIdx.y * per_row + blockIdx.x) * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

struct pnmol_filter {
    pnmol_ctx* ctx = nullptr;
    int d = 0, n = 0, nu = 0, nB = 0, m = 0, dp = 0, mp = 0, CB = 0, RBS = 0, RBW = 0, RT = 0, ellw = 0;
    int* tickets = nullptr;   // device: read-out blocks that have taken their slot (k_readout with the next step's role)
    int* last_ctr = nullptr;  // device: step-counter value of the last step of the running pnmol_filter_steps call
    int fuse_predict = 1;     // PNMOL_HIP_FUSE_PREDICT: predict the next step's covariance in the down-date epilogue
    int* flags = nullptr;  // k_sweep dependency flags: row[RT], diag[CB], abort, claim[CB*CB] (helpers); k_sweep_rl: rl_flags()
    int nflags = 0;        // words allocated (all of them are zeroed before every sweep)
    double* hs_scratch = nullptr;  // helpers' partial sums, one tile per (row, target step)
    int w_gemm = 0;        // ... and W = (P- H^T) Ls^-T as a GEMM behind a sweep without the rows of W (k_w_gemm)
    int dd_big = 0;        // large problems: sweep alone + k_downdate_big (PNMOL_HIP_DD_BIG=0/1 overrides; see there)
    int sweep_mode = 2;    // PNMOL_HIP_SWEEP: 2 = k_sweep with the covariance down-date riding along in the same launch,
                           // 1 = k_sweep, then k_downdate; 0 = k_diag0 + one k_panel launch per panel, then k_downdate
    int ds = 0;  // spatial components of the state (= d, or 2d for the latent-force model [u; eps])
    bool counted = false;  // this filter is in live_rl_filters[device]
    bool registered = false;  // this filter is counted in ctx->children
    std::atomic<int> states{0};  // live pnmol_state objects of this filter (pnmol_filter_destroy refuses while > 0)
    int xcd_home = -1;  // k_sweep_rl: >= 0: XCD-local layout (XL), chain workgroup and S row blocks on this XCD; -1: spread layout
    int p32 = 0;        // pnmol_filter_desc.dtype = 1: covariances (state, predicted, Q) are stored and down-dated in fp32
    size_t psz = 8;     // bytes per covariance element
    long Dp = 0;
    IwpConsts iwp{};
    int* ell_col = nullptr;
    double* ell_val = nullptr;
    // the operator given at creation (pde.L), kept for pnmol_filter_set_operator_diagonal: its ELL image, the slot of the
    // diagonal entry of every PDE row (-1: the row has none), and a pinned staging buffer [jdiag d | shift mp]
    int* ell_col_base = nullptr;
    double* ell_val_base = nullptr;
    int* ell_diag_slot = nullptr;
    int base_w = 0, base_has_diag = 0;
    double* h_op = nullptr;
    double* h_op_dev = nullptr;
    hipEvent_t ev_op = nullptr;
    double *Kg = nullptr, *rdiag = nullptr, *Rdense = nullptr, *shift = nullptr;
    double *G = nullptr, *F = nullptr, *Linv = nullptr, *Ppred = nullptr, *mpred = nullptr, *zbuf = nullptr;
    double *var = nullptr, *Sqinv = nullptr, *rec = nullptr, *part = nullptr, *sdiag = nullptr;
    int* info = nullptr;
    std::vector<double> sqdiag;
    double* Qfull = nullptr;  // Q1 (x) K as a dense Dp x Dp matrix (on-device error model), allocated on first use
    int* one = nullptr;       // device constant 1 (step-counter stand-in for sweeps outside the step loop)
    int* info_err = nullptr;  // info word of those sweeps
    double sq_dt = -1.0;
    std::vector<double> hB;   // host copy of pde.B (nB x d) for operator rebuilds
    int ell_cap = 0;          // allocated ELL width
    // scratch state for ping-pong inside steps()
    double *tmpP = nullptr, *tmpMean = nullptr;
    double *rec_means = nullptr, *rec_stds = nullptr;
    double* h_pin = nullptr;  // pinned host staging: [rec 4k | means k*d | stds k*d | info k ints]
    double* h_pin_dev = nullptr;  // the same buffer as the device sees it (mapped)
    int rec_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = 0.f;
    int* ctr = nullptr;  // device step counter (slot of the per-step outputs)
    int pending_k = 0;   // steps enqueued by pnmol_filter_steps_begin and not yet collected
    double pending_dt = 0.0;
    pnmol_state* pending_state = nullptr;
    struct GraphEntry {
        double *P0, *P1, *var;
        double dt;
        int nsteps;
        bool have_sq;
        bool fused;
        hipGraphExec_t exec;
        bool launched;  // the first launch of an executable graph costs ~50 ms of host time (ROCm 7.2)
    };
    std::vector<GraphEntry> graphs;
    int graph_chunk = 10;  // steps per captured graph (even); 0 disables graphs

};

struct pnmol_state {
    pnmol_filter* f = nullptr;
    double* mean = nullptr;  // Dp
    double* P = nullptr;     // Dp*Dp
    double* var = nullptr;   // Dp   marginal variances, same frame as P
    double t = 0.0;
    double frame_dt = 0.0;  // 0 = raw coordinates, else Nordsieck frame of that dt
};

namespace {

void drop_graphs(pnmol_filter* f);

// kind of a step inside the launch sequence: FULL = self-contained (predict, ..., P written); in the constant-step loop
// of pnmol_filter_steps the down-date epilogue already predicts the next step's covariance (fused_loop()), so FIRST does
// the full predict and STEADY steps only run k_predict's vector workgroup; both write P only in the call's last step.
enum StepKind { STEP_FULL = 0, STEP_FIRST = 1, STEP_STEADY = 2 };

inline bool fused_loop(const pnmol_filter* f) { return f->sweep_mode == 2 && f->n <= 3 && f->fuse_predict != 0; }

// the sweep launch: right-looking register-resident kernel where the row block fits the registers (CB <= 17: N <= 512 in 1-d),
// the left-looking one otherwise (PNMOL_HIP_SWEEP_RL=0 forces it, for A/B runs)
inline bool sweep_rl_enabled() {
    static const int on = [] {
        const char* e = std::getenv("PNMOL_HIP_SWEEP_RL");
        return e ? std::atoi(e) : 1;
    }();
    return on != 0;
}
// PNMOL_HIP_SWEEP_XL: 1 / 0 force the XCD-local layout on / off for every filter; unset: the first live filter of the process
// gets it.  It trades throughput of SEVERAL sweeps in flight (their chain workgroups and S row blocks are pinned to one XCD
// each and everything behind them in dispatch order waits for a CU there: 7.3 k steps/s with eight problems in flight
// against 8.7 k with the spread layout) for latency of one (159 against 170 us per step).
inline int sweep_xl_mode() {
    static const int mode = [] {
        const char* e = std::getenv("PNMOL_HIP_SWEEP_XL");
        return e ? (std::atoi(e) != 0 ? 1 : 0) : -1;
    }();
    return mode;
}
// live filters that launch k_sweep_rl, per device: the first one of a device gets the XCD-local layout (see pnmol_filter_create)
constexpr int MAX_DEVICES = 64;
std::atomic<int> live_rl_filters[MAX_DEVICES];
template <int N, bool FUSED>
void launch_sweep(unsigned grid, hipStream_t st, const double* G, double* F, double* Linv, int ld, int CB, int RT, int* flags,
                  int* info, const int* ctr, const DowndateArgs& dd, int* claim, double* hs, int lenient, int home,
                  bool wskip = false) {
    // k_sweep_rl: block 0 is the chain workgroup; `hs` (>= 2 CB tiles) carries what the chain rows feed it; the CB flags
    // behind the abort word (`claim`) say so
    // XL (row blocks of S on the chain workgroup's XCD: blocks 8 s, the blocks between them empty)
    const bool xl = home >= 0;
    // k_sweep_rl's grid: the caller's `grid` workgroups (RT row blocks [+ pairs]) + the parts B of the split row blocks of W
    // (logical indices RT .. RT + nw - 1, in front of the pairs) + (XL) the parts B of the split row blocks of S
    const unsigned rgrid = grid + (unsigned)w_split_count(RT, CB);
    unsigned xgrid = 8u * (unsigned)(CB - 1 + s_split_count(CB, true)) + 1u;
    for (unsigned left = rgrid - (unsigned)CB; left > 0; ++xgrid)  // the other workgroups, skipping the blocks = 0 mod 8
        if (xgrid % 8u != 0u) --left;
    xgrid += (unsigned)(xl ? home : 0);
    if (sweep_rl_enabled() && CB <= 9) {
        if (xl) k_sweep_rl<N, FUSED, 9, true><<<xgrid, 256, 0, st>>>(G, F, Linv, ld, CB, RT, flags, info, ctr, dd, hs, lenient, home);
        else k_sweep_rl<N, FUSED, 9, false><<<rgrid + 1, 256, 0, st>>>(G, F, Linv, ld, CB, RT, flags, info, ctr, dd, hs, lenient, home);
    } else if (sweep_rl_enabled() && CB <= 17) {
        if (xl) k_sweep_rl<N, FUSED, 17, true><<<xgrid, 256, 0, st>>>(G, F, Linv, ld, CB, RT, flags, info, ctr, dd, hs, lenient, home);
        else k_sweep_rl<N, FUSED, 17, false><<<rgrid + 1, 256, 0, st>>>(G, F, Linv, ld, CB, RT, flags, info, ctr, dd, hs, lenient, home);
    }
    else  // (a 33-tile row block no longer fits the register file: measured 1280 us against 666 at N = 1024)
        k_sweep<N, FUSED><<<grid, 256, 0, st>>>(G, F, Linv, ld, CB, RT, flags, info, ctr, dd, claim, hs, lenient, wskip ? 1 : 0);
}

inline bool short_last_panel() {  // PNMOL_HIP_SHORT_LAST=0: A/B switch
    static const int on = [] {
        const char* e = std::getenv("PNMOL_HIP_SHORT_LAST");
        return e ? std::atoi(e) : 1;
    }();
    return on != 0;
}
template <int N>
int launch_step(pnmol_filter* f, const double* Pin, const double* min, double frame_dt, double dt, double* Pout,
                double* mout, double* varout, bool record, StepKind kind) {
    pnmol_ctx* ctx = f->ctx;
    hipStream_t st = ctx->stream;
    IwpConsts c = f->iwp;
    for (int a = 0; a < f->n; ++a) {
        const double so = frame_dt == 0.0 ? 1.0 : nordsieck_scale(f->nu, a, frame_dt);
        c.ts[a] = so / nordsieck_scale(f->nu, a, dt);
    }
    MeasModel mm{f->ell_col, f->ell_val, f->ellw, f->d, f->m, f->dp, f->mp,
                 nordsieck_scale(f->nu, 0, dt), nordsieck_scale(f->nu, 1, dt)};
    const int dp = f->dp, mp = f->mp;
    const long Dp = f->Dp;

    // K1: P- = A P A^T + Q  (+ one workgroup: m-, z, step counter)
    // (STEADY steps of the constant-step loop need none of it: the previous step's down-date epilogue made P- and its
    //  k_readout launch did the vector part)
    if (kind != STEP_STEADY)
        k_predict<N><<<dim3(dp / 32, dp / 8 + 1), dim3(32, 8), sizeof(double) * Dp, st>>>(
        Pin, f->Ppred, f->Kg, c, dp, min, f->mpred, f->shift, f->G, f->zbuf, mm, f->ctr, f->flags, f->nflags, f->p32);
    // K2: G = [S; P-H^T; z; I]  (STEADY: done by the previous step's k_readout launch)
    if (kind != STEP_STEADY) {
        if (f->ellw > HW && f->sweep_mode != 0) {
            // wide stencils (and sweep kernels that leave G alone): the rows of P- H^T first, S from them (k_front_s)
            k_front<<<dim3((mp + 255) / 256, (unsigned)Dp), 256, 0, st>>>(f->Ppred, f->G, f->rdiag, f->Rdense, mm, Dp, f->p32);
            k_front_s<<<dim3((mp + 255) / 256, (unsigned)mp), 256, 0, st>>>(f->G, f->rdiag, f->Rdense, mm);
        } else {
            k_front<<<dim3((mp + 255) / 256, (unsigned)(Dp + mp + (f->sweep_mode == 0 ? mp : 0))), 256, 0, st>>>(
                f->Ppred, f->G, f->rdiag, f->Rdense, mm, Dp, f->p32);  // (the per-panel path updates G in place: identity block too)
        }
    }
    const long rowI0 = (long)mp + Dp + NB;
    const double* W = f->F + (long)mp * mp;
    const bool have_sq = (f->Sqinv != nullptr && f->sq_dt == dt);
    VecArgs va{f->mpred, f->F + ((long)mp + Dp) * mp, f->F + rowI0 * mp, f->zbuf, have_sq ? f->Sqinv : nullptr,
               mout, f->part};
    const int nreal = f->m - NB * (f->CB - 1);  // real columns of the last column block
    DowndateArgs dd{f->Ppred, Pout, varout, dp, 0, va, kind == STEP_FULL ? nullptr : f->Ppred, f->Kg, f->last_ctr, {}, {}, f->p32,
                    (nreal < 8 && short_last_panel()) ? nreal : 0};
    std::memcpy(dd.A1, f->iwp.A1, sizeof(dd.A1));
    std::memcpy(dd.Q1, f->iwp.Q1, sizeof(dd.Q1));
    if (f->sweep_mode == 2) {
        // K3'+K4: the whole sweep as one dataflow launch (one workgroup per 32-row block) with the covariance
        // down-date riding along (one workgroup per pair of 32-point tiles), then the vector ops
        const int t32 = dp / NB, pairs = t32 * (t32 + 1) / 2;
        dd.vrows = 0;  // (the vector ops are done by the row-block workgroups of the sweep themselves)
        if constexpr (N <= 3)
            launch_sweep<N, true>(f->RT + pairs, st, f->G, f->F, f->Linv, mp, f->CB, f->RT, f->flags, f->info, f->ctr, dd,
                                  f->flags + f->RT + f->CB + 1, f->hs_scratch, f->p32 ? 2 : 0, f->xcd_home);
    } else {
        if (f->sweep_mode == 1) {
            // K3': the sweep alone as one dataflow launch
            if (f->w_gemm) {
                // rows of S, r^T, rows of Ls^-T through the sweep; W = (P- H^T) Ls^-T as a GEMM behind it
                launch_sweep<N, false>(2 * f->CB + 1, st, f->G, f->F, f->Linv, mp, f->CB, f->RT, f->flags, f->info, f->ctr, dd,
                                       f->flags + f->RT + f->CB + 1, f->hs_scratch, f->p32 ? 2 : 0, f->xcd_home, true);
                const int nI = (int)(Dp / BDT), nt = nI * (mp / BDT);
                k_w_gemm<<<8 * ((nt + 7) / 8), 256, 0, st>>>(f->G + (long)mp * mp, f->F + rowI0 * mp, f->F + (long)mp * mp, mp, nI, nt);
            } else
            launch_sweep<N, false>(f->RT, st, f->G, f->F, f->Linv, mp, f->CB, f->RT, f->flags, f->info, f->ctr, dd,
                                   f->flags + f->RT + f->CB + 1, f->hs_scratch, f->p32 ? 2 : 0, f->xcd_home);
        } else {
            k_diag0<<<1, 128, 0, st>>>(f->G, f->F, f->Linv, mp, f->info, f->sdiag, f->ctr);
            // K3: right-looking sweep, one launch per 32-column panel
            for (int j = 0; j < f->CB; ++j) {
                const int nrb = f->RT - (j + 1);
                const int ncb = f->CB - 1 - j > 0 ? f->CB - 1 - j : 1;
                k_panel<<<dim3(nrb, ncb), 256, 0, st>>>(f->G, f->F, f->Linv, mp, j, f->CB, f->RBS, f->info, f->sdiag, f->ctr);
            }
        }
        // K4: P = P- - W W^T (tiles) and, in extra blockIdx.y rows of the same launch, the vector ops
        const int tiles = dp / 16;
        const int vrows = (int)(((Dp + mp + 3) / 4 + tiles - 1) / tiles);
        if (f->dd_big) {
            const int T = (int)(Dp / BDT), nt = T * (T + 1) / 2;
            if (f->p32)
                k_downdate_big<float><<<8 * ((nt + 7) / 8), 256, 0, st>>>(reinterpret_cast<const float*>(f->Ppred), W,
                                                                          reinterpret_cast<float*>(Pout), varout, Dp, mp, nt);
            else
                k_downdate_big<double><<<8 * ((nt + 7) / 8), 256, 0, st>>>(static_cast<const double*>(f->Ppred), W, Pout, varout,
                                                                           Dp, mp, nt);
            k_vecops<<<dim3(tiles, vrows), 256, 0, st>>>(va, W, mp, Dp, tiles);
        } else {
            k_downdate<N><<<dim3(tiles, tiles + vrows), 256, 0, st>>>(f->Ppred, W, Pout, varout, dp, mp, 0, mp / 8, va);
        }
    }
    // K5: read-out + deterministic reduction of the per-row partial sums
    IwpConsts cn = f->iwp;  // the next step of the loop has the same dt: no frame change
    for (int a = 0; a < MAXN; ++a) cn.ts[a] = 1.0;
    RoleArgs ra{cn, dp, mout, f->mpred, f->shift, f->G, f->zbuf, mm, f->flags, f->nflags};
    const unsigned rblocks = (f->d + 255) / 256;
    if (kind == STEP_FULL)
        k_readout<N, false><<<rblocks, 256, 0, st>>>(mout, varout, record ? f->rec_means : nullptr,
                                                    record ? f->rec_stds : nullptr, nordsieck_scale(f->nu, 0, dt), f->d,
                                                    f->part, f->rec, mp, f->ctr, ra, f->tickets, (int)rblocks, f->Ppred, f->rdiag,
                                                    f->Rdense, f->p32);
    else  // + the next step's vector predict (1 block) and the next G (front_block_tiled: Dp/8 + CB(CB+1)/2 blocks)
        k_readout<N, true><<<rblocks + 1 + (unsigned)front_tiled_blocks(Dp, mp), 256, sizeof(double) * Dp, st>>>(
            mout, varout, record ? f->rec_means : nullptr, record ? f->rec_stds : nullptr, nordsieck_scale(f->nu, 0, dt),
            f->d, f->part, f->rec, mp, f->ctr, ra, f->tickets, (int)rblocks, f->Ppred, f->rdiag, f->Rdense, f->p32);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->err = std::string("kernel launch: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int dispatch_step(pnmol_filter* f, const double* Pin, const double* min, double frame_dt, double dt, double* Pout,
                  double* mout, double* varout, bool record, StepKind kind = STEP_FULL) {
    switch (f->n) {
        case 2: return launch_step<2>(f, Pin, min, frame_dt, dt, Pout, mout, varout, record, kind);
        case 3: return launch_step<3>(f, Pin, min, frame_dt, dt, Pout, mout, varout, record, kind);
        case 4: return launch_step<4>(f, Pin, min, frame_dt, dt, Pout, mout, varout, record, kind);
    }
    return -1;
}

void drop_graphs(pnmol_filter* f) {
    for (auto& g : f->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    f->graphs.clear();
}

// A captured run of `nsteps` (even) constant-dt steps that ping-pongs P0 -> P1 -> P0 ...; all per-step
// outputs are addressed through the device step counter, so one executable graph serves any position of
// the sequence.
int get_graph(pnmol_filter* f, double* P0, double* M0, double* P1, double* M1, double* var, double dt, int nsteps,
              hipGraphExec_t* out) {
    pnmol_ctx* ctx = f->ctx;
    const bool have_sq = (f->Sqinv != nullptr && f->sq_dt == dt);
    for (auto& g : f->graphs)
        if (g.P0 == P0 && g.P1 == P1 && g.var == var && g.dt == dt && g.nsteps == nsteps && g.have_sq == have_sq &&
            g.fused == fused_loop(f)) {
            *out = g.exec;
            return 0;
        }
    if (f->graphs.size() >= 8) drop_graphs(f);
    hipGraph_t graph = nullptr;
    HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int it = 0; it < nsteps && rc == 0; ++it)
        rc = (it & 1) ? dispatch_step(f, P1, M1, dt, dt, P0, M0, var, true, fused_loop(f) ? STEP_STEADY : STEP_FULL)
                      : dispatch_step(f, P0, M0, dt, dt, P1, M1, var, true, fused_loop(f) ? STEP_STEADY : STEP_FULL);
    hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    if (rc != 0 || e != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        if (rc == 0) ctx->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e);
        return rc != 0 ? rc : -2;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        ctx->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e);
        return -2;
    }
    f->graphs.push_back({P0, P1, var, dt, nsteps, have_sq, fused_loop(f), exec, false});
    *out = exec;
    return 0;
}

// ELL image of Hv = [-M; B] (rows padded to mp, arrays [e*mp + i]); returns the width
int build_ell(const double* M, const double* B, int d, int ds, int nB, int mp, std::vector<int>& ecol,
              std::vector<double>& eval) {
    const int m = d + nB;
    int w = 1;
    for (int i = 0; i < m; ++i) {
        int c = 0;
        const double* row = i < d ? M + (long)i * ds : B + (long)(i - d) * ds;
        for (int k = 0; k < ds; ++k) c += row[k] != 0.0;
        w = c > w ? c : w;
    }
    ecol.assign((size_t)w * mp, -1);
    eval.assign((size_t)w * mp, 0.0);
    for (int i = 0; i < m; ++i) {
        int e = 0;
        for (int k = 0; k < ds; ++k) {
            const double v = i < d ? -M[(long)i * ds + k] : B[(long)(i - d) * ds + k];
            if (v != 0.0) {
                ecol[(size_t)e * mp + i] = k;  // state index (0, k) = k
                eval[(size_t)e * mp + i] = v;
                ++e;
            }
        }
    }
    return w;
}

// graphs for `k` constant-dt steps from state `s` (host work: capture + instantiate, cached)
int prepare_graphs(pnmol_filter* f, pnmol_state* s, int k, double dt, hipGraphExec_t* gbig, hipGraphExec_t* gpair) {
    *gbig = *gpair = nullptr;
    if (f->graph_chunk < 2) return 0;
    double *curP = s->P, *curM = s->mean, *nxtP = f->tmpP, *nxtM = f->tmpMean;
    // a frame change has its own constants, and in the fused loop the first step does the full predict: that step runs eagerly
    const int lead = (s->frame_dt != dt || fused_loop(f)) ? 1 : 0;
    const int rest = k - lead;
    double *gP0 = lead ? nxtP : curP, *gM0 = lead ? nxtM : curM, *gP1 = lead ? curP : nxtP, *gM1 = lead ? curM : nxtM;
    int rc = 0;
    if (rest >= f->graph_chunk) rc = get_graph(f, gP0, gM0, gP1, gM1, s->var, dt, f->graph_chunk, gbig);
    if (rc == 0 && (rest % f->graph_chunk) >= 2) rc = get_graph(f, gP0, gM0, gP1, gM1, s->var, dt, 2, gpair);
    return rc;
}

// Covariance in the reference's state order (index j n + a, raw coordinates) as a (Dq x Dq) matrix padded with the
// identity: input of the on-device Cholesky behind pnmol_state_get_cov_sqrtm.
__global__ __launch_bounds__(256) void k_cov_reference_order(const void* __restrict__ P, double* __restrict__ Gc, int n,
                                                             int d, int dp, long Dq, const double* __restrict__ sc, int p32) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= Dq * Dq) return;
    const long r = e / Dq, c = e % Dq, D = (long)n * d, Dp = (long)n * dp;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < D && c < D) {
        const int j = (int)(r / n), a = (int)(r % n), k = (int)(c / n), b = (int)(c % n);
        const long idx = ((long)a * dp + j) * Dp + (long)b * dp + k;
        v = sc[a] * sc[b] * (p32 ? (double)static_cast<const float*>(P)[idx] : static_cast<const double*>(P)[idx]);
    }
    Gc[e] = v;
}

// Per-call results go to the host through a KERNEL that writes the mapped pinned staging buffer: one launch instead
// of four hipMemcpyAsync calls (no SDMA queue, no runtime copy path inside the step loop).
__global__ __launch_bounds__(256) void k_copy_out(const double* __restrict__ rec, const double* __restrict__ means,
                                                  const double* __restrict__ stds, const int* __restrict__ info,
                                                  double* __restrict__ out, int k, int d, int cap) {
    const long nm = (long)k * d;
    double* hm = out + (size_t)4 * cap;
    double* hs = hm + (size_t)cap * d;
    int* hi = reinterpret_cast<int*>(hs + (size_t)cap * d);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < nm; e += (long)gridDim.x * 256) {
        if (means) hm[e] = means[e];
        if (stds) hs[e] = stds[e];
    }
    if (blockIdx.x == 0) {
        for (int e = threadIdx.x; e < 4 * k; e += 256) out[e] = rec[e];
        for (int e = threadIdx.x; e < k; e += 256) hi[e] = info[e];
    }
}

// start of a pnmol_filter_step(s) call: info words to "no failure", step counter to 0, last step's counter value
// (a kernel rather than three runtime memsets)
__global__ void k_init_call(int* __restrict__ info, int k, int* __restrict__ ctr, int* __restrict__ last_ctr) {
    for (int e = threadIdx.x; e < k; e += blockDim.x) info[e] = 0x7f7f7f7f;
    if (threadIdx.x == 0) {
        *ctr = 0;
        *last_ctr = k;
    }
}

int ensure_rec(pnmol_filter* f, int k) {
    if (k <= f->rec_cap) return 0;
    if (k < 128) k = 128;
    pnmol_ctx* ctx = f->ctx;
    drop_graphs(f);  // rec/info/read-out pointers are baked into captured graphs
    if (f->rec) hipFree(f->rec);
    if (f->info) hipFree(f->info);
    if (f->rec_means) hipFree(f->rec_means);
    if (f->rec_stds) hipFree(f->rec_stds);
    if (f->h_pin) hipHostFree(f->h_pin);
    f->rec = nullptr, f->info = nullptr, f->rec_means = nullptr, f->rec_stds = nullptr, f->h_pin = nullptr;
    f->rec_cap = 0;
    HIPCHK(ctx, hipHostMalloc(&f->h_pin, sizeof(double) * ((size_t)4 * k + 2 * (size_t)k * f->d + (size_t)k), hipHostMallocMapped));
    HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&f->h_pin_dev), f->h_pin, 0));
    HIPCHK(ctx, hipMalloc(&f->rec, sizeof(double) * 4 * k));
    HIPCHK(ctx, hipMalloc(&f->info, sizeof(int) * k));
    HIPCHK(ctx, hipMalloc(&f->rec_means, sizeof(double) * (size_t)k * f->d));
    HIPCHK(ctx, hipMalloc(&f->rec_stds, sizeof(double) * (size_t)k * f->d));
    f->rec_cap = k;
    return 0;
}

void fill_out(const pnmol_filter* f, const double* rec, int info, double t_new, bool have_sq, pnmol_step_out* o) {
    o->t_new = t_new;
    o->sigma2_whitened = rec[0] / f->m;
    o->diffusion_squared_local = rec[1] / f->m;
    o->error_sigma2 = have_sq ? rec[2] / f->m : std::nan("");
    o->info = (info >= f->mp) ? -1 : info;
}

template <int N>
int run_cov_sqrtm_sweep(pnmol_filter* f, const double* Gc, double* Fc, double* Linvc, int Dq, double* feedc) {
    DowndateArgs dd{};
    const int cb = Dq / NB;
    launch_sweep<N, false>(cb, f->ctx->stream, Gc, Fc, Linvc, Dq, cb, cb, f->flags, f->info_err, f->one, dd,
                           f->flags + 2 * cb + 1, cb <= 17 && sweep_rl_enabled() ? feedc : f->hs_scratch, 1, f->xcd_home);
    return 0;
}

template <int N>
int run_error_model_sweep(pnmol_filter* f, const MeasModel& mm) {
    hipStream_t st = f->ctx->stream;
    const int mp = f->mp;
    const long Dp = f->Dp;
    k_front<<<dim3((mp + 255) / 256, (unsigned)(Dp + mp + (f->sweep_mode == 0 ? mp : 0))), 256, 0, st>>>(f->Qfull, f->G, f->rdiag,
                                                                                                         f->Rdense, mm, Dp, f->p32);
    DowndateArgs dd{};
    launch_sweep<N, false>(f->RT, st, f->G, f->F, f->Linv, mp, f->CB, f->RT, f->flags, f->info_err, f->one, dd,
                           f->flags + f->RT + f->CB + 1, f->hs_scratch, f->p32 ? 2 : 0, f->xcd_home);
    return 0;
}

}  // namespace

extern "C" {

// 2: pnmol_filter_desc.dtype, lifetime rule (refused destroys), pnmol_filter_sweep_layout; 3: pnmol_sqrt_filter_create takes dtype = 1
int pnmol_abi_version(void) { return 3; }

int pnmol_device_count(int* count) {
    if (!count) return -1;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        *count = 0;
        return -2;
    }
    *count = c;
    return 0;
}

int pnmol_ctx_create(int device, pnmol_ctx** out) {
    if (!out) return -1;
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || device < 0 || device >= c) return -2;
    if (hipSetDevice(device) != hipSuccess) return -2;
    pnmol_ctx* ctx = new pnmol_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return -2;
    }
    *out = ctx;
    return 0;
}

int pnmol_ctx_destroy(pnmol_ctx* ctx) {
    if (!ctx) return -1;
    if (ctx->children.load() != 0) {  // (lifetime rule, include/pnmol_hip.h: filters first)
        ctx->err = "pnmol_ctx_destroy: " + std::to_string(ctx->children.load()) + " filter(s) of this ctx are still alive";
        return -1;
    }
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int pnmol_ctx_synchronize(pnmol_ctx* ctx) {
    if (!ctx) return -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

const char* pnmol_last_error(pnmol_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int pnmol_fd_solve_batched(pnmol_ctx* ctx, const double* gram_nss, const double* lk_ns, const double* llk_n, int N, int s,
                           double* weights_ns, double* uncertainty_n) {
    if (!ctx || !gram_nss || !lk_ns || !llk_n || !weights_ns || !uncertainty_n || N < 1 || s < 1 || s > FD_MAXS) {
        if (ctx) ctx->err = "pnmol_fd_solve_batched: bad argument (need N >= 1, 1 <= s <= 16, non-null buffers)";
        return -1;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    double* d = nullptr;
    const size_t nG = (size_t)N * s * s, nV = (size_t)N * s;
    HIPCHK(ctx, hipMalloc(&d, sizeof(double) * (nG + 2 * nV + 2 * (size_t)N)));
    double *dG = d, *dK = dG + nG, *dL = dK + nV, *dW = dL + N, *dU = dW + nV;
    hipError_t e = hipMemcpyAsync(dG, gram_nss, sizeof(double) * nG, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dK, lk_ns, sizeof(double) * nV, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dL, llk_n, sizeof(double) * N, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_fd_solve<<<(unsigned)((N + 127) / 128), 128, 0, st>>>(dG, dK, dL, N, s, dW, dU);
        e = hipMemcpyAsync(weights_ns, dW, sizeof(double) * nV, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(uncertainty_n, dU, sizeof(double) * N, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    (void)hipFree(d);
    if (e != hipSuccess) {
        ctx->err = std::string("pnmol_fd_solve_batched: ") + hipGetErrorString(e);
        return -2;
    }
    return 0;
}

int pnmol_cholesky_lower(pnmol_ctx* ctx, const double* A_nn, int n, double* L_nn) {
    if (!ctx || !A_nn || !L_nn || n < 1) return -1;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int Dq = round_up(n, NB), cb = Dq / NB;
    const size_t nflags = (size_t)std::max(2 * cb + 1 + cb * cb, rl_flags(cb, cb).total);
    const size_t ntile = (size_t)std::max(cb * cb, 2 * cb + 2);
    double *Gc = nullptr, *Fc = nullptr, *Lc = nullptr, *hs = nullptr;
    int* iw = nullptr;  // [flags | info | one]
    std::vector<double> hG((size_t)Dq * Dq, 0.0);
    for (int i = 0; i < n; ++i) std::memcpy(&hG[(size_t)i * Dq], A_nn + (size_t)i * n, sizeof(double) * n);
    for (int i = n; i < Dq; ++i) hG[(size_t)i * Dq + i] = 1.0;
    hipError_t e = hipMalloc(&Gc, sizeof(double) * hG.size());
    if (e == hipSuccess) e = hipMalloc(&Fc, sizeof(double) * hG.size());
    if (e == hipSuccess) e = hipMalloc(&Lc, sizeof(double) * (size_t)cb * NB * NB);
    if (e == hipSuccess) e = hipMalloc(&hs, sizeof(double) * ntile * NB * NB);
    if (e == hipSuccess) e = hipMalloc(&iw, sizeof(int) * (nflags + 2));
    int inf = 0;
    if (e == hipSuccess) {
        const int init[2] = {0x7f7f7f7f, 1};
        e = hipMemcpyAsync(Gc, hG.data(), sizeof(double) * hG.size(), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemsetAsync(Fc, 0, sizeof(double) * hG.size(), st);
        if (e == hipSuccess) e = hipMemsetAsync(Lc, 0, sizeof(double) * (size_t)cb * NB * NB, st);
        if (e == hipSuccess) e = hipMemsetAsync(iw, 0, sizeof(int) * nflags, st);
        if (e == hipSuccess) e = hipMemcpyAsync(iw + nflags, init, sizeof(init), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            DowndateArgs dd{};
            // (strict pivots: a Gram matrix that is not numerically positive definite is an error, as in LAPACK's potrf)
            launch_sweep<2, false>((unsigned)cb, st, Gc, Fc, Lc, Dq, cb, cb, iw, iw + nflags, iw + nflags + 1, dd, iw + 2 * cb + 1, hs, 0, -1);
            e = hipMemcpyAsync(hG.data(), Fc, sizeof(double) * hG.size(), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&inf, iw + nflags, sizeof(int), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipGetLastError();
    }
    for (void* q : {(void*)Gc, (void*)Fc, (void*)Lc, (void*)hs, (void*)iw})
        if (q) (void)hipFree(q);
    if (e != hipSuccess) {
        ctx->err = std::string("pnmol_cholesky_lower: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? -4 : -2;
    }
    if (inf == -2) {
        ctx->err = "pnmol_cholesky_lower: a dependency wait timed out";
        return -2;
    }
    if (inf < Dq) {
        ctx->err = "pnmol_cholesky_lower: matrix not positive definite at pivot " + std::to_string(inf);
        return -3;
    }
    for (int i = 0; i < n; ++i) {
        std::memcpy(L_nn + (size_t)i * n, &hG[(size_t)i * Dq], sizeof(double) * (i + 1));
        for (int k = i + 1; k < n; ++k) L_nn[(size_t)i * n + k] = 0.0;
    }
    return 0;
}

int pnmol_filter_create(pnmol_ctx* ctx, const pnmol_filter_desc* desc, pnmol_filter** out) {
    if (!ctx || !desc || !out) return -1;
    *out = nullptr;
    const int d = desc->d, nu = desc->num_derivatives, nB = desc->nB, n = nu + 1;
    const int ds = desc->d_state > 0 ? desc->d_state : d;
    if (d < 1 || ds < d || nu < 1 || n > MAXN || nB < 0 || !desc->L || !desc->E_sqrtm || !desc->Gamma ||
        (nB > 0 && (!desc->B || !desc->R_sqrtm))) {
        ctx->err = "pnmol_filter_create: bad descriptor (need d>=1, 1<=nu<=3, non-null L/B/E_sqrtm/R_sqrtm/Gamma)";
        return -1;
    }
    if (desc->dtype != 0 && desc->dtype != 1) {
        ctx->err = "pnmol_filter_create: unknown dtype " + std::to_string(desc->dtype) + " (0 = fp64, 1 = fp32 covariance)";
        return -1;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pnmol_filter* f = new pnmol_filter();
    f->ctx = ctx;
    ctx->children.fetch_add(1);
    f->registered = true;
    f->d = d, f->ds = ds, f->nu = nu, f->n = n, f->nB = nB, f->m = d + nB;
    f->p32 = desc->dtype == 1, f->psz = f->p32 ? 4 : 8;
    f->dp = round_up(ds, NB), f->mp = round_up(f->m, NB);
    f->Dp = (long)n * f->dp;
    f->CB = f->mp / NB, f->RBS = f->mp / NB, f->RBW = (int)(f->Dp / NB), f->RT = f->RBS + f->RBW + 1 + f->RBS;  // [S; W; z-block; I]
    const int dp = f->dp, mp = f->mp, m = f->m;
    const long Dp = f->Dp;

    // IWP constants, base/iwp.py:13-30: A1 = flip(pascal lower), Q1 = flip(hilbert)
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            const int ia = n - 1 - a, ib = n - 1 - b;  // flip both axes
            double binom = 0.0;
            if (ib <= ia) {
                binom = 1.0;
                for (int q = 1; q <= ib; ++q) binom = binom * (ia - ib + q) / q;
            }
            f->iwp.A1[a * MAXN + b] = binom;
            f->iwp.Q1[a * MAXN + b] = 1.0 / (ia + ib + 1.0);
        }

    // Hv = [-L; B] in ELL, rows padded to mp
    std::vector<int> ecol;
    std::vector<double> eval;
    f->hB.assign(desc->B ? desc->B : nullptr, desc->B ? desc->B + (size_t)nB * ds : nullptr);
    const int w = build_ell(desc->L, f->hB.data(), d, ds, nB, mp, ecol, eval);
    f->ellw = w;
    f->ell_cap = w;
    // R = blockdiag(E E^T, Rb Rb^T): diagonal fast path, dense otherwise
    bool diag = true;
    for (int i = 0; i < d && diag; ++i)
        for (int k = 0; k < d; ++k)
            if (i != k && desc->E_sqrtm[(long)i * d + k] != 0.0) {
                diag = false;
                break;
            }
    for (int i = 0; i < nB && diag; ++i)
        for (int k = 0; k < nB; ++k)
            if (i != k && desc->R_sqrtm[(long)i * nB + k] != 0.0) {
                diag = false;
                break;
            }
    std::vector<double> rdiag(mp, 0.0), Rd;
    if (diag) {
        for (int i = 0; i < d; ++i) rdiag[i] = desc->E_sqrtm[(long)i * d + i] * desc->E_sqrtm[(long)i * d + i];
        for (int i = 0; i < nB; ++i) rdiag[d + i] = desc->R_sqrtm[(long)i * nB + i] * desc->R_sqrtm[(long)i * nB + i];
    } else {
        Rd.assign((size_t)mp * mp, 0.0);
        for (int i = 0; i < d; ++i)
            for (int k = 0; k <= i; ++k) {
                double s = 0.0;
                for (int q = 0; q < d; ++q) s += desc->E_sqrtm[(long)i * d + q] * desc->E_sqrtm[(long)k * d + q];
                Rd[(size_t)i * mp + k] = Rd[(size_t)k * mp + i] = s;
            }
        for (int i = 0; i < nB; ++i)
            for (int k = 0; k <= i; ++k) {
                double s = 0.0;
                for (int q = 0; q < nB; ++q) s += desc->R_sqrtm[(long)i * nB + q] * desc->R_sqrtm[(long)k * nB + q];
                Rd[(size_t)(d + i) * mp + d + k] = Rd[(size_t)(d + k) * mp + d + i] = s;
            }
    }
    // K = Gamma Gamma^T (white.py:84-85, base/iwp.py:49-52), padded
    std::vector<double> Kg((size_t)dp * dp, 0.0);
    if (desc->K) {  // the caller's Gram matrix (symmetrised: both triangles from its lower one, like the loop below)
        for (int i = 0; i < ds; ++i)
            for (int k = 0; k <= i; ++k) Kg[(size_t)i * dp + k] = Kg[(size_t)k * dp + i] = desc->K[(size_t)i * ds + k];
    } else
    for (int i = 0; i < ds; ++i)
        for (int k = 0; k <= i; ++k) {
            double s = 0.0;
            for (int q = 0; q <= k; ++q) s += desc->Gamma[(long)i * ds + q] * desc->Gamma[(long)k * ds + q];
            Kg[(size_t)i * dp + k] = Kg[(size_t)k * dp + i] = s;
        }

    auto fail = [&](int code) {
        pnmol_filter_destroy(f);
        return code;
    };
#define FCHK(call)                                                                    \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess) {                                                      \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);            \
            return fail(e__ == hipErrorOutOfMemory ? -4 : -2);                        \
        }                                                                             \
    } while (0)
    const size_t tall = (size_t)f->RT * NB * mp;
    FCHK(hipMalloc(&f->ell_col, sizeof(int) * ecol.size()));
    FCHK(hipMalloc(&f->ell_val, sizeof(double) * eval.size()));
    FCHK(hipMalloc(&f->Kg, sizeof(double) * Kg.size()));
    FCHK(hipMalloc(&f->rdiag, sizeof(double) * mp));
    FCHK(hipMalloc(&f->shift, sizeof(double) * mp));
    FCHK(hipMalloc(&f->G, sizeof(double) * tall));
    FCHK(hipMalloc(&f->F, sizeof(double) * tall));
    FCHK(hipMalloc(&f->Linv, sizeof(double) * (size_t)f->CB * NB * NB));
    FCHK(hipMemset(f->Linv, 0, sizeof(double) * (size_t)f->CB * NB * NB));  // (upper-right quadrants stay zero: diag4_factor)
    FCHK(hipMalloc(&f->Ppred, f->psz * (size_t)Dp * Dp));
    FCHK(hipMalloc(&f->tmpP, f->psz * (size_t)Dp * Dp));
    FCHK(hipMalloc(&f->mpred, sizeof(double) * Dp));
    FCHK(hipMalloc(&f->tmpMean, sizeof(double) * Dp));
    FCHK(hipMalloc(&f->var, sizeof(double) * Dp));
    FCHK(hipMalloc(&f->zbuf, sizeof(double) * mp));
    FCHK(hipMalloc(&f->part, sizeof(double) * 3 * mp));
    FCHK(hipMalloc(&f->ctr, sizeof(int)));
    if (const char* gc = std::getenv("PNMOL_HIP_GRAPH_CHUNK")) f->graph_chunk = std::atoi(gc) / 2 * 2;
    if (const char* e = std::getenv("PNMOL_HIP_SWEEP")) f->sweep_mode = std::atoi(e);
    if (f->sweep_mode == 2 && n > 3) f->sweep_mode = 1;  // the fused down-date role is built for n <= 3
    {
        // Large fp64 problems (from D = 4096 with more column blocks than the register-resident sweep holds): the sweep
        // alone, then the SYRK of k_downdate_big -- measured on the 64x64 mesh against the fused launch (see k_downdate_big)
        const char* e = std::getenv("PNMOL_HIP_DD_BIG");
        const bool fits = f->Dp % BDT == 0 && f->mp % BDK == 0 && f->sweep_mode >= 1 && (!f->p32 || f->sweep_mode == 2);
        const bool want = e ? std::atoi(e) != 0 : (f->Dp >= 8192 && f->CB > 17);
        if (fits && want) {
            f->dd_big = 1;
            f->sweep_mode = 1;
            const char* g = std::getenv("PNMOL_HIP_W_GEMM");  // (0: the rows of W stay in the sweep; A/B switch)
            f->w_gemm = (g ? std::atoi(g) != 0 : 1) && f->mp % BDT == 0 && f->CB > 17;  // (k_sweep, not k_sweep_rl)
        }
    }
    {
        // XCD-local layout (k_sweep_rl<.., XL>): only a filter that launches k_sweep_rl is counted, per device; unless
        // PNMOL_HIP_SWEEP_XL forces it, the first such filter alive on its device gets the layout (pnmol_filter_sweep_layout
        // reports what was chosen, so that timings are attributable).
        static std::atomic<int> next_home[MAX_DEVICES];
        const bool rl = sweep_rl_enabled() && f->CB <= 17 && f->sweep_mode >= 1;
        if (rl) {
            const int dv = ctx->device % MAX_DEVICES;
            const int others = live_rl_filters[dv].fetch_add(1);
            f->counted = true;
            const bool xl = sweep_xl_mode() < 0 ? others == 0 : sweep_xl_mode() == 1;
            f->xcd_home = xl ? next_home[dv].fetch_add(1) % 8 : -1;
        }
    }
    FCHK(hipMalloc(&f->sdiag, sizeof(double) * (mp + 1)));
    // k_sweep: row[RT], diag[CB], abort, claim[CB*CB];  k_sweep_rl: rl_flags();  the Cholesky factor of a whole
    // covariance (pnmol_state_get_cov_sqrtm) runs a square sweep of up to Dp/32 blocks on the same words
    f->nflags = std::max({f->RT + f->CB + 1 + f->CB * f->CB, rl_flags(f->RT, f->CB).total,
                          rl_flags((int)(f->Dp / NB) + 1, (int)(f->Dp / NB) + 1).total});
    FCHK(hipMalloc(&f->flags, sizeof(int) * f->nflags));
    FCHK(hipMemset(f->flags, 0, sizeof(int) * f->nflags));
    FCHK(hipMalloc(&f->hs_scratch, sizeof(double) * (size_t)std::max(f->CB * f->CB, 2 * f->CB + 2) * NB * NB));  // (also k_sweep_rl's feed tiles)
    if (const char* e = std::getenv("PNMOL_HIP_FUSE_PREDICT")) f->fuse_predict = std::atoi(e);
    FCHK(hipMalloc(&f->last_ctr, sizeof(int)));
    FCHK(hipMemset(f->last_ctr, 0, sizeof(int)));
    FCHK(hipMalloc(&f->tickets, sizeof(int)));
    FCHK(hipMemset(f->tickets, 0, sizeof(int)));
    if (f->p32 && f->sweep_mode != 2 && !f->dd_big) {
        ctx->err = "pnmol_filter_create: dtype = fp32 covariance needs the fused sweep (num_derivatives <= 2, PNMOL_HIP_SWEEP unset)";
        return fail(-1);
    }
    FCHK(hipMemcpy(f->ell_col, ecol.data(), sizeof(int) * ecol.size(), hipMemcpyHostToDevice));
    FCHK(hipMemcpy(f->ell_val, eval.data(), sizeof(double) * eval.size(), hipMemcpyHostToDevice));
    {   // base operator for pnmol_filter_set_operator_diagonal
        std::vector<int> slot((size_t)mp, 0);
        f->base_has_diag = 1;
        for (int i = 0; i < d; ++i) {
            int e = 0;
            while (e < w && ecol[(size_t)e * mp + i] != i) ++e;
            if (e == w) f->base_has_diag = 0, e = 0;
            slot[i] = e;
        }
        f->base_w = w;
        FCHK(hipMalloc(&f->ell_col_base, sizeof(int) * ecol.size()));
        FCHK(hipMalloc(&f->ell_val_base, sizeof(double) * eval.size()));
        FCHK(hipMalloc(&f->ell_diag_slot, sizeof(int) * mp));
        FCHK(hipMemcpy(f->ell_col_base, ecol.data(), sizeof(int) * ecol.size(), hipMemcpyHostToDevice));
        FCHK(hipMemcpy(f->ell_val_base, eval.data(), sizeof(double) * eval.size(), hipMemcpyHostToDevice));
        FCHK(hipMemcpy(f->ell_diag_slot, slot.data(), sizeof(int) * mp, hipMemcpyHostToDevice));
        FCHK(hipHostMalloc(&f->h_op, sizeof(double) * (size_t)(d + mp), hipHostMallocMapped));
        FCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&f->h_op_dev), f->h_op, 0));
        FCHK(hipEventCreate(&f->ev_op));
    }
    FCHK(hipMemcpy(f->Kg, Kg.data(), sizeof(double) * Kg.size(), hipMemcpyHostToDevice));
    FCHK(hipMemcpy(f->rdiag, rdiag.data(), sizeof(double) * mp, hipMemcpyHostToDevice));
    if (!diag) {
        FCHK(hipMalloc(&f->Rdense, sizeof(double) * Rd.size()));
        FCHK(hipMemcpy(f->Rdense, Rd.data(), sizeof(double) * Rd.size(), hipMemcpyHostToDevice));
    }
    FCHK(hipMemset(f->shift, 0, sizeof(double) * mp));
    FCHK(hipMemset(f->G, 0, sizeof(double) * tall));
    k_set_identity<<<(mp + 255) / 256, 256>>>(f->G + ((size_t)mp + Dp + NB) * mp, mp);
    FCHK(hipDeviceSynchronize());
    FCHK(hipMemset(f->F, 0, sizeof(double) * tall));
    FCHK(hipMemset(f->var, 0, sizeof(double) * Dp));
    FCHK(hipMemset(f->tmpP, 0, f->psz * (size_t)Dp * Dp));
    FCHK(hipMemset(f->tmpMean, 0, sizeof(double) * Dp));
    FCHK(hipEventCreate(&f->ev0));
    FCHK(hipEventCreate(&f->ev1));
#undef FCHK
    if (ensure_rec(f, 128) != 0) return fail(-2);
    *out = f;
    return 0;
}

int pnmol_filter_destroy(pnmol_filter* f) {
    if (!f) return -1;
    if (f->states.load() != 0) {  // (lifetime rule, include/pnmol_hip.h: states first)
        f->ctx->err = "pnmol_filter_destroy: " + std::to_string(f->states.load()) + " state(s) of this filter are still alive";
        return -1;
    }
    if (f->counted) live_rl_filters[f->ctx->device % MAX_DEVICES].fetch_sub(1);
    if (f->registered) f->ctx->children.fetch_sub(1);
    hipSetDevice(f->ctx->device);
    drop_graphs(f);
    if (f->ctr) hipFree(f->ctr);
    if (f->h_pin) hipHostFree(f->h_pin);
    void* ptrs[] = {f->ell_col, f->ell_val, f->Kg,   f->rdiag,   f->Rdense, f->shift,     f->G,        f->F,
                    f->Linv,    f->Ppred,   f->mpred, f->zbuf,   f->var,    f->Sqinv,     f->rec,      f->part,  f->sdiag,
                    f->info,    f->tmpP,    f->tmpMean, f->rec_means, f->rec_stds, f->flags, f->Qfull, f->one, f->info_err, f->hs_scratch, f->last_ctr, f->tickets};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (f->h_op) hipHostFree(f->h_op);
    if (f->ev_op) hipEventDestroy(f->ev_op);
    for (void* q : {(void*)f->ell_col_base, (void*)f->ell_val_base, (void*)f->ell_diag_slot})
        if (q) hipFree(q);
    if (f->ev0) hipEventDestroy(f->ev0);
    if (f->ev1) hipEventDestroy(f->ev1);
    delete f;
    return 0;
}

int pnmol_filter_dims(const pnmol_filter* f, int* d, int* n, int* m, int* dp, int* mp) {
    if (!f) return -1;
    if (d) *d = f->d;
    if (n) *n = f->n;
    if (m) *m = f->m;
    if (dp) *dp = f->dp;
    if (mp) *mp = f->mp;
    return 0;
}

int pnmol_filter_sweep_layout(const pnmol_filter* f, int* kernel, int* xcd_home) {
    if (!f) return -1;
    if (kernel) *kernel = f->sweep_mode == 0 ? 0 : (sweep_rl_enabled() && f->CB <= 17 ? 2 : 1);
    if (xcd_home) *xcd_home = f->xcd_home;
    return 0;
}

int pnmol_filter_set_error_model(pnmol_filter* f, double dt, const double* Sq_inv, const double* Sq_diag) {
    if (!f || !Sq_inv || !Sq_diag || !(dt > 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int m = f->m, mp = f->mp;
    std::vector<double> pad((size_t)mp * mp, 0.0);
    for (int i = 0; i < m; ++i) std::memcpy(&pad[(size_t)i * mp], Sq_inv + (size_t)i * m, sizeof(double) * m);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (!f->Sqinv) HIPCHK(ctx, hipMalloc(&f->Sqinv, sizeof(double) * pad.size()));
    HIPCHK(ctx, hipMemcpy(f->Sqinv, pad.data(), sizeof(double) * pad.size(), hipMemcpyHostToDevice));
    f->sqdiag.assign(Sq_diag, Sq_diag + m);
    f->sq_dt = dt;
    drop_graphs(f);
    return 0;
}

int pnmol_filter_prepare_error_model(pnmol_filter* f, double dt) {
    if (!f || !(dt > 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int mp = f->mp, m = f->m, dp = f->dp;
    const long Dp = f->Dp;
    if (!f->one) {  // (pnmol_state_get_cov_sqrtm may have made them already: found by the sanitizer run, tests/asan)
        HIPCHK(ctx, hipMalloc(&f->one, sizeof(int)));
        HIPCHK(ctx, hipMalloc(&f->info_err, sizeof(int)));
        const int h1 = 1;
        HIPCHK(ctx, hipMemcpy(f->one, &h1, sizeof(int), hipMemcpyHostToDevice));
    }
    if (!f->Qfull) {
        HIPCHK(ctx, hipMalloc(&f->Qfull, f->psz * (size_t)Dp * Dp));
        k_fill_q<<<(unsigned)((Dp * Dp + 255) / 256), 256, 0, st>>>(f->Qfull, f->Kg, f->iwp, f->n, dp, f->p32);
    }
    const bool fresh = (f->Sqinv == nullptr);
    if (fresh) HIPCHK(ctx, hipMalloc(&f->Sqinv, sizeof(double) * (size_t)mp * mp));
    HIPCHK(ctx, hipMemsetAsync(f->flags, 0, sizeof(int) * f->nflags, st));
    HIPCHK(ctx, hipMemsetAsync(f->info_err, 0x7f, sizeof(int), st));
    MeasModel mm{f->ell_col, f->ell_val, f->ellw, f->d, f->m, f->dp, f->mp,
                 nordsieck_scale(f->nu, 0, dt), nordsieck_scale(f->nu, 1, dt)};
    switch (f->n) {
        case 2: run_error_model_sweep<2>(f, mm); break;
        case 3: run_error_model_sweep<3>(f, mm); break;
        case 4: run_error_model_sweep<4>(f, mm); break;
        default: return -1;
    }
    // Sq^-1 = Lq^-T Lq^-1 = T T^T with T = rows of the identity block after the sweep
    const double* T = f->F + ((long)mp + Dp + NB) * mp;
    k_ttt<<<dim3(mp / 16, mp / 16), 256, 0, st>>>(T, f->Sqinv, mp);
    std::vector<double> hd((size_t)m);
    int inf = 0;
    HIPCHK(ctx, hipMemcpy2DAsync(hd.data(), sizeof(double), f->G, sizeof(double) * (mp + 1), sizeof(double), m,
                                 hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(&inf, f->info_err, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->err = std::string("prepare_error_model: ") + hipGetErrorString(e);
        return -2;
    }
    if (inf < mp) {
        ctx->err = inf == -2 ? "prepare_error_model: a dependency wait timed out"
                             : "prepare_error_model: Sq not positive definite at pivot " + std::to_string(inf);
        return inf == -2 ? -2 : -3;
    }
    f->sqdiag = hd;
    f->sq_dt = dt;
    if (fresh) drop_graphs(f);
    return 0;
}

int pnmol_filter_set_operator(pnmol_filter* f, const double* M_dd, const double* shift_d) {
    if (!f || !M_dd) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int> ecol;
    std::vector<double> eval;
    const int w = build_ell(M_dd, f->hB.data(), f->d, f->ds, f->nB, f->mp, ecol, eval);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // (pageable host buffers: the copies below are synchronous anyway)
    if (w != f->ellw || w > f->ell_cap) drop_graphs(f);  // the stencil width / pointers are baked into captured launches
    if (w > f->ell_cap) {
        if (f->ell_col) (void)hipFree(f->ell_col);
        if (f->ell_val) (void)hipFree(f->ell_val);
        f->ell_col = nullptr, f->ell_val = nullptr, f->ell_cap = 0;
        HIPCHK(ctx, hipMalloc(&f->ell_col, sizeof(int) * ecol.size()));
        HIPCHK(ctx, hipMalloc(&f->ell_val, sizeof(double) * eval.size()));
        f->ell_cap = w;
    }
    f->ellw = w;
    HIPCHK(ctx, hipMemcpy(f->ell_col, ecol.data(), sizeof(int) * ecol.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(f->ell_val, eval.data(), sizeof(double) * eval.size(), hipMemcpyHostToDevice));
    std::vector<double> sh((size_t)f->mp, 0.0);
    if (shift_d) std::memcpy(sh.data(), shift_d, sizeof(double) * f->d);
    HIPCHK(ctx, hipMemcpy(f->shift, sh.data(), sizeof(double) * sh.size(), hipMemcpyHostToDevice));
    return 0;
}

int pnmol_filter_set_operator_diagonal(pnmol_filter* f, const double* jdiag_d, const double* shift_d) {
    if (!f || !jdiag_d) return -1;
    pnmol_ctx* ctx = f->ctx;
    if (!f->base_has_diag) {
        ctx->err = "pnmol_filter_set_operator_diagonal: a row of the operator given at creation has no diagonal entry";
        return -1;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // the staging buffer may still be read by the previous call's kernel: wait for THAT kernel, not for the stream
    HIPCHK(ctx, hipEventSynchronize(f->ev_op));
    const int d = f->d, mp = f->mp;
    std::memcpy(f->h_op, jdiag_d, sizeof(double) * d);
    for (int i = 0; i < mp; ++i) f->h_op[d + i] = (shift_d && i < d) ? shift_d[i] : 0.0;
    if (f->ellw != f->base_w) {  // a dense pnmol_filter_set_operator came in between: back to the base image
        drop_graphs(f);
        HIPCHK(ctx, hipMemcpyAsync(f->ell_col, f->ell_col_base, sizeof(int) * (size_t)f->base_w * mp, hipMemcpyDeviceToDevice, st));
        HIPCHK(ctx, hipMemcpyAsync(f->ell_val, f->ell_val_base, sizeof(double) * (size_t)f->base_w * mp, hipMemcpyDeviceToDevice, st));
        f->ellw = f->base_w;
    }
    k_operator_diagonal<<<(unsigned)((mp + 255) / 256), 256, 0, st>>>(f->ell_val, f->ell_val_base, f->ell_diag_slot, f->h_op_dev,
                                                                       f->shift, d, mp);
    HIPCHK(ctx, hipEventRecord(f->ev_op, st));
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int pnmol_filter_predict_mean(pnmol_filter* f, const pnmol_state* in, double dt, double* m_at_d) {
    if (!f || !in || in->f != f || !m_at_d || !(dt > 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<double> hm((size_t)f->Dp);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(hm.data(), in->mean, sizeof(double) * hm.size(), hipMemcpyDeviceToHost));
    // m^- = A m in the Nordsieck frame of dt (white.py:104-107); row 0 of A1, then back to raw coordinates
    const double s0 = nordsieck_scale(f->nu, 0, dt);
    for (int j = 0; j < f->d; ++j) {
        double acc = 0.0;
        for (int a = 0; a < f->n; ++a) {
            const double so = in->frame_dt == 0.0 ? 1.0 : nordsieck_scale(f->nu, a, in->frame_dt);
            acc += f->iwp.A1[a] * (so / nordsieck_scale(f->nu, a, dt)) * hm[(size_t)a * f->dp + j];
        }
        m_at_d[j] = s0 * acc;
    }
    return 0;
}

int pnmol_state_create(pnmol_filter* f, pnmol_state** out) {
    if (!f || !out) return -1;
    *out = nullptr;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    pnmol_state* s = new pnmol_state();
    s->f = f;
    f->states.fetch_add(1);
    const size_t Dp = (size_t)f->Dp;
    if (hipMalloc(&s->mean, sizeof(double) * Dp) != hipSuccess || hipMalloc(&s->var, sizeof(double) * Dp) != hipSuccess ||
        hipMalloc(&s->P, f->psz * Dp * Dp) != hipSuccess) {
        ctx->err = "pnmol_state_create: hipMalloc failed";
        pnmol_state_destroy(s);
        return -4;
    }
    hipMemset(s->mean, 0, sizeof(double) * Dp);
    hipMemset(s->var, 0, sizeof(double) * Dp);
    hipMemset(s->P, 0, f->psz * Dp * Dp);
    *out = s;
    return 0;
}

int pnmol_state_destroy(pnmol_state* s) {
    if (!s) return -1;
    if (s->f->pending_state == s) {
        s->f->ctx->err = "pnmol_state_destroy: this state is the target of an unfinished pnmol_filter_steps_begin";
        return -1;
    }
    s->f->states.fetch_sub(1);
    hipSetDevice(s->f->ctx->device);
    if (s->mean) hipFree(s->mean);
    if (s->var) hipFree(s->var);
    if (s->P) hipFree(s->P);
    delete s;
    return 0;
}

int pnmol_state_clone(const pnmol_state* s, pnmol_state** out) {
    if (!s || !out) return -1;
    int rc = pnmol_state_create(s->f, out);
    if (rc != 0) return rc;
    pnmol_ctx* ctx = s->f->ctx;
    pnmol_state* o = *out;
    const size_t Dp = (size_t)s->f->Dp;
    HIPCHK(ctx, hipMemcpyAsync(o->mean, s->mean, sizeof(double) * Dp, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(o->var, s->var, sizeof(double) * Dp, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(o->P, s->P, s->f->psz * Dp * Dp, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    o->t = s->t, o->frame_dt = s->frame_dt;
    return 0;
}

int pnmol_state_set(pnmol_state* s, double t, const double* mean_nd, const double* cov_DD) {
    if (!s || !mean_nd || !cov_DD) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = f->n, d = f->ds, dp = f->dp;
    const long D = (long)n * d, Dp = f->Dp;
    std::vector<double> hm((size_t)Dp, 0.0), hv((size_t)Dp, 0.0), hP((size_t)Dp * Dp, 0.0);
    for (int a = 0; a < n; ++a)
        for (int j = 0; j < d; ++j) {
            hm[(size_t)a * dp + j] = mean_nd[(size_t)a * d + j];
            hv[(size_t)a * dp + j] = cov_DD[((size_t)j * n + a) * D + (size_t)j * n + a];
        }
    for (int a = 0; a < n; ++a)
        for (int j = 0; j < d; ++j) {
            const double* src = cov_DD + ((size_t)j * n + a) * D;
            double* dst = &hP[((size_t)a * dp + j) * Dp];
            for (int b = 0; b < n; ++b)
                for (int k = 0; k < d; ++k) dst[(size_t)b * dp + k] = src[(size_t)k * n + b];
        }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(s->mean, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(s->var, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice));
    if (f->p32) {
        std::vector<float> hPf(hP.begin(), hP.end());
        HIPCHK(ctx, hipMemcpy(s->P, hPf.data(), sizeof(float) * hPf.size(), hipMemcpyHostToDevice));
    } else {
        HIPCHK(ctx, hipMemcpy(s->P, hP.data(), sizeof(double) * hP.size(), hipMemcpyHostToDevice));
    }
    s->t = t;
    s->frame_dt = 0.0;
    return 0;
}

namespace {
// P = C C^T written straight into the state's derivative-major padded layout, C (D x D, any square root) in the
// reference's F-flattened order: P[(a dp + j)(Dp) + (b dp + k)] = sum_c C[j n + a][c] C[k n + b][c].  One 32x32 tile of
// points (j, k) for one derivative pair (a, b) per block, on the MFMA; also the marginal variances.
__global__ __launch_bounds__(256) void k_cct_layout(void* __restrict__ P, double* __restrict__ var,
                                                    const double* __restrict__ C, int d, int n, int dp, long Dp, int p32) {
    __shared__ double sA[32][33], sB[32][33];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4, wr = w >> 1, wc = w & 1;
    const int tx = t & 31, ty = t >> 5;
    const int a = blockIdx.z / n, b = blockIdx.z % n, j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const long D = (long)n * d;
    d4 acc = {0, 0, 0, 0};
    for (long c0 = 0; c0 < D; c0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            const int j = j0 + r, k = k0 + r;
            const long c = c0 + tx;
            sA[r][tx] = (j < d && c < D) ? C[((long)j * n + a) * D + c] : 0.0;
            sB[r][tx] = (k < d && c < D) ? C[((long)k * n + b) * D + c] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 8; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[wr * 16 + fr][4 * st + fk], sB[wc * 16 + fr][4 * st + fk], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + wr * 16 + fk + 4 * r, k = k0 + wc * 16 + fr;
        if (j < d && k < d) {
            const long idx = ((long)a * dp + j) * Dp + (long)b * dp + k;
            if (p32) static_cast<float*>(P)[idx] = (float)acc[r];
            else static_cast<double*>(P)[idx] = acc[r];
            if (a == b && j == k) var[(long)a * dp + j] = acc[r];
        }
    }
}
}  // namespace

int pnmol_state_set_sqrtm(pnmol_state* s, double t, const double* mean_nd, const double* cov_sqrtm_DD) {
    if (!s || !mean_nd || !cov_sqrtm_DD) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = f->n, d = f->ds, dp = f->dp;
    const long D = (long)n * d, Dp = f->Dp;
    std::vector<double> hm((size_t)Dp, 0.0);
    for (int a = 0; a < n; ++a)
        for (int j = 0; j < d; ++j) hm[(size_t)a * dp + j] = mean_nd[(size_t)a * d + j];
    double* dC = nullptr;
    if (hipMalloc(&dC, sizeof(double) * (size_t)D * D) != hipSuccess) return -4;
    int rc = 0;
    do {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = -2; break; }
        if (hipMemcpy(s->mean, hm.data(), sizeof(double) * hm.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = -2; break; }
        if (hipMemcpy(dC, cov_sqrtm_DD, sizeof(double) * (size_t)D * D, hipMemcpyHostToDevice) != hipSuccess) { rc = -2; break; }
        if (hipMemsetAsync(s->P, 0, f->psz * (size_t)Dp * Dp, ctx->stream) != hipSuccess) { rc = -2; break; }
        if (hipMemsetAsync(s->var, 0, sizeof(double) * (size_t)Dp, ctx->stream) != hipSuccess) { rc = -2; break; }
        hipLaunchKernelGGL(k_cct_layout, dim3((d + 31) / 32, (d + 31) / 32, n * n), dim3(256), 0, ctx->stream, (void*)s->P, s->var,
                           dC, d, n, dp, Dp, f->p32);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = -2; break; }
    } while (0);
    hipFree(dC);
    if (rc) {
        ctx->err = std::string("pnmol_state_set_sqrtm: ") + hipGetErrorString(hipGetLastError());
        return rc;
    }
    s->t = t;
    s->frame_dt = 0.0;
    return 0;
}

int pnmol_state_get_time(const pnmol_state* s, double* t) {
    if (!s || !t) return -1;
    *t = s->t;
    return 0;
}

static void frame_scales(const pnmol_state* s, double* sc) {
    for (int a = 0; a < s->f->n; ++a) sc[a] = s->frame_dt == 0.0 ? 1.0 : nordsieck_scale(s->f->nu, a, s->frame_dt);
}

int pnmol_state_get_mean(const pnmol_state* s, double* mean_nd) {
    if (!s || !mean_nd) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<double> hm((size_t)f->Dp);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(hm.data(), s->mean, sizeof(double) * hm.size(), hipMemcpyDeviceToHost));
    double sc[MAXN];
    frame_scales(s, sc);
    for (int a = 0; a < f->n; ++a)
        for (int j = 0; j < f->ds; ++j) mean_nd[(size_t)a * f->ds + j] = sc[a] * hm[(size_t)a * f->dp + j];
    return 0;
}

int pnmol_state_get_cov_sqrtm(const pnmol_state* s, double* C_DD) {
    if (!s || !C_DD) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int n = f->n, d = f->ds;
    const long D = (long)n * d;
    const int Dq = round_up((int)D, NB), cb = Dq / NB;
    if (std::max(2 * cb + 1, rl_flags(cb, cb).total) > f->nflags) {
        ctx->err = "pnmol_state_get_cov_sqrtm: flag buffer too small for this shape";
        return -1;
    }
    double *Gc = nullptr, *Fc = nullptr, *Lc = nullptr, *dsc = nullptr, *feedc = nullptr;
    hipError_t e = hipMalloc(&Gc, sizeof(double) * (size_t)Dq * Dq);
    if (e == hipSuccess) e = hipMalloc(&feedc, sizeof(double) * (size_t)(2 * cb + 2) * NB * NB);
    if (e == hipSuccess) e = hipMalloc(&Fc, sizeof(double) * (size_t)Dq * Dq);
    if (e == hipSuccess) e = hipMalloc(&Lc, sizeof(double) * (size_t)cb * NB * NB);
    if (e == hipSuccess) e = hipMalloc(&dsc, sizeof(double) * MAXN);
    if (e == hipSuccess && !f->one) {
        e = hipMalloc(&f->one, sizeof(int));
        if (e == hipSuccess) e = hipMalloc(&f->info_err, sizeof(int));
        const int h1 = 1;
        if (e == hipSuccess) e = hipMemcpy(f->one, &h1, sizeof(int), hipMemcpyHostToDevice);
    }
    int rc = 0, inf = 0;
    std::vector<double> hF;
    if (e == hipSuccess) {
        double sc[MAXN];
        frame_scales(s, sc);
        e = hipMemcpy(dsc, sc, sizeof(double) * MAXN, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemsetAsync(Fc, 0, sizeof(double) * (size_t)Dq * Dq, st);
        if (e == hipSuccess) e = hipMemsetAsync(Lc, 0, sizeof(double) * (size_t)cb * NB * NB, st);
        if (e == hipSuccess) e = hipMemsetAsync(f->flags, 0, sizeof(int) * f->nflags, st);
        if (e == hipSuccess) e = hipMemsetAsync(f->info_err, 0x7f, sizeof(int), st);
        if (e == hipSuccess) {
            k_cov_reference_order<<<(unsigned)(((long)Dq * Dq + 255) / 256), 256, 0, st>>>(s->P, Gc, n, d, f->dp, Dq, dsc, f->p32);
            switch (n) {
                case 2: run_cov_sqrtm_sweep<2>(f, Gc, Fc, Lc, Dq, feedc); break;
                case 3: run_cov_sqrtm_sweep<3>(f, Gc, Fc, Lc, Dq, feedc); break;
                case 4: run_cov_sqrtm_sweep<4>(f, Gc, Fc, Lc, Dq, feedc); break;
                default: rc = -1;
            }
            hF.resize((size_t)Dq * Dq);
            e = hipMemcpyAsync(hF.data(), Fc, sizeof(double) * hF.size(), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(&inf, f->info_err, sizeof(int), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) e = hipGetLastError();
        }
    }
    for (void* p : {(void*)Gc, (void*)Fc, (void*)Lc, (void*)dsc, (void*)feedc})
        if (p) (void)hipFree(p);
    if (e != hipSuccess) {
        ctx->err = std::string("pnmol_state_get_cov_sqrtm: ") + hipGetErrorString(e);
        return -2;
    }
    if (rc != 0) return rc;
    if (inf == -2) {
        ctx->err = "pnmol_state_get_cov_sqrtm: a dependency wait timed out";
        return -2;
    }
    if (inf < Dq) {
        ctx->err = "pnmol_state_get_cov_sqrtm: covariance not positive semi-definite at pivot " + std::to_string(inf);
        return -3;
    }
    for (long r = 0; r < D; ++r) std::memcpy(C_DD + r * D, &hF[(size_t)r * Dq], sizeof(double) * D);
    return 0;
}

int pnmol_state_get_marginal_var(const pnmol_state* s, double* var_nd) {
    if (!s || !var_nd) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<double> hv((size_t)f->Dp);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(hv.data(), s->var, sizeof(double) * hv.size(), hipMemcpyDeviceToHost));
    double sc[MAXN];
    frame_scales(s, sc);
    for (int a = 0; a < f->n; ++a)
        for (int j = 0; j < f->ds; ++j) var_nd[(size_t)a * f->ds + j] = sc[a] * sc[a] * hv[(size_t)a * f->dp + j];
    return 0;
}

int pnmol_state_get_cov(const pnmol_state* s, double* cov_DD) {
    if (!s || !cov_DD) return -1;
    pnmol_filter* f = s->f;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int n = f->n, d = f->ds, dp = f->dp;
    const long D = (long)n * d, Dp = f->Dp;
    std::vector<double> hP((size_t)Dp * Dp);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (f->p32) {
        std::vector<float> hPf(hP.size());
        HIPCHK(ctx, hipMemcpy(hPf.data(), s->P, sizeof(float) * hPf.size(), hipMemcpyDeviceToHost));
        std::copy(hPf.begin(), hPf.end(), hP.begin());
    } else {
        HIPCHK(ctx, hipMemcpy(hP.data(), s->P, sizeof(double) * hP.size(), hipMemcpyDeviceToHost));
    }
    double sc[MAXN];
    frame_scales(s, sc);
    for (int a = 0; a < n; ++a)
        for (int j = 0; j < d; ++j) {
            double* dst = cov_DD + ((size_t)j * n + a) * D;
            const double* src = &hP[((size_t)a * dp + j) * Dp];
            for (int b = 0; b < n; ++b)
                for (int k = 0; k < d; ++k) dst[(size_t)k * n + b] = sc[a] * sc[b] * src[(size_t)b * dp + k];
        }
    return 0;
}

int pnmol_filter_step(pnmol_filter* f, const pnmol_state* in, double dt, pnmol_state* out, pnmol_step_out* info,
                      double* error_estimate_d) {
    if (!f || !in || !out || in == out || in->f != f || out->f != f || !(dt > 0.0)) {
        if (f) f->ctx->err = "pnmol_filter_step: bad argument (null, aliasing states, foreign state or dt <= 0)";
        return -1;
    }
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    k_init_call<<<1, 64, 0, ctx->stream>>>(f->info, 1, f->ctr, f->last_ctr);
    int rc = dispatch_step(f, in->P, in->mean, in->frame_dt, dt, out->P, out->mean, out->var, false);
    if (rc != 0) return rc;
    k_copy_out<<<1, 256, 0, ctx->stream>>>(f->rec, nullptr, nullptr, f->info, f->h_pin_dev, 1, f->d, f->rec_cap);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    double rec[4];
    std::memcpy(rec, f->h_pin, sizeof(rec));
    const int inf = *reinterpret_cast<const int*>(f->h_pin + (size_t)4 * f->rec_cap + 2 * (size_t)f->rec_cap * f->d);
    out->t = in->t + dt;
    out->frame_dt = dt;
    const bool have_sq = (f->Sqinv != nullptr && f->sq_dt == dt);
    pnmol_step_out o;
    fill_out(f, rec, inf, out->t, have_sq, &o);
    if (info) *info = o;
    if (error_estimate_d) {
        // white.py:117-129: error = dt * sqrt(diag Sq) * sigma, truncated to the d PDE rows
        for (int i = 0; i < f->d; ++i)
            error_estimate_d[i] = have_sq ? dt * std::sqrt(f->sqdiag[i]) * std::sqrt(o.error_sigma2) : std::nan("");
    }
    if (o.info == -2) {
        ctx->err = "sweep kernel: a dependency wait timed out (workgroups not co-scheduled?)";
        return -2;
    }
    if (o.info >= 0) {
        ctx->err = "innovation matrix not positive definite at pivot " + std::to_string(o.info);
        return -3;
    }
    return 0;
}

int pnmol_filter_steps_begin(pnmol_filter* f, pnmol_state* s, int k, double dt) {
    if (!f || !s || s->f != f || k < 1 || !(dt > 0.0)) {
        if (f) f->ctx->err = "pnmol_filter_steps_begin: bad argument";
        return -1;
    }
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_rec(f, k);
    if (rc != 0) return rc;
    hipStream_t st = ctx->stream;
    k_init_call<<<1, 256, 0, st>>>(f->info, k, f->ctr, f->last_ctr);
    const bool fused = fused_loop(f);
    double *curP = s->P, *curM = s->mean, *nxtP = f->tmpP, *nxtM = f->tmpMean;
    double frame = s->frame_dt;
    const bool trace = std::getenv("PNMOL_HIP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tA = now();
    hipGraphExec_t gbig = nullptr, gpair = nullptr;
    rc = prepare_graphs(f, s, k, dt, &gbig, &gpair);
    if (rc != 0) return rc;
    const double tB = now();
    HIPCHK(ctx, hipEventRecord(f->ev0, st));
    int it = 0;
    while (it < k) {
        const int left = k - it;
        const bool may_graph = frame == dt && (it > 0 || !fused);  // fused loop: step 0 does the full predict (eager)
        if (may_graph && gbig && left >= f->graph_chunk) {
            HIPCHK(ctx, hipGraphLaunch(gbig, st));
            for (auto& g : f->graphs) g.launched = g.launched || g.exec == gbig;
            it += f->graph_chunk;  // even number of steps: buffers are back where they started
            continue;
        }
        if (may_graph && gpair && left >= 2) {
            HIPCHK(ctx, hipGraphLaunch(gpair, st));
            for (auto& g : f->graphs) g.launched = g.launched || g.exec == gpair;
            it += 2;
            continue;
        }
        rc = dispatch_step(f, curP, curM, frame, dt, nxtP, nxtM, s->var, true,
                           !fused ? STEP_FULL : (it == 0 ? STEP_FIRST : STEP_STEADY));
        if (rc != 0) return rc;
        double* t;
        t = curP, curP = nxtP, nxtP = t;
        t = curM, curM = nxtM, nxtM = t;
        frame = dt;
        ++it;
    }
    HIPCHK(ctx, hipEventRecord(f->ev1, st));
    const double tC = now();
    // the filter owns tmpP/tmpMean: after an odd number of steps the result lives there -> swap ownership
    if (curP != s->P) {
        f->tmpP = s->P, f->tmpMean = s->mean;
        s->P = curP, s->mean = curM;
    }
    // device -> pinned staging (allocated in ensure_rec); handed to the caller by pnmol_filter_steps_end
    k_copy_out<<<64, 256, 0, st>>>(f->rec, f->rec_means, f->rec_stds, f->info, f->h_pin_dev, k, f->d, f->rec_cap);
    if (trace)
        std::fprintf(stderr, "[pnmol] steps_begin(k=%d): prepare %.2f ms, enqueue %.2f ms, graphs big=%d pair=%d\n", k, tB - tA,
                     tC - tB, gbig != nullptr, gpair != nullptr);
    f->pending_k = k;
    f->pending_dt = dt;
    f->pending_state = s;
    return 0;
}

int pnmol_filter_steps_end(pnmol_filter* f, pnmol_state* s, double* means_kd, double* stds_kd, pnmol_step_out* info_k) {
    if (!f || !s || f->pending_state != s || f->pending_k < 1) {
        if (f) f->ctx->err = "pnmol_filter_steps_end: no matching pnmol_filter_steps_begin";
        return -1;
    }
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int k = f->pending_k;
    const double dt = f->pending_dt;
    f->pending_k = 0, f->pending_state = nullptr;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipEventElapsedTime(&f->last_ms, f->ev0, f->ev1));
    const double* rec = f->h_pin;
    const double* hm = rec + (size_t)4 * f->rec_cap;
    const double* hs = hm + (size_t)f->rec_cap * f->d;
    const int* inf = reinterpret_cast<const int*>(hs + (size_t)f->rec_cap * f->d);
    if (means_kd) std::memcpy(means_kd, hm, sizeof(double) * (size_t)k * f->d);
    if (stds_kd) std::memcpy(stds_kd, hs, sizeof(double) * (size_t)k * f->d);
    const bool have_sq = (f->Sqinv != nullptr && f->sq_dt == dt);
    int bad = -1, stuck = -1;
    double t = s->t;
    for (int it = 0; it < k; ++it) {
        t += dt;
        pnmol_step_out o;
        fill_out(f, &rec[(size_t)4 * it], inf[it], t, have_sq, &o);
        if (info_k) info_k[it] = o;
        if (o.info >= 0 && bad < 0) bad = it;
        if (o.info == -2 && stuck < 0) stuck = it;
    }
    s->t = t;
    s->frame_dt = dt;
    if (stuck >= 0) {
        ctx->err = "sweep kernel: a dependency wait timed out in step " + std::to_string(stuck);
        return -2;
    }
    if (bad >= 0) {
        ctx->err = "innovation matrix not positive definite in step " + std::to_string(bad);
        return -3;
    }
    return 0;
}

int pnmol_filter_steps(pnmol_filter* f, pnmol_state* s, int k, double dt, double* means_kd, double* stds_kd,
                       pnmol_step_out* info_k) {
    const int rc = pnmol_filter_steps_begin(f, s, k, dt);
    if (rc != 0) return rc;
    return pnmol_filter_steps_end(f, s, means_kd, stds_kd, info_k);
}

int pnmol_filter_prepare_steps(pnmol_filter* f, pnmol_state* s, int k, double dt) {
    if (!f || !s || s->f != f || k < 1 || !(dt > 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ensure_rec(f, k);
    if (rc != 0) return rc;
    // Rehearse the exact launch sequence once on the live buffers and restore them: graph capture,
    // instantiation and the runtime's first-launch work (tens of ms on ROCm 7.2) then happen here.
    const size_t Dp = (size_t)f->Dp, nP = Dp * Dp * f->psz / sizeof(double);  // (doubles the covariance occupies)
    double* bak = nullptr;
    HIPCHK(ctx, hipMalloc(&bak, sizeof(double) * (nP + 2 * Dp)));
    hipStream_t st = ctx->stream;
    const double t0 = s->t, frame0 = s->frame_dt;
    hipError_t e = hipMemcpyAsync(bak, s->P, sizeof(double) * nP, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(bak + nP, s->mean, sizeof(double) * Dp, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(bak + nP + Dp, s->var, sizeof(double) * Dp, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) {
        rc = pnmol_filter_steps(f, s, k, dt, nullptr, nullptr, nullptr);
        if (rc == -3) rc = 0;  // a non-PD rehearsal is reported by the real call
        if ((k & 1) && rc == 0) {  // odd k swapped buffer ownership: swap back so the cached graphs stay valid
            double* t;
            t = s->P, s->P = f->tmpP, f->tmpP = t;
            t = s->mean, s->mean = f->tmpMean, f->tmpMean = t;
        }
        e = hipMemcpyAsync(s->P, bak, sizeof(double) * nP, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->mean, bak + nP, sizeof(double) * Dp, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(s->var, bak + nP + Dp, sizeof(double) * Dp, hipMemcpyDeviceToDevice, st);
        s->t = t0, s->frame_dt = frame0;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(bak);
    if (e != hipSuccess) {
        ctx->err = std::string("pnmol_filter_prepare_steps: ") + hipGetErrorString(e);
        return -2;
    }
    return rc;
}

int pnmol_filter_debug_poison(pnmol_filter* f) {
    if (!f) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const long nF = (long)f->RT * NB * f->mp, nL = (long)f->CB * NB * NB;
    const long nH = (long)std::max(f->CB * f->CB, 2 * f->CB + 2) * NB * NB;
    k_poison<<<1024, 256, 0, ctx->stream>>>(f->F, nF, f->Linv, nL, f->hs_scratch, nH);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

int pnmol_filter_last_steps_ms(pnmol_filter* f, float* ms) {
    if (!f || !ms) return -1;
    *ms = f->last_ms;
    return 0;
}

int pnmol_filter_debug_read(pnmol_filter* f, int which, double* dst, long count) {
#ifdef PNMOL_SWEEP_STAMP
    if (f && which == 5) {
        hipStreamSynchronize(f->ctx->stream);
        std::vector<long long> h(512 * 8);
        if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(pnmol_sweep_stamp), sizeof(long long) * 512 * 8) != hipSuccess) return -2;
        for (long i = 0; i < count && i < 512 * 8; ++i) dst[i] = (double)h[i];
        return 0;
    }
#endif
    if (!f || !dst || count < 0) return -1;
    pnmol_ctx* ctx = f->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double* src = nullptr;
    long avail = 0;
    const long tall = (long)f->RT * NB * f->mp;
    switch (which) {
        case 0: src = f->Ppred, avail = f->Dp * f->Dp; break;
        case 1: src = f->G, avail = tall; break;
        case 2: src = f->F, avail = tall; break;
        case 3: src = f->mpred, avail = f->Dp; break;
        case 4: src = f->zbuf, avail = f->mp; break;
        default: return -1;
    }
    if (count > avail) return -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (which == 0 && f->p32) {  // fp32 covariance: widened for the caller
        std::vector<float> h((size_t)count);
        HIPCHK(ctx, hipMemcpy(h.data(), src, sizeof(float) * count, hipMemcpyDeviceToHost));
        std::copy(h.begin(), h.end(), dst);
        return 0;
    }
    HIPCHK(ctx, hipMemcpy(dst, src, sizeof(double) * count, hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
