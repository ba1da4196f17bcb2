// pnmol_sqrt.hip -- square-root (QR) primitives of the PNMOL filter on MI355X (gfx950): fp64, and an fp32 build of the QR
// (work matrix, reflectors and trailing updates in fp32 on v_mfma_f32_16x16x4_f32) for the filter's dtype = 1 mode.
//
// What the reference does with `jnp.linalg.qr(mode="r")` (base/sqrt.py:8-95): the R factor of a tall stacked matrix.
// Here: a communication-avoiding Householder QR (TSQR panels + compact-WY trailing updates on the f64 MFMA).
//
//   work matrix W   row-major, Mp x ld doubles (floats in the fp32 build), Mp and ld multiples of 32 (zero padded), Mp >= ld
//   panel p         columns [32p, 32p+32); active row blocks p .. nrb-1 (32 rows each)
//   tree level s    members = row blocks p + s*k; a chunk = FAN consecutive members (FAN*32 stacked rows).  k_qr_factor
//                   does a Householder QR of the chunk's 32 panel columns in registers and LDS (LAPACK dlarfg/dlarft
//                   conventions), leaves R in the chunk's first member and V, T in a workspace; k_qr_apply applies
//                   (I - V T V^T)^T to the chunk's rows of every trailing column.  Survivors (first members) form level
//                   s*FAN, until one is left.
//   launches        per panel: factor(level 1); [apply(level s) + factor(level 8 s)] in ONE launch (k_qr_apply_factor: the two
//                   do not depend on each other); apply(last level) -- qr_launch_panel.
//
// Only R is kept (the reference never forms Q either).  Rows of R are sign-normalised to a non-negative diagonal at the
// end: the canonical representative of the reference's factor (its signs are LAPACK's, arbitrary).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pnmol_internal.hpp"
#include "pnmol_sqrt.h"

namespace {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int QB = 32;            // panel width = row-block height
#ifndef PNMOL_QR_FAN
#define PNMOL_QR_FAN 8
#endif
constexpr int FAN = PNMOL_QR_FAN; // members per chunk (8, or 4: -DPNMOL_QR_FAN=4 -- half the waves per factor block, one tree level more)
static_assert(FAN == 8 || FAN == 4, "the factor kernel's row groups are 16 or 8 lanes of a DPP row");
constexpr int CR = FAN * QB;      // stacked rows of a chunk (256)
constexpr int FT = FAN * 64;      // threads of k_qr_factor: thread (k, g) = (t & 31, t >> 5), rows g, g + NG, ...
constexpr int NG = FT / 32;       // row groups (16)
constexpr int RPT = CR / NG;      // rows per thread (16)

// What differs between the two 16x16x4 MFMA forms: A[l & 15][k = l >> 4] and B[k = l >> 4][l & 15] are the same, the
// accumulator is not -- register r of a lane with l >> 4 == fk is row fk + 4 r (f64) or row 4 fk + r (f32).  VLD: LDS row
// stride of V in k_qr_apply (bank-conflict-free for the row walks of the first product and the column walks of the last).
template <typename T> struct Mf;
template <> struct Mf<double> {
    typedef d4 acc_t;
    static constexpr int VLD = 33;
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int fk, int r) { return fk + 4 * r; }
};
template <> struct Mf<float> {
    typedef f4 acc_t;
    static constexpr int VLD = 36;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int fk, int r) { return 4 * fk + r; }
};

#define QCHECK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return -2;                                                                     \
        }                                                                                  \
    } while (0)

constexpr int FLD = QB + 1;   // LDS row stride of the chunk in k_qr_factor

template <typename T>
struct FactorLds {
    T A[CR * FLD];     // the chunk's panel columns (staging of the coalesced load), later V (unit lower trapezoidal)
    T R[QB * QB];      // rows of R as they are finished
    T G[QB * QB];      // V^T V
    T tau[QB], scale[QB], beta[QB];
    // column J of the working matrix (ping-pong by step parity), [row group g][r] = row g + NG r with a stride per row group
    // that keeps the 16-byte accesses of the 16 (or 8) lanes of a column on distinct banks: 20 floats / 18 doubles
    __attribute__((aligned(16))) T col[2][(CR / RPT) * 20];
    T rowb[2][QB];     // row J
};

// The row blocks a panel's reflectors act on ("members"), in tree order: `ntop` consecutive blocks from the panel's own
// diagonal block p, then blocks bot0, bot0+1, ... up to `cnt` members in all.  Dense matrix: ntop = cnt = nrb - p.  Two
// stacked triangular blocks (the predict QR): {p, p+1} and the first p+1 blocks of the lower one.
struct MemberMap {
    int ntop, bot0, cnt;
};
// Member q of chunk c at tree level s of panel p (-1: none).
__device__ __forceinline__ int member_rb(const MemberMap& mm, int p, int s, int c, int q) {
    const int idx = s * (FAN * c + q);
    if (idx >= mm.cnt) return -1;
    return idx < mm.ntop ? p + idx : mm.bot0 + (idx - mm.ntop);
}

// x + (x rotated right by N lanes within its row of 16 lanes): after N = 8, 4, 2, 1 every lane holds the row's sum
template <int N>
__device__ __forceinline__ double row_ror_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x120 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x120 + N, 0xf, 0xf, false);
    return x + __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ float row_ror_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false));
}
// x + (x moved by the DPP control word CTRL: 0xB1 / 0x4E = lanes xor 1 / xor 2 within a quad, 0x141 = mirror within 8 lanes)
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return x + __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// sum over the NGL (16 or 8) lanes of a column's row groups, left in all of them
template <int NGL, typename T>
__device__ __forceinline__ T group_sum(T x) {
    if constexpr (NGL == 16) {
        x = row_ror_add<8>(x), x = row_ror_add<4>(x), x = row_ror_add<2>(x), x = row_ror_add<1>(x);
    } else {   // 8 lanes: within the quads (xor 1, xor 2), then the two quads of a half row (mirror)
        x = dpp_add<0xB1>(x), x = dpp_add<0x4E>(x), x = dpp_add<0x141>(x);
    }
    return x;
}

// 1 / sqrt(x) and 1 / x to the type's precision from the hardware estimates (two Newton steps for fp64, one for fp32)
__device__ __forceinline__ double rsq_full(double x) {
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    return r * (1.5 - 0.5 * x * r * r);
}
__device__ __forceinline__ float rsq_full(float x) {
    const float r = __builtin_amdgcn_rsqf(x);
    return r * (1.5f - 0.5f * x * r * r);
}
__device__ __forceinline__ double rcp_full(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = r * (2.0 - x * r);
    return r * (2.0 - x * r);
}
__device__ __forceinline__ float rcp_full(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.0f - x * r);
}
// dlarfg: a column whose entries below the diagonal have a sum of squares at or below this counts as already eliminated (tau = 0).
// fp64: exactly zero, as LAPACK.  fp32: squares below ~1e-38 are denormal -- entries of 1e-19.5 .. 1e-22.5, which the noise-free
// Dirichlet rows produce from N ~ 1000 on (smaller ones square to zero and took the tau = 0 path all along) -- and v_rsq_f32 of a
// denormal is inf: NaNs from the first step at N = 1024.  Everything below 3e-18 in absolute value is dropped instead; the matrix's
// own entries are 1e-10 .. 1e3 at 6e-8 relative.
template <typename T> __device__ __forceinline__ T dlarfg_negligible();
template <> __device__ __forceinline__ double dlarfg_negligible<double>() { return 0.0; }
template <> __device__ __forceinline__ float dlarfg_negligible<float>() { return 1e-35f; }
__device__ __forceinline__ double abs_t(double x) { return fabs(x); }
__device__ __forceinline__ float abs_t(float x) { return fabsf(x); }
__device__ __forceinline__ double copysign_t(double x, double y) { return copysign(x, y); }
__device__ __forceinline__ float copysign_t(float x, float y) { return copysignf(x, y); }

// Householder QR of one chunk's panel (CR x 32).  Thread (k, g) = (t >> 4, t & 15) keeps rows g, g+16, ... of column k in
// registers, so the 16 row groups of a column sit in ONE wave: the inner products of column J with the wave's own four
// columns (and with itself: the dlarfg sigma, formed redundantly by every wave) are reduced over 16 lanes by shuffles --
// no LDS, no barrier.  What crosses waves is column J and row J of the current matrix (L.col, L.rowb, ping-pong by step
// parity, published by their owners at the end of the step before): ONE block barrier per column.  Row J of R goes to
// L.R; the reflector stays unscaled in the registers (scale[J] applied at the end).  Rows beyond the chunk's members are
// zero on input and stay zero.
// (defined behind k_qr_factor) the trailing update of one chunk on one 16-column slab per wave, result left in registers
template <typename T, int NT>
__device__ __forceinline__ bool qr_apply_core(unsigned char* lds_raw, const T* __restrict__ W, long ld, const MemberMap& mm, int p,
                                              int s, int c, const T* __restrict__ Vws, const T* __restrict__ Tws, long col,
                                              T (&cs)[CR / 4], int& nm);

// INLOOP (default): V^T V and dlarft's T are formed inside the column loop -- (V^T V)[k][J], k < J, is the inner product every
// thread of column k forms anyway, and column J-1 of T is a triangular product that the first wave (whose own columns are
// long finished) does at the top of step J -- and V leaves straight from the registers: no LDS image of V, no Gram product
// on the MFMA, no serial dlarft behind the loop (8-9 of the 13 us a launch spent outside its column loop).
// PRE: the LAST tree level of the panel before (members mma / pa / sa, one chunk, reflectors Va, Ta) has not been applied to
// this panel's columns yet -- its trailing update runs in this very launch, on the columns behind this panel --: a chunk that
// holds some of that level's row blocks forms their updated rows itself (two waves, 16 of the 32 columns each) and
// patches them into its staged copy.  Nothing is written back: below R, a factored panel's columns are never read again.
// MODE 2 (default) = INLOOP + the reflector's scalars made ONCE per column: the 16 lanes that own column J+1 form its
// dlarfg (norm of the rows below the diagonal over their DPP row, tau, scale, beta) right behind their update in step J and
// leave the three numbers in LDS beside the column; in step J+1 everybody reads them instead of forming them -- 16 FMAs, a
// DPP reduction and ~25 dependent scalar operations fewer per thread and column (of ~100 instructions; the loop is
// issue-bound at two waves per SIMD).  MODE 1 = INLOOP as before, MODE 0 = the round-1 form (PNMOL_QR_INLOOP / PNMOL_QR_OWNER).
template <typename T, int MODE = 2, bool PRE = false>
__device__ __forceinline__ void qr_factor_body(unsigned char* lds_raw, T* __restrict__ W, long ld, const MemberMap& mm, int p,
                                               int s, int c, T* __restrict__ Vws, T* __restrict__ Tws,
                                               const MemberMap& mma, int pa, int sa, const T* __restrict__ Va,
                                               const T* __restrict__ Ta, int* __restrict__ readers) {
    constexpr bool INLOOP = MODE >= 1, OWN = MODE == 2;
    constexpr int CST = sizeof(T) == 4 ? 20 : 18, VN = 16 / sizeof(T);   // stride of a row group in L.col, elements per 16 bytes
    typedef T colvec_t __attribute__((ext_vector_type(VN)));
    FactorLds<T>& L = *reinterpret_cast<FactorLds<T>*>(lds_raw);
    const int t = threadIdx.x, k = t / NG, g = t % NG, w = t >> 6, lane = t & 63;
    int nm = 0;
    for (int q = 0; q < FAN; ++q) nm += member_rb(mm, p, s, c, q) >= 0;
    if (nm == 0 || (nm == 1 && s > 1)) {  // a lone survivor is already triangular (k_qr_apply skips it too)
        if constexpr (PRE)
            if (t == 0) __hip_atomic_fetch_add(readers, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    T pre_cs[CR / 4];
    int jmap[FAN];
    bool have_pre = false;
    if constexpr (PRE) {
        int any = 0;
#pragma unroll
        for (int q = 0; q < FAN; ++q) {
            const int rbq = member_rb(mma, pa, sa, 0, q);
            jmap[q] = -1;
            if (rbq >= 0)
                for (int j = 0; j < nm; ++j)
                    if (member_rb(mm, p, s, c, j) == rbq) jmap[q] = j, any = 1;
        }
        if (any) {   // (block-uniform)
            int nma;
            have_pre = qr_apply_core<T, FT>(lds_raw, W, ld, mma, pa, sa, 0, Va, Ta, (long)p * QB + (w < 2 ? w * 16 : 0) + (lane & 15),
                                            pre_cs, nma);
            __syncthreads();   // the LDS image of V, T is dead: the chunk is staged in the same bytes
        }
        // This block has read the pending level's rows of this panel's columns (their values went through the products
        // above): the one row block of that level that is NOT a member of this panel -- the panel before's R row -- may now
        // be overwritten by the apply block that owns it (qr_apply_body, `writers_wait`).
        if (t == 0) __hip_atomic_fetch_add(readers, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int e = t; e < CR * QB; e += FT) {   // coalesced: 32 consecutive columns of one row
        const int i = e >> 5, kk = e & 31, q = i >> 5;
        L.A[i * FLD + kk] = (q < nm) ? W[((long)member_rb(mm, p, s, c, q) * QB + (i & 31)) * ld + (long)p * QB + kk] : T(0);
    }
    __syncthreads();
    if constexpr (PRE) {
        if (have_pre) {
            if (w < 2) {
                const int fr = lane & 15, fk = lane >> 4;
#pragma unroll
                for (int st = 0; st < CR / 4; ++st) {
                    const int j = jmap[st >> 3], i = 16 * (st >> 2) + Mf<T>::row(fk, st & 3);
                    if (j >= 0) L.A[(QB * j + (i & 31)) * FLD + w * 16 + fr] = pre_cs[st];
                }
            }
            __syncthreads();
        }
    }
    T a[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) a[r] = L.A[(g + NG * r) * FLD + k];
    auto col_store = [&](int buf) {   // this thread's RPT rows of its column, as 16-byte stores
        colvec_t* dst = reinterpret_cast<colvec_t*>(&L.col[buf][g * CST]);
#pragma unroll
        for (int q = 0; q < RPT / VN; ++q) {
            colvec_t x;
#pragma unroll
            for (int e = 0; e < VN; ++e) x[e] = a[q * VN + e];
            dst[q] = x;
        }
    };
    if (k == 0) col_store(0);
    if (g == 0) L.rowb[0][k] = a[0];
    // (OWN) dlarfg of column JN by the lanes that hold it (k == JN; the other column groups of the wave compute along and store
    // nothing): H = I - tau v v^T, v = [1; x / (alpha - beta)], beta = -sign(alpha) |(alpha, x)|; with nrm = |(alpha, x)|:
    // tau = 1 + |alpha| / nrm, 1 / (alpha - beta) = sign(alpha) / (|alpha| + nrm)
    auto owner_dlarfg = [&](const int JN) {   // (JN is a constant wherever this is expanded: the column loop is unrolled)
        T pq[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (NG * r + NG - 1 > JN) {
                const T v = (NG * r > JN || g + NG * r > JN) ? a[r] : T(0);
                pq[r & 3] += v * v;
            }
        T sJ = (pq[0] + pq[1]) + (pq[2] + pq[3]);
        sJ = group_sum<NG>(sJ);
        const T alpha = a[JN / NG];   // (the lane g == JN % NG holds row JN)
        T beta = alpha, tau = T(0), scale = T(0);
        if (!(sJ <= dlarfg_negligible<T>())) {
            const T n2 = alpha * alpha + sJ, aa = abs_t(alpha);
            const T rn = rsq_full(n2);
            const T nrm = n2 * rn, den = aa + nrm;
            const T rd = rcp_full(den);
            beta = -copysign_t(nrm, alpha);
            tau = T(1) + aa * rn;
            scale = copysign_t(rd, alpha);
        }
        if (k == JN && g == JN % NG) L.tau[JN] = tau, L.scale[JN] = scale, L.beta[JN] = beta;
    };
    if constexpr (OWN)
        if (w == 0) owner_dlarfg(0);
    __syncthreads();

    T my_scale = T(0), tau_prev = T(0);
    T trow[QB];   // threads t < 32: row t of T (dlarft, forward columnwise: T[0:J, J] = -tau_J T[0:J, 0:J] (V^T V)[0:J, J])
#pragma unroll
    for (int J = 0; J < QB; ++J) {
        const int cur = J & 1;
        if constexpr (INLOOP) {
            if (J > 0 && t < QB) {   // column J-1 of T from column J-1 of V^T V (written in step J-1, behind its barrier)
                T acc = T(0);
#pragma unroll
                for (int l = 0; l < J - 1; ++l) acc += trow[l] * L.G[l * QB + (J - 1)];
                trow[J - 1] = (t < J - 1) ? -tau_prev * acc : (t == J - 1 ? tau_prev : T(0));
            }
        }
        T vi[RPT], vc[RPT];
        {   // column J, this thread's rows: 16-byte loads
            const colvec_t* src = reinterpret_cast<const colvec_t*>(&L.col[cur][g * CST]);
#pragma unroll
            for (int q = 0; q < RPT / VN; ++q) {
                const colvec_t x = src[q];
#pragma unroll
                for (int e = 0; e < VN; ++e) vc[q * VN + e] = x[e];
            }
        }
        T pk[4] = {0, 0, 0, 0}, pJ[4] = {0, 0, 0, 0};   // four independent accumulation chains each
#pragma unroll
        for (int r = 0; r < RPT; ++r) {   // rows i = g + 16 r > J of column J (r and J are compile-time after unrolling)
            if (NG * r + NG - 1 <= J) {
                vi[r] = T(0);
            } else {
                const T v = vc[r];
                vi[r] = (NG * r > J || g + NG * r > J) ? v : T(0);
                pk[r & 3] += vi[r] * a[r];
                if constexpr (!OWN) pJ[r & 3] += vi[r] * vi[r];
            }
        }
        T sk = (pk[0] + pk[1]) + (pk[2] + pk[3]), sJ = (pJ[0] + pJ[1]) + (pJ[2] + pJ[3]);
        // over the column's 16 row groups = one DPP row of the wave
        if constexpr (OWN) {
            sk = group_sum<NG>(sk);
        } else if constexpr (NG == 16) {
            sk = row_ror_add<8>(sk), sJ = row_ror_add<8>(sJ);
            sk = row_ror_add<4>(sk), sJ = row_ror_add<4>(sJ);
            sk = row_ror_add<2>(sk), sJ = row_ror_add<2>(sJ);
            sk = row_ror_add<1>(sk), sJ = row_ror_add<1>(sJ);
        } else {   // 8 lanes: within the quads (xor 1, xor 2), then the two quads of a half row (mirror)
            sk = dpp_add<0xB1>(sk), sJ = dpp_add<0xB1>(sJ);
            sk = dpp_add<0x4E>(sk), sJ = dpp_add<0x4E>(sJ);
            sk = dpp_add<0x141>(sk), sJ = dpp_add<0x141>(sJ);
        }
        // dlarfg: H = I - tau v v^T, v = [1; x / (alpha - beta)], beta = -sign(alpha) |(alpha, x)|; with
        // nrm = |(alpha, x)|: tau = (beta - alpha) / beta = 1 + |alpha| / nrm, 1 / (alpha - beta) = sign(alpha) / (|alpha| + nrm)
        // -- one rsqrt and one reciprocal on the step's critical path instead of a sqrt and two divisions
        const T alpha = L.rowb[cur][J], aJk = L.rowb[cur][k];
        T beta = alpha, tau = T(0), scale = T(0);
        if constexpr (OWN) {
            tau = L.tau[J], scale = L.scale[J], beta = L.beta[J];
        } else if (!(sJ <= dlarfg_negligible<T>())) {
            const T n2 = alpha * alpha + sJ, aa = abs_t(alpha);
            const T rn = rsq_full(n2);
            const T nrm = n2 * rn, den = aa + nrm;
            const T rd = rcp_full(den);
            beta = -copysign_t(nrm, alpha);
            tau = T(1) + aa * rn;
            scale = copysign_t(rd, alpha);
        }
        const T f = tau * (aJk + sk * scale);  // tau v^T A[:, k]
        if constexpr (INLOOP) {
            // k < J: a[] holds reflector k (unscaled) below row k, aJk is its row J, sk its product with rows > J of column J:
            // v_k^T v_J = scale_k (aJk + scale_J sk)
            if (k == J) my_scale = scale;
            if (g == 0 && k < J) L.G[k * QB + J] = my_scale * (aJk + scale * sk);
            tau_prev = tau;
        }
        if (g == 0) {
            L.R[J * QB + k] = (k > J) ? aJk - f : (k == J ? beta : T(0));
            if (!OWN && k == J) {
                L.tau[J] = tau;
                L.scale[J] = scale;
            }
        }
        if (k > J) {
            const T fs = f * scale;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (NG * r + NG - 1 > J) a[r] -= fs * vi[r];
        }
        if (J + 1 < QB) {
            if (w == ((J + 1) * NG) >> 6) {        // the wave that owns column J+1
                if (k == J + 1) col_store(cur ^ 1);
                if constexpr (OWN) owner_dlarfg(J + 1);
            }
            if (g == (J + 1) % NG) L.rowb[cur ^ 1][k] = a[(J + 1) / NG];
        }
        __syncthreads();
    }

    T* Tg = Tws + (long)c * QB * QB;
    T* Vg = Vws + (long)c * CR * QB;
    if constexpr (INLOOP) {
        if (t < QB) {   // last column of T, then the rows of T
            T acc = T(0);
#pragma unroll
            for (int l = 0; l < QB - 1; ++l) acc += trow[l] * L.G[l * QB + (QB - 1)];
            trow[QB - 1] = (t < QB - 1) ? -tau_prev * acc : tau_prev;
#pragma unroll
            for (int J = 0; J < QB; ++J) Tg[t * QB + J] = trow[J];
        }
        // V: unit lower trapezoidal, straight from the registers (column k is rows i > k, scaled now)
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = g + NG * r;
            Vg[i * QB + k] = (i > k) ? a[r] * my_scale : (i == k ? T(1) : T(0));
        }
    } else {
        // V: unit lower trapezoidal (column k of the reflectors is rows i > k of the registers, unscaled so far)
        {
            const T sc = L.scale[k];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int i = g + NG * r;
                L.A[i * FLD + k] = (i > k) ? a[r] * sc : (i == k ? T(1) : T(0));
            }
        }
        __syncthreads();
        // G = V^T V on the MFMA: waves 0..3 take one 16x16 tile each
        if (w < 4) {
            const int fr = lane & 15, fk = lane >> 4, jt = w >> 1, ct = w & 1;
            typename Mf<T>::acc_t acc = {0, 0, 0, 0};
            for (int st = 0; st < CR / 4; ++st) {
                const T va = L.A[(4 * st + fk) * FLD + jt * 16 + fr];
                const T vb = L.A[(4 * st + fk) * FLD + ct * 16 + fr];
                acc = Mf<T>::mfma(va, vb, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) L.G[(jt * 16 + Mf<T>::row(fk, r)) * QB + ct * 16 + fr] = acc[r];
        }
        __syncthreads();
        // dlarft (forward, columnwise): T[0:J, J] = -tau_J T[0:J, 0:J] (V^T V)[0:J, J]; lane j keeps row j of T
        if (t < QB) {
#pragma unroll
            for (int J = 0; J < QB; ++J) {
                T acc = T(0);
#pragma unroll
                for (int l = 0; l < J; ++l) acc += trow[l] * L.G[l * QB + J];   // trow[l] = 0 for l < t
                const T tj = L.tau[J];
                trow[J] = (t < J) ? -tj * acc : (t == J ? tj : T(0));
            }
#pragma unroll
            for (int J = 0; J < QB; ++J) Tg[t * QB + J] = trow[J];
        }
        // V to the workspace: coalesced from LDS
        for (int e = t; e < CR * QB; e += FT) Vg[e] = L.A[(e >> 5) * FLD + (e & 31)];
    }
    // R (upper, zero below) to the first member
    for (int e = t; e < QB * QB; e += FT)
        W[((long)member_rb(mm, p, s, c, 0) * QB + (e >> 5)) * ld + (long)p * QB + (e & 31)] = L.R[e];
}

template <typename T, int MODE = 2>
__global__ __launch_bounds__(FT) void k_qr_factor(T* __restrict__ W, long ld, MemberMap mm, int p, int s,
                                                  T* __restrict__ Vws, T* __restrict__ Tws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    qr_factor_body<T, MODE, false>(qr_lds_raw, W, ld, mm, p, s, blockIdx.x, Vws, Tws, mm, 0, 0, nullptr, nullptr, nullptr);
}

// C <- (I - V T V^T)^T C = C - V (T^T (V^T C)) on the chunk's rows of 64 trailing columns per block, 16 per wave.  The
// wave keeps its 256 x 16 slab of C in registers in the MFMA's B-operand layout (lane (fr, fk) holds C[4s + fk][fr]),
// which is also the accumulator layout of the 16-row tile s / 4 (row fk + 4 (s % 4)): the last product accumulates
// straight into the slab.
// fp32 build: the slab's element st of a lane with l >> 4 == fk is row 16 (st / 4) + Mf::row(fk, st % 4) -- the order in
// which the f32 accumulator holds a 16-row tile -- and the first product walks V in that order too (a sum over rows does
// not care which lane group brings which row).
// NT threads: NT / 64 waves, 16 columns each; cg: the block's column group (NT / 4 columns).
template <typename T, int NT>
__device__ __forceinline__ bool qr_apply_core(unsigned char* lds_raw, const T* __restrict__ W, long ld, const MemberMap& mm, int p,
                                              int s, int c, const T* __restrict__ Vws, const T* __restrict__ Tws, long col,
                                              T (&cs)[CR / 4], int& nm) {
    constexpr int VLD = Mf<T>::VLD;
    T* sV = reinterpret_cast<T*>(lds_raw);     // CR x VLD
    T* sT = sV + CR * VLD;                      // 32 x VLD
    T* sWall = sT + QB * VLD;                   // per wave: 32 x 17
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4;
    nm = 0;
    for (int q = 0; q < FAN; ++q) nm += member_rb(mm, p, s, c, q) >= 0;
    if (nm == 0 || (nm == 1 && s > 1)) return false;
    const T* Vg = Vws + (long)c * CR * QB;
    const T* Tg = Tws + (long)c * QB * QB;
    for (int e = t; e < CR * QB; e += NT) sV[(e >> 5) * VLD + (e & 31)] = Vg[e];
    for (int e = t; e < QB * QB; e += NT) sT[(e >> 5) * VLD + (e & 31)] = Tg[e];
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        const int i = 16 * (st >> 2) + Mf<T>::row(fk, st & 3), q = i >> 5;
        cs[st] = (q < nm) ? W[((long)member_rb(mm, p, s, c, q) * QB + (i & 31)) * ld + col] : T(0);
    }
    __syncthreads();

    // W1 = V^T C  (32 x 16)
    typename Mf<T>::acc_t w1[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        const int i = 16 * (st >> 2) + Mf<T>::row(fk, st & 3);
        w1[0] = Mf<T>::mfma(sV[i * VLD + fr], cs[st], w1[0]);
        w1[1] = Mf<T>::mfma(sV[i * VLD + 16 + fr], cs[st], w1[1]);
    }
    T* sW = sWall + w * (QB * 17);
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[(jt * 16 + Mf<T>::row(fk, r)) * 17 + fr] = w1[jt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // W2 = T^T W1: D[j][n] = sum_l T[l][j] W1[l][n]
    typename Mf<T>::acc_t w2[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int st = 0; st < QB / 4; ++st) {
        const T b = sW[(4 * st + fk) * 17 + fr];
        w2[0] = Mf<T>::mfma(sT[(4 * st + fk) * VLD + fr], b, w2[0]);
        w2[1] = Mf<T>::mfma(sT[(4 * st + fk) * VLD + 16 + fr], b, w2[1]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[(jt * 16 + Mf<T>::row(fk, r)) * 17 + fr] = w2[jt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    T b2[QB / 4];
#pragma unroll
    for (int st = 0; st < QB / 4; ++st) b2[st] = sW[(4 * st + fk) * 17 + fr];
    // C -= V W2, 16 rows at a time
#pragma unroll
    for (int it = 0; it < CR / 16; ++it) {
        typename Mf<T>::acc_t acc = {cs[4 * it], cs[4 * it + 1], cs[4 * it + 2], cs[4 * it + 3]};
#pragma unroll
        for (int st = 0; st < QB / 4; ++st)
            acc = Mf<T>::mfma(-sV[(it * 16 + fr) * VLD + 4 * st + fk], b2[st], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[4 * it + r] = acc[r];
    }
    return true;
}

// FIRST (k_qr_apply_factor<.., PRE>): the first 32 columns are the NEXT panel's.  Of those only the level's first row block -- the
// R row of this panel, not a member of the next one -- is stored (the factor blocks of the same launch bring the other rows'
// update along in registers, and read all of them from W): after `readers` has reached `readers_target`, i.e. after every
// factor block of the launch has read them.  The factor blocks come first in the dispatch order and wait for nobody.
template <typename T, int NT, bool FIRST = false>
__device__ __forceinline__ void qr_apply_body(unsigned char* lds_raw, T* __restrict__ W, long ld, const MemberMap& mm, int p,
                                              int s, int c, int cg, const T* __restrict__ Vws, const T* __restrict__ Tws,
                                              int col0, int ncols, const int* readers = nullptr, int readers_target = 0) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, fr = lane & 15, fk = lane >> 4;
    const int cw = cg * (NT / 4) + w * 16;         // this wave's first column, relative to col0
    const bool active = cw < ncols;                // ncols is a multiple of 16
    const long col = col0 + (active ? cw : 0) + fr;
    T cs[CR / 4];
    int nm;
    if (!qr_apply_core<T, NT>(lds_raw, W, ld, mm, p, s, c, Vws, Tws, col, cs, nm) || !active) return;
    int qend = nm;
    if constexpr (FIRST) {
        if (cw < QB) {   // (wave-uniform)
            qend = 1;
            for (int spin = 0; spin < (1 << 24); ++spin) {
                if (__hip_atomic_load(readers, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - readers_target >= 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
#pragma unroll
    for (int st = 0; st < CR / 4; ++st) {
        const int i = 16 * (st >> 2) + Mf<T>::row(fk, st & 3), q = i >> 5;
        if (q < qend) W[((long)member_rb(mm, p, s, c, q) * QB + (i & 31)) * ld + col] = cs[st];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_qr_apply(T* __restrict__ W, long ld, MemberMap mm, int p, int s,
                                                  const T* __restrict__ Vws, const T* __restrict__ Tws,
                                                  int col0, int ncols) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    qr_apply_body<T, 256>(qr_lds_raw, W, ld, mm, p, s, blockIdx.x, blockIdx.y, Vws, Tws, col0, ncols);
}

// One launch for two things that do not depend on each other: the trailing update of tree level `sa` (reads the
// reflectors of that level, touches the columns behind the panel) and the panel factorisation of the NEXT level `sf` (touches the panel's
// columns of the level-`sa` survivors, which k_qr_factor of level `sa` left final).  Blocks [0, nchf) factorise -- first in
// the dispatch order, they are the longer ones --, the rest apply: block nchf + cg * ncha + c is chunk c, column group cg.
// PRE: the apply is the LAST level of panel pa = pf - 1 on the columns behind panel pf, the factorisation the first level of
// panel pf, whose blocks bring that level's update of their own columns along (qr_factor_body).
template <typename T, int MODE = 2, bool PRE = false>
__global__ __launch_bounds__(FT) void k_qr_apply_factor(T* __restrict__ W, long ld, MemberMap mma, int pa, int sa, int ncha,
                                                        const T* __restrict__ Va, const T* __restrict__ Ta, int col0,
                                                        int ncols, MemberMap mmf, int pf, int sf, int nchf,
                                                        T* __restrict__ Vf, T* __restrict__ Tf, int* readers, int readers_target) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    const int b = blockIdx.x;
    if (b < nchf) {
        qr_factor_body<T, MODE, PRE>(qr_lds_raw, W, ld, mmf, pf, sf, b, Vf, Tf, mma, pa, sa, Va, Ta, readers);
    } else {
        const int a = b - nchf;
        qr_apply_body<T, FT, PRE>(qr_lds_raw, W, ld, mma, pa, sa, a % ncha, a / ncha, Va, Ta, col0, ncols, readers, readers_target);
    }
}

// dst[j][i] = src[i][j], i < nr, j < nc (tile transpose through LDS)
template <typename DT>
__global__ __launch_bounds__(256) void k_copy_t(DT* __restrict__ dst, long ldd, const double* __restrict__ src,
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
        if (i < nr && j < nc) dst[(long)j * ldd + i] = (DT)tile[tx][r];
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
    int f32 = 0;                                  // element type of W, Vws, Tws: double or float
    int ws_chunks = 0;
    void *W = nullptr, *Vws = nullptr, *Tws = nullptr;
    int* readers = nullptr;            // device counter of k_qr_apply_factor<.., PRE> (factor blocks that have read the pending rows)
    mutable int readers_target = 0;    // its value once every factor block launched so far has counted (wraps; compared by difference)
    // four sets of reflector workspaces, by the parities of tree level and panel (what one launch writes, the same launch's
    // other half -- a level behind, or a panel behind -- reads from another set)
    template <typename T> T* vws(int lvl, int p) const { return static_cast<T*>(Vws) + (size_t)((lvl & 1) + 2 * (p & 1)) * ws_chunks * CR * QB; }
    template <typename T> T* tws(int lvl, int p) const { return static_cast<T*>(Tws) + (size_t)((lvl & 1) + 2 * (p & 1)) * ws_chunks * QB * QB; }
    size_t es() const { return f32 ? sizeof(float) : sizeof(double); }
    size_t bytes() const { return es() * (size_t)Mp * ld; }
    template <typename T> T* w() const { return static_cast<T*>(W); }
};

int qr_plan_alloc(pnmol_ctx* ctx, int rows, int cols, QrPlan* pl, int f32 = 0) {
    pl->rows = rows;
    pl->cols = cols;
    pl->f32 = f32;
    pl->ld = (cols + QB - 1) / QB * QB;
    pl->Mp = (std::max(rows, pl->ld) + QB - 1) / QB * QB;
    pl->nrb = pl->Mp / QB;
    pl->ncb = pl->ld / QB;
    const int maxchunks = (pl->nrb + FAN - 1) / FAN;
    QCHECK(ctx, hipMalloc(&pl->W, pl->bytes()));
    pl->ws_chunks = maxchunks;
    QCHECK(ctx, hipMalloc(&pl->Vws, 4 * pl->es() * (size_t)maxchunks * CR * QB));
    QCHECK(ctx, hipMalloc(&pl->Tws, 4 * pl->es() * (size_t)maxchunks * QB * QB));
    QCHECK(ctx, hipMalloc(&pl->readers, sizeof(int)));
    QCHECK(ctx, hipMemset(pl->readers, 0, sizeof(int)));
    pl->readers_target = 0;
    return 0;
}

void qr_plan_free(QrPlan* pl) {
    if (pl->W) hipFree(pl->W);
    if (pl->Vws) hipFree(pl->Vws);
    if (pl->Tws) hipFree(pl->Tws);
    if (pl->readers) hipFree(pl->readers);
    *pl = QrPlan();
}

template <typename T> constexpr size_t kFactorLds = sizeof(FactorLds<T>);
template <typename T, int NT> constexpr size_t kApplyLds = sizeof(T) * (CR * Mf<T>::VLD + QB * Mf<T>::VLD + (NT / 64) * QB * 17);
template <typename T> constexpr size_t kFusedLds = kFactorLds<T> > kApplyLds<T, FT> ? kFactorLds<T> : kApplyLds<T, FT>;

template <typename T>
int qr_configure_t(pnmol_ctx* ctx) {
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_factor<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFactorLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_factor<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFactorLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_factor<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFactorLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kApplyLds<T, 256>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply_factor<T, 0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply_factor<T, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply_factor<T, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds<T>));
    QCHECK(ctx, hipFuncSetAttribute((const void*)k_qr_apply_factor<T, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds<T>));
    return 0;
}

int qr_configure(pnmol_ctx* ctx) {   // attributes are per function and device; cheap to repeat
    if (int rc = qr_configure_t<double>(ctx)) return rc;
    return qr_configure_t<float>(ctx);
}

// A trailing update that has not been launched yet: the last tree level of a panel (one chunk), which goes out together with the
// first factorisation of the next panel (k_qr_apply_factor<.., PRE>).
struct PendingApply {
    bool on = false;
    MemberMap mm{};
    int p = 0, s = 0, lvl = 0;
};

// The launches of one panel.  factor(1) -- with the panel before's last trailing update in the same launch when one is
// pending --; [apply(s) + factor(8 s)] in one launch per further level; the last level's apply stays pending (or, without
// `pre`, is a launch of its own).  Switches (same arithmetic up to the order of sums, for A/B timings):
// PNMOL_QR_FUSE=0: factor and apply of a level one after the other.  PNMOL_QR_PRE=1: the last apply of a panel pending as
// described (default: a launch of its own -- faster, see qr_inplace).  PNMOL_QR_INLOOP=0: V^T V on the MFMA and dlarft behind the column loop, V through LDS (the first k_qr_factor).
template <typename T, int MODE>
void qr_launch_panel(pnmol_ctx* ctx, const QrPlan& pl, const MemberMap& mm, int p, int ntrail, bool fuse, bool pre,
                     PendingApply& pend) {
    T* W = pl.w<T>();
    const long ld = pl.ld;
    const int col0 = (p + 1) * QB;
    constexpr size_t apply_lds = kApplyLds<T, 256>;
    constexpr int NCG = FT / 4;   // columns per apply block of the fused launches
    int lvl = 0, s_prev = 0, nch_prev = 0;
    for (int s = 1;; s *= FAN, ++lvl) {
        const int nmem = (mm.cnt + s - 1) / s, nch = (nmem + FAN - 1) / FAN;
        if (lvl == 0 && pend.on) {
            if constexpr (MODE == 2) {
                // the pending level (one chunk) on this panel's columns (its R row only, see qr_apply_body) and everything behind
                const int ncols = ntrail + QB, ncg = (ncols + NCG - 1) / NCG;
                pl.readers_target += nch;
                hipLaunchKernelGGL((k_qr_apply_factor<T, 2, true>), dim3(nch + ncg), dim3(FT), kFusedLds<T>, ctx->stream, W, ld,
                                   pend.mm, pend.p, pend.s, 1, pl.vws<T>(pend.lvl, pend.p), pl.tws<T>(pend.lvl, pend.p), p * QB, ncols,
                                   mm, p, s, nch, pl.vws<T>(lvl, p), pl.tws<T>(lvl, p), pl.readers, pl.readers_target);
            }
            pend.on = false;
        } else if (lvl == 0 || ntrail <= 0 || !fuse) {
            hipLaunchKernelGGL((k_qr_factor<T, MODE>), dim3(nch), dim3(FT), kFactorLds<T>, ctx->stream, W, ld, mm, p, s,
                               pl.vws<T>(lvl, p), pl.tws<T>(lvl, p));
        } else {
            const int ncg = (ntrail + NCG - 1) / NCG;
            hipLaunchKernelGGL((k_qr_apply_factor<T, MODE, false>), dim3(nch + nch_prev * ncg), dim3(FT), kFusedLds<T>,
                               ctx->stream, W, ld, mm, p, s_prev, nch_prev, pl.vws<T>(lvl - 1, p), pl.tws<T>(lvl - 1, p), col0, ntrail,
                               mm, p, s, nch, pl.vws<T>(lvl, p), pl.tws<T>(lvl, p), (int*)nullptr, 0);
        }
        if (ntrail > 0 && (!fuse || nch == 1)) {
            if (pre && MODE == 2 && fuse && nch == 1) {
                pend.on = true, pend.mm = mm, pend.p = p, pend.s = s, pend.lvl = lvl;
            } else {
                hipLaunchKernelGGL(k_qr_apply<T>, dim3(nch, (ntrail + 63) / 64), dim3(256), apply_lds, ctx->stream, W, ld, mm,
                                   p, s, pl.vws<T>(lvl, p), pl.tws<T>(lvl, p), col0, ntrail);
            }
        }
        s_prev = s, nch_prev = nch;
        if (nch == 1) break;
    }
}

// R factor of the padded work matrix, in place (upper triangle of W's leading ld x ld block).  tri_bot0 > 0: the matrix
// is two stacked blocks, rows [0, 32 tri_bot0) and from row block tri_bot0 on, the upper one upper triangular up to
// `bulge` < 32 rows below its diagonal, the lower one upper triangular: panel p then only touches the member list above.
// stacked_tri > 0: a dense block of `stacked_tri` row blocks on top of an upper TRIANGULAR block (the one-QR step below):
// panel p has the dense rows from its diagonal on and the triangle's row blocks 0..p; once the dense rows are used up
// (p >= stacked_tri) the diagonal block is in the triangle, whose blocks p - stacked_tri .. p carry the panel.
int qr_inplace(pnmol_ctx* ctx, const QrPlan& pl, int tri_bot0 = 0, int stacked_tri = 0) {
    const bool fuse = !(std::getenv("PNMOL_QR_FUSE") && std::atoi(std::getenv("PNMOL_QR_FUSE")) == 0);
    const bool inloop = !(std::getenv("PNMOL_QR_INLOOP") && std::atoi(std::getenv("PNMOL_QR_INLOOP")) == 0);
    const bool owner = !(std::getenv("PNMOL_QR_OWNER") && std::atoi(std::getenv("PNMOL_QR_OWNER")) == 0);
    const int mode = !inloop ? 0 : (owner ? 2 : 1);
    // (measured, N = 512: the pending form is SLOWER -- fp64 7.53 against 7.45 ms per step, fp32 5.74 against 5.36: the update a
    //  factor block brings along, two waves on 256 x 16 slabs, costs more than the launch it saves -- off unless asked for)
    const bool pre = std::getenv("PNMOL_QR_PRE") && std::atoi(std::getenv("PNMOL_QR_PRE")) == 1;
    PendingApply pend;
    for (int p = 0; p < pl.ncb; ++p) {
        MemberMap mm;
        if (stacked_tri > 0) {
            const int nbot = pl.nrb - stacked_tri;
            if (p < stacked_tri) {
                mm.ntop = stacked_tri - p;
                mm.bot0 = stacked_tri;
                mm.cnt = mm.ntop + std::min(p + 1, nbot);
            } else {
                mm.ntop = mm.cnt = std::min(p, nbot - 1) - (p - stacked_tri) + 1;
                mm.bot0 = 0;
            }
        } else if (tri_bot0 > 0) {
            mm.ntop = std::min(2, tri_bot0 - p);
            mm.bot0 = tri_bot0;
            mm.cnt = mm.ntop + std::min(p + 1, pl.nrb - tri_bot0);
        } else {
            mm.ntop = mm.cnt = pl.nrb - p;
            mm.bot0 = 0;
        }
        const int ntrail = pl.ld - (p + 1) * QB;
        if (pl.f32) {
            if (mode == 2) qr_launch_panel<float, 2>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
            else if (mode == 1) qr_launch_panel<float, 1>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
            else qr_launch_panel<float, 0>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
        } else {
            if (mode == 2) qr_launch_panel<double, 2>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
            else if (mode == 1) qr_launch_panel<double, 1>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
            else qr_launch_panel<double, 0>(ctx, pl, mm, p, ntrail, fuse, pre, pend);
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
        if (hipMemsetAsync(pl.w<double>(), 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream) != hipSuccess) { rc = -2; break; }
        if (hipMemcpy2DAsync(pl.w<double>(), sizeof(double) * pl.ld, A, sizeof(double) * cols, sizeof(double) * cols, rows,
                             hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = -2; break; }
        {
            QrTimer tm(ctx->stream);
            rc = qr_inplace(ctx, pl);
            tm.stop();
        }
        if (rc) break;
        hipLaunchKernelGGL(k_qr_block, dim3((cols + 31) / 32, (cols + 7) / 8), dim3(32, 8), 0, ctx->stream, pl.w<double>(),
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
    QCHECK(ctx, hipMemsetAsync(pl.w<double>(), 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream));
    // [[C^T H^T, C^T], [E^T, 0]]
    hipLaunchKernelGGL(k_atbt, tiles(m, D), dim3(256), 0, ctx->stream, pl.w<double>(), (long)pl.ld, dC.p, (long)D, dH.p, (long)D, D, m, D);
    hipLaunchKernelGGL(k_copy_t<double>, tiles(D, D), dim3(256), 0, ctx->stream, pl.w<double>() + m, (long)pl.ld, dC.p, (long)D, D, D);
    if (E)
        hipLaunchKernelGGL(k_copy_t<double>, tiles(m, m), dim3(256), 0, ctx->stream, pl.w<double>() + (long)D * pl.ld, (long)pl.ld, dE.p, (long)m, m, m);
    if (int rc = run_qr_timed(ctx, pl)) return rc;
    const dim3 tb(32, 8);
    if (Sl) {   // R1^T
        hipLaunchKernelGGL(k_qr_block, dim3((m + 31) / 32, (m + 7) / 8), tb, 0, ctx->stream, pl.w<double>(), (long)pl.ld, 0, 0, m, m, dOut.p, 1, 1, 1);
        QCHECK(ctx, hipMemcpyAsync(Sl, dOut.p, sizeof(double) * (size_t)m * m, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (C_new) {   // R3^T
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_qr_block, dim3((D + 31) / 32, (D + 7) / 8), tb, 0, ctx->stream, pl.w<double>(), (long)pl.ld, m, m, D, D, dOut.p, 1, 1, 1);
        QCHECK(ctx, hipMemcpyAsync(C_new, dOut.p, sizeof(double) * (size_t)D * D, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (gain) {   // (R1^-1 R2)^T
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
        hipLaunchKernelGGL(k_trsm_upper, dim3((D + 31) / 32), dim3(256), 0, ctx->stream, pl.w<double>(), (long)pl.ld, m, m, D);
        hipLaunchKernelGGL(k_qr_block, dim3((D + 31) / 32, (m + 7) / 8), tb, 0, ctx->stream, pl.w<double>(), (long)pl.ld, 0, m, m, D, dOut.p, 1, 0, 0);
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
    QCHECK(ctx, hipMemsetAsync(pl.w<double>(), 0, sizeof(double) * (size_t)pl.Mp * pl.ld, ctx->stream));
    // vstack(S1^T, S2^T), base/sqrt.py:11
    hipLaunchKernelGGL(k_copy_t<double>, tiles(k1, n), dim3(256), 0, ctx->stream, pl.w<double>(), (long)pl.ld, d1.p, (long)k1, n, k1);
    if (S2)
        hipLaunchKernelGGL(k_copy_t<double>, tiles(k2, n), dim3(256), 0, ctx->stream, pl.w<double>() + (long)k1 * pl.ld, (long)pl.ld, d2.p, (long)k2, n, k2);
    if (int rc = run_qr_timed(ctx, pl)) return rc;
    hipLaunchKernelGGL(k_qr_block, dim3((n + 31) / 32, (n + 7) / 8), dim3(32, 8), 0, ctx->stream, pl.w<double>(), (long)pl.ld, 0, 0, n, n, dOut.p, 1, 1, 1);
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

// ==========================================================================================================
// The filter step in square-root form, as the reference writes it (white.py:96-146): the state is (mean, cov_sqrtm),
// the predict is one QR of [(A Pinv Cl)^T; Ql^T] (2D x D), the update one QR of [[R H^T, R], [E^T, 0]] ((D+m) x (m+D)).
// Everything in the reference's F-flattened state order (index j*n + a = derivative a at mesh point j), dense H.
// ==========================================================================================================
namespace {

constexpr int SQN = 4;   // n = nu + 1 <= 4

struct SqConst {
    double A1[SQN * SQN];   // base/iwp.py:17  flip(pascal lower)
    double p[SQN], pinv[SQN];   // Nordsieck scales of this step (base/iwp.py:55-62)
    int n;
};

// T1 = A Pinv Cl, A = I (x) A1 (white.py:101-103): thread (point j, column c) does the n rows of point j
__global__ void k_sq_rows(double* __restrict__ T1, const double* __restrict__ Cl, int d, int D, SqConst k) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (c >= D || j >= d) return;
    double v[SQN];
    for (int b = 0; b < k.n; ++b) v[b] = k.pinv[b] * Cl[(long)(j * k.n + b) * D + c];
    for (int a = 0; a < k.n; ++a) {
        double s = 0.0;
        for (int b = 0; b < k.n; ++b) s += k.A1[a * SQN + b] * v[b];
        T1[(long)(j * k.n + a) * D + c] = s;
    }
}

__global__ void k_sq_mean(double* __restrict__ mp, const double* __restrict__ mean, int d, SqConst k) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    for (int a = 0; a < k.n; ++a) {
        double s = 0.0;
        for (int b = 0; b < k.n; ++b) s += k.A1[a * SQN + b] * k.pinv[b] * mean[j * k.n + b];
        mp[j * k.n + a] = s;
    }
}

// out[i][j] = sum_{k >= i} R[i][k] Hraw[j][k] p[k % n]  (= (H Cl-)^T with Cl- = R^T, H = Hraw P): 32x32 tile per block
template <typename OT, typename RT>
__global__ __launch_bounds__(256) void k_rht(OT* __restrict__ out, long ldo, const RT* __restrict__ R, long ldr,
                                             const double* __restrict__ Hraw, int D, int m, SqConst kc) {
    __shared__ double sA[32][33], sH[32][33];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4, wr = w >> 1, wc = w & 1;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tx = t & 31, ty = t >> 5;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = i0; k0 < D; k0 += 32) {
        const int k = k0 + tx;
        const double pk = kc.p[k % kc.n];
        for (int r = ty; r < 32; r += 8) {
            const int i = i0 + r, j = j0 + r;
            sA[r][tx] = (i < D && k < D && k >= i) ? (double)R[(long)i * ldr + k] : 0.0;
            sH[r][tx] = (j < m && k < D) ? Hraw[(long)j * D + k] * pk : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 8; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[wr * 16 + fr][4 * st + fk], sH[wc * 16 + fr][4 * st + fk], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 16 + fk + 4 * r, j = j0 + wc * 16 + fr;
        if (i < D && j < m) out[(long)i * ldo + j] = (OT)acc[r];
    }
}

// W2[i][c0 + k] = R[i][k] (upper triangle, zero below)
template <typename WT>
__global__ void k_sq_fill_r(WT* __restrict__ W2, long ld2, int c0, const WT* __restrict__ R, long ldr, int D) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (k >= D || i >= D) return;
    W2[(long)i * ld2 + c0 + k] = (k >= i) ? R[(long)i * ldr + k] : WT(0);
}

// out[i][j] = sum_k T1[k][i] Hraw[j][k] p[k % n]  (= (H T1)^T, H = Hraw P, T1 = A Pinv Cl lower triangular up to the n x n
// point blocks when Cl is (`tri`): T1[k][i] = 0 for i >= k + n, so k starts a tile before i0): 32x32 tile per block
template <typename OT>
__global__ __launch_bounds__(256) void k_sq_tht(OT* __restrict__ out, long ldo, const double* __restrict__ T1,
                                                const double* __restrict__ Hraw, int D, int m, SqConst kc, int tri) {
    __shared__ double sA[32][33], sH[32][33];
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, fr = lane & 15, fk = lane >> 4, wr = w >> 1, wc = w & 1;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, tx = t & 31, ty = t >> 5;
    d4 acc = {0, 0, 0, 0};
    for (int k0 = (tri && i0 >= 32) ? i0 - 32 : 0; k0 < D; k0 += 32) {
        for (int r = ty; r < 32; r += 8) {
            const int k = k0 + r, i = i0 + tx, j = j0 + r, kh = k0 + tx;
            sA[r][tx] = (k < D && i < D) ? T1[(long)k * D + i] : 0.0;                          // sA[k][i]
            sH[r][tx] = (j < m && kh < D) ? Hraw[(long)j * D + kh] * kc.p[kh % kc.n] : 0.0;   // sH[j][k]
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < 8; ++st)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[4 * st + fk][wr * 16 + fr], sH[wc * 16 + fr][4 * st + fk], acc, 0, 0, 0);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 16 + fk + 4 * r, j = j0 + wc * 16 + fr;
        if (i < D && j < m) out[(long)i * ldo + j] = (OT)acc[r];
    }
}

// The same two products when the rows of H are short (a finite-difference stencil plus the derivative entry: 4-5 entries
// of D; `ell` built by sq_upload_operator, width <= SQ_ELLW): gathers instead of dense m x D x D products.
//   k_sq_tht_ell: out[i][j] = sum_e T1[c_e(j)][i] v_e(j) p[c_e(j) % n] -- a 32 x 32 tile per block, T1 rows read along i,
//                 the tile turned through LDS so that out is written along j
//   k_rht_ell:    out[i][j] = sum_{e: c_e(j) >= i} R[i][c_e(j)] v_e(j) p[c_e(j) % n] -- one row i of R per block
constexpr int SQ_ELLW = 8;
struct SqEll {
    const int* col;      // [e * m + j], -1: none
    const double* val;   // [e * m + j]
    int w;
};

template <typename OT>
__global__ __launch_bounds__(256) void k_sq_tht_ell(OT* __restrict__ out, long ldo, const double* __restrict__ T1, SqEll ell,
                                                    int D, int m, SqConst kc) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32, i = i0 + tx;
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r;
        double s = 0.0;
        if (j < m && i < D)
            for (int e = 0; e < ell.w; ++e) {
                const int c = ell.col[e * m + j];
                if (c >= 0) s += T1[(long)c * D + i] * (ell.val[e * m + j] * kc.p[c % kc.n]);
            }
        tile[r][tx] = s;   // tile[j][i]
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ii = i0 + r, j = j0 + tx;
        if (ii < D && j < m) out[(long)ii * ldo + j] = (OT)tile[tx][r];
    }
}

template <typename OT, typename RT>
__global__ __launch_bounds__(256) void k_rht_ell(OT* __restrict__ out, long ldo, const RT* __restrict__ R, long ldr, SqEll ell,
                                                 int D, int m, SqConst kc) {
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < m; j += 256) {
        double s = 0.0;
        for (int e = 0; e < ell.w; ++e) {
            const int c = ell.col[e * m + j];
            if (c >= i) s += (double)R[(long)i * ldr + c] * (ell.val[e * m + j] * kc.p[c % kc.n]);
        }
        out[(long)i * ldo + j] = (OT)s;
    }
}

// dst[i][c] = src[i][c] for c >= i, 0 below the diagonal (what a finished QR leaves below R is not R)
template <typename WT>
__global__ void k_sq_copy_upper(WT* __restrict__ dst, long ldd, const WT* __restrict__ src, long lds, int n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (c >= n || i >= n) return;
    dst[(long)i * ldd + c] = (c >= i) ? src[(long)i * lds + c] : WT(0);
}

// z = Hraw P mp + shift (white.py:169-186): one wave per row
__global__ __launch_bounds__(256) void k_sq_gemv_z(double* __restrict__ z, const double* __restrict__ Hraw,
                                                   const double* __restrict__ mp, const double* __restrict__ shift,
                                                   int m, int D, SqConst kc) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= m) return;
    double s = 0.0;
    for (int k = lane; k < D; k += 64) s += Hraw[(long)j * D + k] * kc.p[k % kc.n] * mp[k];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) z[j] = s + shift[j];
}

// y = R1^-T z and x = R1c^-1 z (R1c = R1 with rows flipped to a positive diagonal; white.py:125 solves with Sl^T = R1),
// R1 = W[0:m, 0:m] upper.  norms = {|y|^2, |x|^2}.  Two blocks -- the two solves do not depend on each other: block 0 the
// forward one, block 1 the backward one --; 32-wide diagonal solves by one wave in registers.
template <typename WT>
__global__ __launch_bounds__(1024) void k_sq_trsv(const WT* __restrict__ W, long ld, int m,
                                                  const double* __restrict__ z, double* __restrict__ y,
                                                  double* __restrict__ x, double* __restrict__ norms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qr_lds_raw[];
    const int mb = (m + 31) / 32, mp = mb * 32;
    double* sv = reinterpret_cast<double*>(qr_lds_raw);   // running right-hand side
    double* ss = sv + mp;                                  // solution
    double* tile = ss + mp;                                // 32 x 33
    double* red = tile + 32 * 33;                          // 32
    const int t = threadIdx.x, lane = t & 63;
    const bool fwd = blockIdx.x == 0;
    double nrm = 0.0;
    if (fwd) {
        // ---- forward: (R1^T) y = z, right-looking
        for (int i = t; i < mp; i += 1024) sv[i] = i < m ? z[i] : 0.0;
        for (int b = 0; b < mb; ++b) {
            __syncthreads();
            {
                const int r = t >> 5, c = t & 31, gi = b * 32 + r, gj = b * 32 + c;
                tile[r * 33 + c] = (gi < m && gj < m) ? (double)W[(long)gi * ld + gj] : (gi == gj ? 1.0 : 0.0);
            }
            __syncthreads();
            if (t < 64) {
                const int i = lane & 31;
                double v = sv[b * 32 + i];
                for (int k = 0; k < 32; ++k) {
                    const double yk = __shfl(v, k) / tile[k * 33 + k];
                    if (i > k) v -= tile[k * 33 + i] * yk;
                    if (i == k) v = yk;
                }
                if (lane < 32) ss[b * 32 + i] = v;
            }
            __syncthreads();
            for (int i = (b + 1) * 32 + t; i < m; i += 1024) {
                double a = 0.0;
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const int gk = b * 32 + k;
                    if (gk < m) a += (double)W[(long)gk * ld + i] * ss[gk];
                }
                sv[i] -= a;
            }
        }
        __syncthreads();
        for (int i = t; i < m; i += 1024) {
            y[i] = ss[i];
            nrm += ss[i] * ss[i];
        }
    } else {
        // ---- backward: R1 x = S z, left-looking (row dot products)
        for (int i = t; i < mp; i += 1024) ss[i] = 0.0;
        __syncthreads();
        for (int b = mb - 1; b >= 0; --b) {
            const int r = t >> 5, l32 = t & 31, gi = b * 32 + r;
            double a = 0.0;
            if (gi < m)
                for (int k = (b + 1) * 32 + l32; k < m; k += 32) a += (double)W[(long)gi * ld + k] * ss[k];
            for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (l32 == 0) {
                double rhs = 0.0;
                if (gi < m) rhs = (W[(long)gi * ld + gi] < WT(0) ? -z[gi] : z[gi]) - a;
                red[r] = rhs;
            }
            {
                const int c = t & 31, gj = b * 32 + c;
                tile[r * 33 + c] = (gi < m && gj < m) ? (double)W[(long)gi * ld + gj] : (gi == gj ? 1.0 : 0.0);
            }
            __syncthreads();
            if (t < 64) {
                const int i = lane & 31;
                double v = red[i];
                for (int k = 31; k >= 0; --k) {
                    const double xk = __shfl(v, k) / tile[k * 33 + k];
                    if (i < k) v -= tile[i * 33 + k] * xk;
                    if (i == k) v = xk;
                }
                if (lane < 32) ss[b * 32 + i] = v;
            }
            __syncthreads();
        }
        for (int i = t; i < m; i += 1024) {
            x[i] = ss[i];
            nrm += ss[i] * ss[i];
        }
    }
    // block reduction of the norm
    for (int o = 32; o > 0; o >>= 1) nrm += __shfl_xor(nrm, o);
    __syncthreads();
    if (lane == 0) tile[t >> 6] = nrm;
    __syncthreads();
    if (t == 0) {
        double a = 0.0;
        for (int q = 0; q < 16; ++q) a += tile[q];
        norms[fwd ? 0 : 1] = a;
    }
}

// mean[c] = p[c % n] (mp[c] - sum_i R2[i][c] y[i]),  R2 = W2[0:m, c0:c0+D]  (m_new = mp - K z, K z = R2^T R1^-T z)
template <typename WT>
__global__ __launch_bounds__(256) void k_sq_mean_update(double* __restrict__ mean, const double* __restrict__ mp,
                                                        const WT* __restrict__ W2, long ld2, int c0,
                                                        const double* __restrict__ y, int m, int D, SqConst kc) {
    __shared__ double part[4][64];
    const int tx = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (c < D)
        for (int i = g; i < m; i += 4) s += (double)W2[(long)i * ld2 + c0 + c] * y[i];
    part[g][tx] = s;
    __syncthreads();
    if (g == 0 && c < D) mean[c] = kc.p[c % kc.n] * (mp[c] - (part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]));
}

// Cl[r][c] = p[r % n] s_c R3[c][r] for c <= r, 0 above: the new factor P R3^T with a non-negative diagonal
// (R3 = W2[c0 + c][c0 + r]); tile transpose through LDS
template <typename WT>
__global__ __launch_bounds__(256) void k_sq_state_out(double* __restrict__ Cl, int D, const WT* __restrict__ W2,
                                                      long ld2, int c0, SqConst kc) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int cb = blockIdx.x * 32, rb = blockIdx.y * 32;   // tile of Cl: rows rb.., cols cb..
    for (int q = ty; q < 32; q += 8) {
        const int c = cb + q, r = rb + tx;                   // read R3[c][r], contiguous in r
        double v = 0.0;
        if (c < D && r < D && c <= r) {
            v = (double)W2[(long)(c0 + c) * ld2 + c0 + r];
            if (W2[(long)(c0 + c) * ld2 + c0 + c] < WT(0)) v = -v;
        }
        tile[q][tx] = v;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int r = rb + q, c = cb + tx;
        if (r < D && c < D) Cl[(long)r * D + c] = kc.p[r % kc.n] * tile[tx][q];
    }
}

// per-step read-out (figure1.py:76-80): mean of derivative 0 and sqrt(diag(Cl Cl^T)) at it; one wave per mesh point
__global__ __launch_bounds__(256) void k_sq_readout(double* __restrict__ means, double* __restrict__ stds,
                                                    const double* __restrict__ mean, const double* __restrict__ Cl,
                                                    int d, int n, int D) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= d) return;
    double s = 0.0;
    const double* row = Cl + (long)(j * n) * D;
    for (int c = lane; c < D; c += 64) s += row[c] * row[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        means[j] = mean[j * n];
        stds[j] = sqrt(s);
    }
}

// diag(Sq)[j] = sum_{k <= j} Rq[k][j]^2, Rq = W[0:m, 0:m] upper (white.py:160: sqrt(diag(S)))
template <typename WT>
__global__ void k_sq_coldiag(double* __restrict__ out, const WT* __restrict__ W, long ld, int m) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    double s = 0.0;
    for (int k = 0; k <= j; ++k) s += (double)W[(long)k * ld + j] * (double)W[(long)k * ld + j];
    out[j] = s;
}

}  // namespace

struct pnmol_sqrt_filter {
    pnmol_ctx* ctx = nullptr;
    int d = 0, ds = 0, n = 0, nu = 0, nB = 0, m = 0, D = 0;   // ds = state components (d, or 2d: latent-force model), D = n ds
    double A1[SQN * SQN] = {0};
    int f32 = 0;                // pnmol_filter_desc.dtype = 1: the QR work matrices, the QR itself and Rc in fp32
    double *Hraw = nullptr, *shift = nullptr, *EtT = nullptr, *QlT = nullptr;
    void *EtT_w = nullptr, *QlT_w = nullptr;   // the same two in the work matrices' element type (fp64: the arrays above)
    int* ell_col = nullptr;     // ELL image of Hraw, [SQ_ELLW * m]; ell_w = 0: some row of H is longer, the dense kernels run
    double* ell_val = nullptr;
    int ell_w = 0;
    double *mean = nullptr, *Cl = nullptr, *T1 = nullptr, *mp = nullptr, *z = nullptr, *y = nullptr, *x = nullptr,
           *norms = nullptr;
    double t = 0.0;
    QrPlan q4;                  // one-QR step: [[T1^T H^T, T1^T], [Rc]] (see sq_step)
    void* Rc = nullptr;          // R of the step-invariant rows [[Ql^T H^T, Ql^T], [E^T, 0]] for (rc_dt, rc_op), element type of q2
    double rc_dt = -1.0, prev_dt = -1.0;
    long opver = 0, rc_op = -1, prev_op = -1;
    QrPlan q1, q2, q3;          // q3: [(H Ql)^T; E^T] of estimate_error (white.py:153-162), factor kept in place
    double *sqdiag = nullptr, *yq = nullptr, *xq = nullptr, *normsq = nullptr;
    double err_dt = -1.0;        // the dt q3 holds the error model of (-1: none); reset by set_operator
    bool cl_tri = false;        // the resident factor is lower triangular: the predict QR can use the structured member lists
    std::vector<double> hB;     // boundary rows, kept for set_operator
    float last_ms = -1.f;
};

namespace {

double sq_nordsieck(int nu, int a, double dt) {   // base/iwp.py:55-62
    double fact = 1.0;
    for (int q = 2; q <= nu - a; ++q) fact *= q;
    return std::pow(std::fabs(dt), nu - a + 0.5) / fact;
}

SqConst sq_const(const pnmol_sqrt_filter* f, double dt) {
    SqConst k;
    std::memset(&k, 0, sizeof(k));
    std::memcpy(k.A1, f->A1, sizeof(k.A1));
    k.n = f->n;
    for (int a = 0; a < f->n; ++a) {
        k.p[a] = sq_nordsieck(f->nu, a, dt);
        k.pinv[a] = 1.0 / k.p[a];
    }
    return k;
}

// Hraw = [E1 - M E0; B E0] (white.py:169-186 with E0 P, E1 P factored into the per-step column scaling) and shift
int sq_upload_operator(pnmol_sqrt_filter* f, const double* M, const double* shift_d) {
    const int d = f->d, ds = f->ds, n = f->n, m = f->m, D = f->D, nB = f->nB;
    std::vector<double> H((size_t)m * D, 0.0), sh(m, 0.0);
    for (int i = 0; i < d; ++i) {   // M is (d, ds): [J_x + L] or, latent-force model, [J_x + L, I] (latent.py:253-257)
        for (int j = 0; j < ds; ++j) H[(size_t)i * D + (size_t)j * n] = -M[(size_t)i * ds + j];
        H[(size_t)i * D + (size_t)i * n + 1] += 1.0;
        if (shift_d) sh[i] = shift_d[i];
    }
    for (int i = 0; i < nB; ++i)
        for (int j = 0; j < ds; ++j) H[(size_t)(d + i) * D + (size_t)j * n] = f->hB[(size_t)i * ds + j];
    pnmol_ctx* ctx = f->ctx;
    ++f->opver;
    {   // short rows (PNMOL_SQRT_ELL=0: always the dense products)
        std::vector<int> ecol((size_t)SQ_ELLW * m, -1);
        std::vector<double> eval((size_t)SQ_ELLW * m, 0.0);
        int w = 0;
        const bool off = std::getenv("PNMOL_SQRT_ELL") && std::atoi(std::getenv("PNMOL_SQRT_ELL")) == 0;
        for (int i = 0; i < m && w <= SQ_ELLW && !off; ++i) {
            int e = 0;
            for (int k = 0; k < D; ++k) {
                const double v = H[(size_t)i * D + k];
                if (v == 0.0) continue;
                if (e < SQ_ELLW) ecol[(size_t)e * m + i] = k, eval[(size_t)e * m + i] = v;
                ++e;
            }
            w = std::max(w, e);
        }
        f->ell_w = (off || w > SQ_ELLW) ? 0 : w;
        if (f->ell_w) {
            QCHECK(ctx, hipMemcpyAsync(f->ell_col, ecol.data(), sizeof(int) * ecol.size(), hipMemcpyHostToDevice, ctx->stream));
            QCHECK(ctx, hipMemcpyAsync(f->ell_val, eval.data(), sizeof(double) * eval.size(), hipMemcpyHostToDevice, ctx->stream));
            QCHECK(ctx, hipStreamSynchronize(ctx->stream));   // (the two host vectors go out of scope)
        }
    }
    QCHECK(ctx, hipMemcpyAsync(f->Hraw, H.data(), sizeof(double) * H.size(), hipMemcpyHostToDevice, ctx->stream));
    QCHECK(ctx, hipMemcpyAsync(f->shift, sh.data(), sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// out = (H X^T ...)^T products of the step: the gather kernels when H has an ELL image, the dense MFMA kernels otherwise
template <typename OT, typename RT>
void launch_rht(pnmol_sqrt_filter* f, OT* out, long ldo, const RT* R, long ldr, const SqConst& kc) {
    hipStream_t st = f->ctx->stream;
    if (f->ell_w)
        hipLaunchKernelGGL((k_rht_ell<OT, RT>), dim3(f->D), dim3(256), 0, st, out, ldo, R, ldr, SqEll{f->ell_col, f->ell_val, f->ell_w},
                           f->D, f->m, kc);
    else
        hipLaunchKernelGGL((k_rht<OT, RT>), tiles(f->m, f->D), dim3(256), 0, st, out, ldo, R, ldr, f->Hraw, f->D, f->m, kc);
}
template <typename OT>
void launch_tht(pnmol_sqrt_filter* f, OT* out, long ldo, const SqConst& kc) {
    hipStream_t st = f->ctx->stream;
    if (f->ell_w)
        hipLaunchKernelGGL(k_sq_tht_ell<OT>, tiles(f->m, f->D), dim3(256), 0, st, out, ldo, f->T1, SqEll{f->ell_col, f->ell_val, f->ell_w},
                           f->D, f->m, kc);
    else
        hipLaunchKernelGGL(k_sq_tht<OT>, tiles(f->m, f->D), dim3(256), 0, st, out, ldo, f->T1, f->Hraw, f->D, f->m, kc,
                           f->cl_tri ? 1 : 0);
}

// rows x cols block copy between arrays of the work matrices' element type (device to device)
template <typename WT>
hipError_t copy_block(WT* dst, long ldd, const void* src, long lds, int rows, int cols, hipStream_t st) {
    return hipMemcpy2DAsync(dst, sizeof(WT) * ldd, src, sizeof(WT) * lds, sizeof(WT) * cols, rows, hipMemcpyDeviceToDevice, st);
}

template <typename WT>
int sq_step_t(pnmol_sqrt_filter* f, double dt, double* norms_out) {
    pnmol_ctx* ctx = f->ctx;
    hipStream_t st = ctx->stream;
    const int d = f->ds, m = f->m, D = f->D;   // d: state components here
    const SqConst kc = sq_const(f, dt);
    const QrPlan &q1 = f->q1, &q2 = f->q2;
    const int Dtop = (D + QB - 1) / QB * QB;   // lower blocks start on a row-block boundary
    const int mpad = (m + 31) / 32 * 32;
    const size_t trsv_lds = sizeof(double) * (2 * (size_t)mpad + 32 * 33 + 32 + 128);
    if (trsv_lds > 64 * 1024)   // beyond the default dynamic-LDS limit (m > ~3500: 2-D meshes); create() bounds it by 160 KB
        QCHECK(ctx, hipFuncSetAttribute((const void*)k_sq_trsv<WT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)trsv_lds));
    // mp = A Pinv m, T1 = A Pinv Cl, z = H mp + shift (white.py:101-104)
    hipLaunchKernelGGL(k_sq_mean, dim3((d + 255) / 256), dim3(256), 0, st, f->mp, f->mean, d, kc);
    hipLaunchKernelGGL(k_sq_rows, dim3((D + 255) / 256, d), dim3(256), 0, st, f->T1, f->Cl, d, D, kc);
    hipLaunchKernelGGL(k_sq_gemv_z, dim3((m + 3) / 4), dim3(256), 0, st, f->z, f->Hraw, f->mp, f->shift, m, D, kc);
    if (f->err_dt == dt)   // estimate_error: sigma^2 = z^T Sq^-1 z / m = |Rq^-T z|^2 / m (white.py:159)
        hipLaunchKernelGGL(k_sq_trsv<WT>, dim3(2), dim3(1024), trsv_lds, st, f->q3.w<WT>(), (long)f->q3.ld, m, f->z, f->yq,
                           f->xq, f->normsq);

    // The reference's two QRs (white.py:114, :120) are one: with [(A Pinv Cl)^T; Ql^T] = Q Rp, the update's pre-array
    // [[Rp H^T, Rp], [E^T, 0]] and  B = [[T1^T H^T, T1^T], [Ql^T H^T, Ql^T], [E^T, 0]]  differ by an orthogonal factor
    // from the left, so they have the same R.  The last two block rows of B do not depend on the state: for a linear PDE
    // under a constant step they are factored ONCE (Rc, upper triangular), and a step is the QR of [[T1^T H^T, T1^T], [Rc]]
    // -- D dense rows on a triangle, 49 members per panel at N=512 instead of up to 113, two tree levels throughout.
    // Taken from the second consecutive step with the same (dt, operator) on; before that (and for a semilinear PDE,
    // whose operator changes every step) the two QRs run as written.
    const bool repeat = (f->prev_dt == dt && f->prev_op == f->opver);
    f->prev_dt = dt, f->prev_op = f->opver;
    const bool disabled = std::getenv("PNMOL_SQRT_ONE_QR") && std::atoi(std::getenv("PNMOL_SQRT_ONE_QR")) == 0;
    WT* Rc = static_cast<WT*>(f->Rc);
    if (repeat && !disabled && !(f->rc_dt == dt && f->rc_op == f->opver)) {
        QCHECK(ctx, hipMemsetAsync(q2.W, 0, q2.bytes(), st));
        launch_rht<WT, double>(f, q2.w<WT>(), (long)q2.ld, f->QlT, (long)D, kc);
        QCHECK(ctx, copy_block<WT>(q2.w<WT>() + m, q2.ld, f->QlT_w, D, D, D, st));
        QCHECK(ctx, copy_block<WT>(q2.w<WT>() + (long)D * q2.ld, q2.ld, f->EtT_w, m, m, m, st));
        if (int rc = qr_inplace(ctx, q2)) return rc;
        hipLaunchKernelGGL(k_sq_copy_upper<WT>, dim3((q2.ld + 31) / 32, (q2.ld + 7) / 8), dim3(32, 8), 0, st, Rc,
                           (long)q2.ld, q2.w<WT>(), (long)q2.ld, q2.ld);
        f->rc_dt = dt, f->rc_op = f->opver;
    }
    const QrPlan* qr = &q2;   // where R = [[R1, R2], [0, R3]] ends up
    if (f->rc_dt == dt && f->rc_op == f->opver && !disabled) {
        const QrPlan& q4 = f->q4;
        QCHECK(ctx, hipMemsetAsync(q4.W, 0, q4.bytes(), st));
        launch_tht<WT>(f, q4.w<WT>(), (long)q4.ld, kc);
        hipLaunchKernelGGL(k_copy_t<WT>, tiles(D, D), dim3(256), 0, st, q4.w<WT>() + m, (long)q4.ld, f->T1, (long)D, D, D);
        QCHECK(ctx, copy_block<WT>(q4.w<WT>() + (long)Dtop * q4.ld, q4.ld, Rc, q2.ld, q2.ld, q2.ld, st));
        if (int rc = qr_inplace(ctx, q4, 0, Dtop / QB)) return rc;
        qr = &q4;
    } else {
        // predict (white.py:114): Cl- = R^T of [(A Pinv Cl)^T; Ql^T]
        QCHECK(ctx, hipMemsetAsync(q1.W, 0, q1.bytes(), st));
        hipLaunchKernelGGL(k_copy_t<WT>, tiles(D, D), dim3(256), 0, st, q1.w<WT>(), (long)q1.ld, f->T1, (long)D, D, D);
        QCHECK(ctx, copy_block<WT>(q1.w<WT>() + (long)Dtop * q1.ld, q1.ld, f->QlT_w, D, D, D, st));
        // (A Pinv Cl)^T is upper triangular up to the n x n point blocks when Cl is lower triangular, Ql^T exactly
        if (int rc = qr_inplace(ctx, q1, f->cl_tri ? Dtop / QB : 0)) return rc;
        // update (white.py:120-123): QR of [[R H^T, R], [E^T, 0]]
        QCHECK(ctx, hipMemsetAsync(q2.W, 0, q2.bytes(), st));
        launch_rht<WT, WT>(f, q2.w<WT>(), (long)q2.ld, q1.w<WT>(), (long)q1.ld, kc);
        hipLaunchKernelGGL(k_sq_fill_r<WT>, dim3((D + 31) / 32, (D + 7) / 8), dim3(32, 8), 0, st, q2.w<WT>(), (long)q2.ld, m,
                           q1.w<WT>(), (long)q1.ld, D);
        QCHECK(ctx, copy_block<WT>(q2.w<WT>() + (long)D * q2.ld, q2.ld, f->EtT_w, m, m, m, st));
        if (int rc = qr_inplace(ctx, q2)) return rc;
    }
    hipLaunchKernelGGL(k_sq_trsv<WT>, dim3(2), dim3(1024), trsv_lds, st, qr->w<WT>(), (long)qr->ld, m, f->z, f->y, f->x,
                       norms_out);
    hipLaunchKernelGGL(k_sq_mean_update<WT>, dim3((D + 63) / 64), dim3(256), 0, st, f->mean, f->mp, qr->w<WT>(), (long)qr->ld,
                       m, f->y, m, D, kc);
    hipLaunchKernelGGL(k_sq_state_out<WT>, tiles(D, D), dim3(256), 0, st, f->Cl, D, qr->w<WT>(), (long)qr->ld, m, kc);
    QCHECK(ctx, hipGetLastError());
    f->t += dt;
    f->cl_tri = true;   // P R3^T
    return 0;
}

int sq_step(pnmol_sqrt_filter* f, double dt, double* norms_out) {
    return f->f32 ? sq_step_t<float>(f, dt, norms_out) : sq_step_t<double>(f, dt, norms_out);
}

template <typename WT>
int sq_error_model_t(pnmol_sqrt_filter* f, double dt) {
    pnmol_ctx* ctx = f->ctx;
    hipStream_t st = ctx->stream;
    const int m = f->m, D = f->D;
    const SqConst kc = sq_const(f, dt);
    const QrPlan& q3 = f->q3;
    // Sq = H (Ql Ql^T) H^T + E E^T (white.py:156-158) = R^T R with R of [(H Ql)^T; E^T]; (H Ql)^T = Ql^T H^T, Ql^T upper
    QCHECK(ctx, hipMemsetAsync(q3.W, 0, q3.bytes(), st));
    launch_rht<WT, double>(f, q3.w<WT>(), (long)q3.ld, f->QlT, (long)D, kc);
    QCHECK(ctx, copy_block<WT>(q3.w<WT>() + (long)D * q3.ld, q3.ld, f->EtT_w, m, m, m, st));
    if (int rc = qr_inplace(ctx, q3)) return rc;
    hipLaunchKernelGGL(k_sq_coldiag<WT>, dim3((m + 255) / 256), dim3(256), 0, st, f->sqdiag, q3.w<WT>(), (long)q3.ld, m);
    QCHECK(ctx, hipGetLastError());
    f->err_dt = dt;
    return 0;
}

void sq_fill_out(pnmol_step_out* o, double t, const double* norms, int m) {
    o->t_new = t;
    o->sigma2_whitened = norms[0] / m;
    o->diffusion_squared_local = norms[1] / m;
    o->error_sigma2 = std::nan("");
    o->info = (std::isfinite(norms[0]) && std::isfinite(norms[1])) ? -1 : 0;
}

}  // namespace

extern "C" {

int pnmol_sqrt_filter_destroy(pnmol_sqrt_filter* f) {
    if (!f) return -1;
    f->ctx->children.fetch_sub(1);  // (lifetime rule, include/pnmol_hip.h: pnmol_ctx_destroy refuses while filters live)
    hipSetDevice(f->ctx->device);
    for (double* p : {f->Hraw, f->shift, f->EtT, f->QlT, f->mean, f->Cl, f->T1, f->mp, f->z, f->y, f->x, f->norms,
                      f->sqdiag, f->yq, f->xq, f->normsq})
        if (p) hipFree(p);
    if (f->Rc) hipFree(f->Rc);
    if (f->ell_col) hipFree(f->ell_col);
    if (f->ell_val) hipFree(f->ell_val);
    if (f->f32) {   // (fp64: aliases of EtT / QlT)
        if (f->EtT_w) hipFree(f->EtT_w);
        if (f->QlT_w) hipFree(f->QlT_w);
    }
    qr_plan_free(&f->q1);
    qr_plan_free(&f->q2);
    qr_plan_free(&f->q3);
    qr_plan_free(&f->q4);
    delete f;
    return 0;
}

int pnmol_sqrt_filter_create(pnmol_ctx* ctx, const pnmol_filter_desc* desc, pnmol_sqrt_filter** out) {
    if (!ctx || !desc || !out) return -1;
    *out = nullptr;
    const int d = desc->d, nu = desc->num_derivatives, n = nu + 1, nB = desc->nB;
    if (d <= 0 || nu < 1 || n > SQN || nB < 0 || !desc->L || !desc->E_sqrtm || !desc->Gamma || (nB > 0 && (!desc->B || !desc->R_sqrtm))) {
        ctx->err = "pnmol_sqrt_filter_create: bad descriptor";
        return -1;
    }
    const int ds = desc->d_state ? desc->d_state : d;
    if (ds != d && ds != 2 * d) {
        ctx->err = "pnmol_sqrt_filter_create: d_state must be 0, d (white-noise model) or 2d (latent-force model)";
        return -1;
    }
    if (desc->dtype != 0 && desc->dtype != 1) {
        ctx->err = "pnmol_sqrt_filter_create: dtype must be 0 (fp64) or 1 (fp32 QR)";
        return -1;
    }
    const int f32 = desc->dtype;
    QCHECK(ctx, hipSetDevice(ctx->device));
    if (int rc = qr_configure(ctx)) return rc;
    const int m = d + nB, D = n * ds;
    if (sizeof(double) * (2 * (size_t)((m + 31) / 32 * 32) + 32 * 33 + 160) > 160 * 1024) {
        ctx->err = "pnmol_sqrt_filter_create: m too large for the single-block triangular solve";
        return -1;
    }
    pnmol_sqrt_filter* f = new pnmol_sqrt_filter();
    ctx->children.fetch_add(1);
    f->ctx = ctx, f->d = d, f->ds = ds, f->n = n, f->nu = nu, f->nB = nB, f->m = m, f->D = D, f->f32 = f32;
    if (nB) f->hB.assign(desc->B, desc->B + (size_t)nB * ds);
    double Q1[SQN * SQN] = {0}, Lq[SQN * SQN] = {0};
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {   // base/iwp.py:13-30: flip both axes of pascal (lower) and hilbert
            const int ia = n - 1 - a, ib = n - 1 - b;
            double binom = 0.0;
            if (ib <= ia) {
                binom = 1.0;
                for (int q = 1; q <= ib; ++q) binom = binom * (ia - ib + q) / q;
            }
            f->A1[a * SQN + b] = binom;
            Q1[a * SQN + b] = 1.0 / (ia + ib + 1.0);
        }
    for (int j = 0; j < n; ++j)      // Lq = chol(Q1), n <= 4
        for (int i = j; i < n; ++i) {
            double s = Q1[i * SQN + j];
            for (int k = 0; k < j; ++k) s -= Lq[i * SQN + k] * Lq[j * SQN + k];
            Lq[i * SQN + j] = (i == j) ? std::sqrt(s) : s / Lq[j * SQN + j];
        }
    int rc = 0;
    do {
        auto alloc = [&](double** p, size_t count) { return hipMalloc(p, sizeof(double) * count) == hipSuccess; };
        if (!alloc(&f->Hraw, (size_t)m * D) || !alloc(&f->shift, m) || !alloc(&f->EtT, (size_t)m * m) ||
            !alloc(&f->QlT, (size_t)D * D) || !alloc(&f->mean, D) || !alloc(&f->Cl, (size_t)D * D) ||
            !alloc(&f->T1, (size_t)D * D) || !alloc(&f->mp, D) || !alloc(&f->z, m) || !alloc(&f->y, m) ||
            !alloc(&f->x, m) || !alloc(&f->norms, 2) || !alloc(&f->sqdiag, m) || !alloc(&f->yq, m) ||
            !alloc(&f->xq, m) || !alloc(&f->normsq, 2) || !alloc(&f->ell_val, (size_t)SQ_ELLW * m) ||
            hipMalloc(&f->ell_col, sizeof(int) * (size_t)SQ_ELLW * m) != hipSuccess) { rc = -4; break; }
        if ((rc = qr_plan_alloc(ctx, (D + QB - 1) / QB * QB + D, D, &f->q1, f32))) { rc = -4; break; }
        if ((rc = qr_plan_alloc(ctx, D + m, m + D, &f->q2, f32))) { rc = -4; break; }
        if ((rc = qr_plan_alloc(ctx, D + m, m, &f->q3, f32))) { rc = -4; break; }
        if ((rc = qr_plan_alloc(ctx, (D + QB - 1) / QB * QB + f->q2.ld, m + D, &f->q4, f32))) { rc = -4; break; }
        if (hipMalloc(&f->Rc, f->q2.es() * (size_t)f->q2.ld * f->q2.ld) != hipSuccess) { rc = -4; break; }
        // Ql^T = (Gamma (x) Lq)^T (base/iwp.py:32-53), E^T = blockdiag(E_sqrtm, R_sqrtm)^T (white.py:184)
        std::vector<double> QlT((size_t)D * D, 0.0), EtT((size_t)m * m, 0.0);
        for (int j = 0; j < ds; ++j)
            for (int j2 = 0; j2 <= j; ++j2) {
                const double gjj = desc->Gamma[(size_t)j * ds + j2];
                if (gjj == 0.0) continue;
                for (int a = 0; a < n; ++a)
                    for (int b = 0; b <= a; ++b)
                        QlT[(size_t)(j2 * n + b) * D + (size_t)(j * n + a)] = gjj * Lq[a * SQN + b];
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) EtT[(size_t)j * m + i] = desc->E_sqrtm[(size_t)i * d + j];
        for (int i = 0; i < nB; ++i)
            for (int j = 0; j < nB; ++j) EtT[(size_t)(d + j) * m + d + i] = desc->R_sqrtm[(size_t)i * nB + j];
        if (hipMemcpy(f->QlT, QlT.data(), sizeof(double) * QlT.size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->EtT, EtT.data(), sizeof(double) * EtT.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = -2; break; }
        f->QlT_w = f->QlT, f->EtT_w = f->EtT;
        if (f32) {   // the two constant blocks as the fp32 work matrices take them
            f->QlT_w = f->EtT_w = nullptr;
            std::vector<float> q32(QlT.begin(), QlT.end()), e32(EtT.begin(), EtT.end());
            if (hipMalloc(&f->QlT_w, sizeof(float) * q32.size()) != hipSuccess ||
                hipMalloc(&f->EtT_w, sizeof(float) * e32.size()) != hipSuccess) { rc = -4; break; }
            if (hipMemcpy(f->QlT_w, q32.data(), sizeof(float) * q32.size(), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(f->EtT_w, e32.data(), sizeof(float) * e32.size(), hipMemcpyHostToDevice) != hipSuccess) { rc = -2; break; }
        }
        if ((rc = sq_upload_operator(f, desc->L, nullptr))) break;
    } while (0);
    if (rc) {
        if (rc == -2 && ctx->err.empty()) ctx->err = hipGetErrorString(hipGetLastError());
        pnmol_sqrt_filter_destroy(f);
        return rc;
    }
    *out = f;
    return 0;
}

int pnmol_sqrt_filter_set_operator(pnmol_sqrt_filter* f, const double* M_dd, const double* shift_d) {
    if (!f || !M_dd) return -1;
    QCHECK(f->ctx, hipSetDevice(f->ctx->device));
    f->err_dt = -1.0;
    return sq_upload_operator(f, M_dd, shift_d);
}

int pnmol_sqrt_filter_set_state(pnmol_sqrt_filter* f, double t, const double* mean_nd, const double* cov_sqrtm_DD) {
    if (!f || !mean_nd || !cov_sqrtm_DD) return -1;
    pnmol_ctx* ctx = f->ctx;
    QCHECK(ctx, hipSetDevice(ctx->device));
    std::vector<double> mv(f->D);
    for (int a = 0; a < f->n; ++a)
        for (int j = 0; j < f->ds; ++j) mv[(size_t)j * f->n + a] = mean_nd[(size_t)a * f->ds + j];   // reshape(-1, order="F")
    QCHECK(ctx, hipMemcpyAsync(f->mean, mv.data(), sizeof(double) * f->D, hipMemcpyHostToDevice, ctx->stream));
    QCHECK(ctx, hipMemcpyAsync(f->Cl, cov_sqrtm_DD, sizeof(double) * (size_t)f->D * f->D, hipMemcpyHostToDevice, ctx->stream));
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    f->t = t;
    bool tri = true;
    for (int i = 0; i < f->D && tri; ++i)
        for (int j = i + 1; j < f->D; ++j)
            if (cov_sqrtm_DD[(size_t)i * f->D + j] != 0.0) { tri = false; break; }
    f->cl_tri = tri;
    return 0;
}

int pnmol_sqrt_filter_get_state(pnmol_sqrt_filter* f, double* t, double* mean_nd, double* cov_sqrtm_DD) {
    if (!f) return -1;
    pnmol_ctx* ctx = f->ctx;
    QCHECK(ctx, hipSetDevice(ctx->device));
    if (t) *t = f->t;
    if (mean_nd) {
        std::vector<double> mv(f->D);
        QCHECK(ctx, hipMemcpyAsync(mv.data(), f->mean, sizeof(double) * f->D, hipMemcpyDeviceToHost, ctx->stream));
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
        for (int a = 0; a < f->n; ++a)
            for (int j = 0; j < f->ds; ++j) mean_nd[(size_t)a * f->ds + j] = mv[(size_t)j * f->n + a];
    }
    if (cov_sqrtm_DD) {
        QCHECK(ctx, hipMemcpyAsync(cov_sqrtm_DD, f->Cl, sizeof(double) * (size_t)f->D * f->D, hipMemcpyDeviceToHost, ctx->stream));
        QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int pnmol_sqrt_filter_predict_mean(pnmol_sqrt_filter* f, double dt, double* m_at_d) {
    if (!f || !m_at_d) return -1;
    pnmol_ctx* ctx = f->ctx;
    QCHECK(ctx, hipSetDevice(ctx->device));
    const SqConst kc = sq_const(f, dt);
    hipLaunchKernelGGL(k_sq_mean, dim3((f->ds + 255) / 256), dim3(256), 0, ctx->stream, f->mp, f->mean, f->ds, kc);
    std::vector<double> mv(f->D);
    QCHECK(ctx, hipMemcpyAsync(mv.data(), f->mp, sizeof(double) * f->D, hipMemcpyDeviceToHost, ctx->stream));
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int j = 0; j < f->d; ++j) m_at_d[j] = kc.p[0] * mv[(size_t)j * f->n];   // E0 P mp, white.py:192
    return 0;
}

int pnmol_sqrt_filter_prepare_error_model(pnmol_sqrt_filter* f, double dt) {
    if (!f || !(dt >= 0.0)) return -1;
    QCHECK(f->ctx, hipSetDevice(f->ctx->device));
    return f->f32 ? sq_error_model_t<float>(f, dt) : sq_error_model_t<double>(f, dt);
}

int pnmol_sqrt_filter_step(pnmol_sqrt_filter* f, double dt, pnmol_step_out* info, double* error_estimate_d) {
    if (!f || !(dt >= 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    QCHECK(ctx, hipSetDevice(ctx->device));
    const bool have_err = f->err_dt == dt;
    if (int rc = sq_step(f, dt, f->norms)) return rc;
    double nrm[2], nq[2] = {std::nan(""), std::nan("")};
    QCHECK(ctx, hipMemcpyAsync(nrm, f->norms, sizeof(nrm), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<double> sd;
    if (have_err) {
        QCHECK(ctx, hipMemcpyAsync(nq, f->normsq, sizeof(nq), hipMemcpyDeviceToHost, ctx->stream));
        if (error_estimate_d) {
            sd.resize(f->d);
            QCHECK(ctx, hipMemcpyAsync(sd.data(), f->sqdiag, sizeof(double) * f->d, hipMemcpyDeviceToHost, ctx->stream));
        }
    }
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (info) {
        sq_fill_out(info, f->t, nrm, f->m);
        info->error_sigma2 = nq[0] / f->m;
    }
    if (error_estimate_d)   // dt * sqrt(diag Sq) * sigma, PDE rows only (white.py:117-119, :129)
        for (int i = 0; i < f->d; ++i)
            error_estimate_d[i] = have_err ? dt * std::sqrt(sd[i]) * std::sqrt(nq[0] / f->m) : std::nan("");
    return 0;
}

int pnmol_sqrt_filter_steps(pnmol_sqrt_filter* f, int k, double dt, double* means_kd, double* stds_kd,
                            pnmol_step_out* info_k) {
    if (!f || k <= 0 || !(dt >= 0.0)) return -1;
    pnmol_ctx* ctx = f->ctx;
    QCHECK(ctx, hipSetDevice(ctx->device));
    const int w = f->ds;   // read-out width: every state component (u, and eps behind it in the latent-force model)
    DevBuf dm, dsd, dn;
    if (dm.alloc((size_t)k * w) || dsd.alloc((size_t)k * w) || dn.alloc((size_t)2 * k)) return -4;
    int rc = 0;
    {
        QrTimer tm(ctx->stream);   // HIP events around the loop; its destructor waits for the stop event
        for (int s = 0; s < k && !rc; ++s) {
            rc = sq_step(f, dt, dn.p + 2 * s);
            hipLaunchKernelGGL(k_sq_readout, dim3((w + 3) / 4), dim3(256), 0, ctx->stream, dm.p + (size_t)s * w,
                               dsd.p + (size_t)s * w, f->mean, f->Cl, w, f->n, f->D);
        }
        tm.stop();
    }
    f->last_ms = g_last_qr_ms;
    if (rc) return rc;
    std::vector<double> nrm((size_t)2 * k);
    QCHECK(ctx, hipMemcpyAsync(nrm.data(), dn.p, sizeof(double) * nrm.size(), hipMemcpyDeviceToHost, ctx->stream));
    if (means_kd) QCHECK(ctx, hipMemcpyAsync(means_kd, dm.p, sizeof(double) * (size_t)k * w, hipMemcpyDeviceToHost, ctx->stream));
    if (stds_kd) QCHECK(ctx, hipMemcpyAsync(stds_kd, dsd.p, sizeof(double) * (size_t)k * w, hipMemcpyDeviceToHost, ctx->stream));
    QCHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (info_k) {
        const double t0 = f->t - k * dt;
        for (int s = 0; s < k; ++s) sq_fill_out(info_k + s, t0 + (s + 1) * dt, nrm.data() + 2 * s, f->m);
    }
    return 0;
}

int pnmol_sqrt_filter_last_steps_ms(pnmol_sqrt_filter* f, float* ms) {
    if (!f || !ms) return -1;
    *ms = f->last_ms;
    return 0;
}

}  // extern "C"
