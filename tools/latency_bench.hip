// Dependent-chain latencies of the instructions on the Cholesky pivot chain (one wave, s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHAIN(name, body)                                                                    \
    __global__ void k_##name(double* out, unsigned long long* cyc, double x0, double c) {     \
        double x = x0 + threadIdx.x * 1e-12;                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                           \
        _Pragma("unroll 1") for (int it = 0; it < 64; ++it) {                                 \
            _Pragma("unroll") for (int k = 0; k < 16; ++k) { body; }                          \
        }                                                                                     \
        __builtin_amdgcn_s_waitcnt(0);                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                           \
        out[threadIdx.x] = x;                                                                 \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                               \
    }
CHAIN(fma, x = fma(x, c, 1e-3))
CHAIN(mul, x = x * c)
CHAIN(rsq, x = __builtin_amdgcn_rsq(x) + 1.0)
CHAIN(rsqonly, x = __builtin_amdgcn_rsq(x))
CHAIN(cnd, x = (x > 0.5) ? x * 1.0000001 : c)
CHAIN(readlane, { int lo = __builtin_amdgcn_readlane(__double2loint(x), 3); int hi = __builtin_amdgcn_readlane(__double2hiint(x), 3); x = fma(__hiloint2double(hi, lo), c, 1e-3); })
CHAIN(fma32, { float y = (float)x; y = fmaf(y, 1.0001f, 1e-3f); x = y; })
typedef double d4v __attribute__((ext_vector_type(4)));
// one wave, idle chip: dependent MFMA chain (through the accumulator), independent MFMAs back to back, and the round trip
// MFMA -> VALU -> MFMA operand (what one 4-column step of diag4_factor pays twice)
__global__ void k_mfma_dep(double* out, unsigned long long* cyc, double x0, double c) {
    d4v acc = {0, 0, 0, 0};
    const double a = x0 + threadIdx.x * 1e-12;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mfma_ind(double* out, unsigned long long* cyc, double x0, double c) {
    d4v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (d4v){0, 0, 0, 0};
    const double a = x0 + threadIdx.x * 1e-12;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c, acc[k & 7], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mfma_rt(double* out, unsigned long long* cyc, double x0, double c) {
    const d4v zero = {0, 0, 0, 0};
    double a = x0 + threadIdx.x * 1e-12;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const d4v r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c, zero, 0, 0, 0);
            a = r[0] * 1e-3;   // VALU op on the result, fed back as the next operand
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <class K> void run(const char* n, K k, double x0, double c, int extra) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
    k<<<1, 64>>>(out, cyc, x0, c); hipDeviceSynchronize();
    k<<<1, 64>>>(out, cyc, x0, c); hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    std::printf("%-10s %.1f cycles per iteration (%d dependent ops each)\n", n, h / 1024.0, extra);
}
int main() {
    run("fma_f64", k_fma, 1.0, 0.999, 1); run("mul_f64", k_mul, 1.0, 0.9999, 1); run("rsq+add", k_rsq, 2.0, 0.0, 2);
    run("rsq only", k_rsqonly, 2.0, 0.0, 1); run("cmp+cnd+mul", k_cnd, 1.0, 0.7, 3); run("rdlane+fma", k_readlane, 1.0, 0.999, 3);
    run("cvt+fmaf+cvt", k_fma32, 1.0, 0.0, 3);
    run("mfma dep", k_mfma_dep, 1.0, 1e-3, 1); run("mfma indep", k_mfma_ind, 1.0, 1e-3, 1); run("mfma+mul rt", k_mfma_rt, 1.0, 1e-3, 2);
    return 0;
}
