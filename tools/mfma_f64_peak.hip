// fp64 MFMA peak probe: every wave issues back-to-back v_mfma_f64_16x16x4_f64 on NACC independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_peak(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs, int wpb) {
    double* out; hipMalloc(&out, sizeof(double) * wgs * 256);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_peak<NACC><<<wgs, wpb * 64>>>(out, 10, 1.0, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0);
    k_peak<NACC><<<wgs, wpb * 64>>>(out, iters, 1.0, 1.0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)wgs * wpb * iters * NACC;
    std::printf("NACC=%d wgs=%d waves/wg=%d: %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz (assuming %d waves/SIMD)\n", NACC, wgs, wpb,
                mf * 2048 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (mf / (256.0 * 4)), wgs * wpb / 1024);
    hipFree(out);
}
int main() {
    run<1>(256, 4); run<4>(256, 4); run<9>(256, 4); run<9>(512, 4); run<9>(768, 4); run<16>(256, 4);
    return 0;
}
