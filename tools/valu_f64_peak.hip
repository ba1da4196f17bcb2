#include <hip/hip_runtime.h>
#include <cstdio>
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i * 1e-3;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wgs) {
    double* out; hipMalloc(&out, sizeof(double) * wgs * 256);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_fma<NACC><<<wgs, 256>>>(out, 10, 0.999, 1e-3); hipDeviceSynchronize();
    hipEventRecord(e0);
    k_fma<NACC><<<wgs, 256>>>(out, iters, 0.999, 1e-3);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)wgs * 256 * iters * NACC * 2;
    std::printf("v_fma_f64 NACC=%d wgs=%d: %.1f TFLOP/s\n", NACC, wgs, fl / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() { run<8>(256); run<16>(256); run<16>(512); run<16>(1024); run<32>(512); return 0; }
