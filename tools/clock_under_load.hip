// Does a latency-bound instruction stream (dependent v_fma_f64 chain; dependent MFMA chain) on ONE workgroup slow down
// when the rest of the chip issues fp64 MFMAs?  (Is the chip-wide fp64-MFMA ceiling a clock/power throttle that everything
// pays, or an MFMA-issue limit?)  Block 0 measures; blocks 1.. run `bg`: 0 idle, 1 dependent MFMA chains, 2 VALU fma chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, long long* t, int iters, int bg, int* stop) {
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        double x = 1.0 + tid * 1e-9, y = 0.999999;
        d4 a = {0, 0, 0, 0};
        long long t0 = wall_clock64();
        long long c0 = clock64();
        for (int i = 0; i < iters; ++i) x = __builtin_fma(x, y, 1e-9);
        long long t1 = wall_clock64();
        long long c1 = clock64();
        for (int i = 0; i < iters / 8; ++i) a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0);
        long long t2 = wall_clock64();
        out[tid] = x + a[0];
        if (tid == 0) {
            t[0] = t1 - t0, t[1] = t2 - t1, t[2] = c1 - c0;
            __hip_atomic_store(stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if (!bg) return;
    double x = 1.0 + tid * 1e-9, y = 0.999999;
    d4 a = {0, 0, 0, 0};
    for (int it = 0; it < 4000; ++it) {
        if (bg == 1)
            for (int i = 0; i < 256; ++i) a = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a, 0, 0, 0);
        else
            for (int i = 0; i < 2048; ++i) x = __builtin_fma(x, y, 1e-9);
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    out[blockIdx.x * 256 + tid] = x + a[0];
}
int main() {
    double* out; long long* t; int* stop;
    hipMalloc(&out, 8 * 256 * 1024); hipMalloc(&t, 64); hipMalloc(&stop, 4);
    const int iters = 200000;
    for (int bg : {0, 1, 2, 1, 0})
        for (int grid : {64, 128, 256}) {
            hipMemset(stop, 0, 4);
            k<<<grid, 256>>>(out, t, iters, bg, stop);
            hipDeviceSynchronize();
            long long h[3];
            hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
            std::printf("bg %d grid %3d: fma chain %.2f ns/op (%.1f cycles at 2.4 GHz), mfma chain %.1f ns/op; s_memtime ticks per us: %.0f  %s\n",
                        bg, grid, h[0] * 10.0 / iters, h[0] * 10.0 / iters * 2.4, h[1] * 10.0 / (iters / 8), (double)h[2] / (h[0] * 0.01),
                        hipGetErrorString(hipGetLastError()));
            std::fflush(stdout);
        }
    return 0;
}
