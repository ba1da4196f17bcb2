// Cost of a producer -> consumer hand-over between two workgroups through global memory (agent-scope
// release/acquire flag + an 8 KB tile), the primitive of a dataflow (persistent) Cholesky sweep.
// WG a writes tile[a], fences, sets flag; WG b polls, reads the tile, writes its own, ... ping-pong over `hops`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_pingpong(double* tiles, int* flags, int hops, int nwg, double* out,
                                                  long long* cyc) {
    const int me = blockIdx.x, tid = threadIdx.x;
    __shared__ double red[256];
    double acc = 0.0;
    long long t0 = 0;
    if (tid == 0) t0 = wall_clock64();
    for (int h = 0; h < hops; ++h) {
        const int owner = h % nwg;
        if (owner == me) {
            // consume the previous hop's tile (written by the other WG), produce mine
            if (h > 0) {
                if (tid == 0) {
                    int spins = 0;
                    while (__hip_atomic_load(&flags[h - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                           ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                }
                __syncthreads();
                const double* src = tiles + (long)((h - 1) % nwg) * 1024;
                for (int q = 0; q < 4; ++q) acc += __builtin_nontemporal_load(src + tid + 256 * q);
            }
            double* dst = tiles + (long)me * 1024;
            for (int q = 0; q < 4; ++q) dst[tid + 256 * q] = acc + q + h;
            __threadfence();
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&flags[h], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) {
        out[me] = red[0] + red[255];
        cyc[me] = wall_clock64() - t0;
    }
}
int main() {
    double *tiles, *out; int* flags; long long* cyc;
    const int hops = 2000;
    hipMalloc(&tiles, 1024 * 8 * 512); hipMalloc(&out, 8 * 512); hipMalloc(&flags, 4 * hops); hipMalloc(&cyc, 8 * 512);
    for (int nwg : {2, 8, 9, 64}) {
        hipMemset(flags, 0, 4 * hops);
        hipMemset(tiles, 0, 1024 * 8 * 512);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_pingpong<<<nwg, 256>>>(tiles, flags, hops, nwg, out, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double h0; hipMemcpy(&h0, out, 8, hipMemcpyDeviceToHost);
        std::printf("round-robin over %2d WGs: %.3f us per hop (flag + 8 KB tile), checksum %.3e  %s\n", nwg,
                    ms * 1e3 / hops, h0, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
