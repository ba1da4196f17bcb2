// Cost of a producer -> consumer hand-over between two workgroups through global memory (flag + an 8 KB tile),
// the primitive of a dataflow (persistent) Cholesky sweep.  Protocols:
//   0 agent : release/acquire fences at agent scope (L2 write-back + invalidate)
//   1 xcd   : relaxed agent-scope flag (L1 bypass, L2 hit) + plain data; only valid when both workgroups share an
//             XCD (one L2)
//   2 wt    : data written with agent-scope (write-through, sc1) stores, relaxed flag carrying the producer's XCC id;
//             the consumer invalidates (buffer_inv sc1) only if it sits on another XCD
// Participating workgroups are `stride` apart in blockIdx (stride 8 = same XCD under the round-robin
// workgroup -> XCD dispatch; verified through XCC_ID).  Every consumer checks the tile contents.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }

template <int MODE>
__global__ __launch_bounds__(256) void k_pingpong(double* tiles, int* flags, int hops, int nwg, int stride,
                                                  int* bad, int* xcc) {
    if (blockIdx.x % stride != 0) return;
    const int me = blockIdx.x / stride, tid = threadIdx.x;
    if (me >= nwg) return;
    const int myx = xcc_id();
    if (tid == 0) xcc[me] = myx;
    __shared__ int fl;
    int nbad = 0;
    for (int h = me; h < hops; h += nwg) {
        if (h > 0) {
            if (tid == 0) {
                int spins = 0, f;
                while ((f = __hip_atomic_load(&flags[h - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 &&
                       ++spins < (1 << 22)) {}
                if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (MODE == 2 && f - 1 != myx) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                fl = f;
            }
            __syncthreads();
            const double* src = tiles + (long)((h - 1) % nwg) * 1024;
            for (int q = 0; q < 4; ++q) {
                // MODE >= 1: L1-bypassing loads (sc1): this CU's L1 may hold a stale copy of the line
                const double v = (MODE == 0 || MODE == 4) ? src[tid + 256 * q]
                                           : __hip_atomic_load(src + tid + 256 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nbad += v != (double)(h - 1) * 4096.0 + tid + 256 * q;
            }
        }
        double* dst = tiles + (long)me * 1024;
        for (int q = 0; q < 4; ++q) {
            const double v = (double)h * 4096.0 + tid + 256 * q;
            if (MODE >= 2) __hip_atomic_store(dst + tid + 256 * q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[tid + 256 * q] = v;
        }
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&flags[h], 1 + myx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    double* tiles; int *flags, *xcc, *bad;
    const int hops = 2000;
    hipMalloc(&tiles, 1024 * 8 * 512); hipMalloc(&bad, 4); hipMalloc(&flags, 4 * hops); hipMalloc(&xcc, 4 * 512);
    const char* names[5] = {"agent", "xcd", "wt", "wt-noinv", "wt-inv-plain"};
    for (int mode = 0; mode < 5; ++mode)
        for (int stride : {1, 8})
            for (int nwg : {2, 17}) {
                hipMemset(flags, 0, 4 * hops); hipMemset(bad, 0, 4);
                hipMemset(tiles, 0, 1024 * 8 * 512);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0);
                if (mode == 0) k_pingpong<0><<<nwg * stride, 256>>>(tiles, flags, hops, nwg, stride, bad, xcc);
                if (mode == 1) k_pingpong<1><<<nwg * stride, 256>>>(tiles, flags, hops, nwg, stride, bad, xcc);
                if (mode == 2) k_pingpong<2><<<nwg * stride, 256>>>(tiles, flags, hops, nwg, stride, bad, xcc);
                if (mode == 3) k_pingpong<3><<<nwg * stride, 256>>>(tiles, flags, hops, nwg, stride, bad, xcc);
                if (mode == 4) k_pingpong<4><<<nwg * stride, 256>>>(tiles, flags, hops, nwg, stride, bad, xcc);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                int hb, hx[32];
                hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                hipMemcpy(hx, xcc, 4 * nwg, hipMemcpyDeviceToHost);
                std::printf("%-12s stride %d, %2d WGs: %.3f us/hop  xcc ids %d %d %d  wrong values %d  %s\n", names[mode],
                            stride, nwg, ms * 1e3 / hops, hx[0], hx[1], hx[nwg - 1], hb,
                            hipGetErrorString(hipGetLastError()));
            }
    return 0;
}
