// Producer -> consumer hand-over (flag + 8 KB tile) between workgroups of ONE XCD while the rest of the chip streams
// memory traffic: does a hand-over through the XCD's own L2 (plain stores, L1-bypassing loads) stay fast where the
// write-through path (sc1 stores, served at the memory side) queues behind the fabric traffic?
//   mode 0  wt   : 16-byte sc1 stores + sc1 flag; consumer: sc1 flag poll, 16-byte sc1 loads      (k_sweep_rl today)
//   mode 1  l2   : plain 16-byte stores + plain flag store; consumer: sc1 flag poll, sc1 16-byte loads (same XCD only)
//   mode 3  l2f  : mode 1 with the flag stored write-through (sc1)
//   mode 2  l2+wt: mode 1, and the tile is stored a second time write-through behind the flag (what a producer with
//                  readers on other XCDs would do); hop time as seen by the same-XCD consumer
// Ping-pong ring of `nwg` workgroups with blockIdx = 8 k (same XCD under round-robin dispatch, checked via XCC_ID);
// every other workgroup of the grid is background: `bg` = 0 idle, 1 = streams reads + write-through writes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// aux bits (gfx940): 1 = sc0, 2 = nt, 16 = sc1
template <int AUX>
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, unsigned off, double a, double b) {
    d2 v = {a, b};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ d2 ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ int ldflag(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void stflag(__amdgpu_buffer_rsrc_t r, unsigned off, int v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, AUX);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_ring(double* tiles, int* flags, int hops, int nwg, int* bad, int* xcc, int* stop,
                                              double* bgbuf, long bgn, int bg) {
    const int tid = threadIdx.x;
    constexpr int LAUX = 16;                   // loads / polls: sc1 (an sc0-only load may hit this CU's L1: never sees the flag)
    constexpr int SAUX = MODE == 0 ? 16 : 0;   // data stores
    constexpr int FAUX = (MODE == 0 || MODE == 3) ? 16 : 0;   // flag stores
    if (blockIdx.x % 8 != 0 || blockIdx.x / 8 >= nwg) {  // background
        if (!bg) return;
        const __amdgpu_buffer_rsrc_t rs = rsrc(stop);
        double acc = 0;
        const long per = bgn / gridDim.x / 2 * 2;
        const __amdgpu_buffer_rsrc_t rb = rsrc(bgbuf + blockIdx.x * per);  // (32-bit offsets: one window per workgroup)
        for (int it = 0; it < 4000; ++it) {
            for (long e = tid * 2; e < per; e += 512) {
                d2 v = ld16<0>(rb, (unsigned)(e * 8));
                acc += v[0] + v[1];
                if (bg == 2) st16<16>(rb, (unsigned)(e * 8), v[0] + 1.0, v[1]);
            }
            if (ldflag<16>(rs, 0)) break;
        }
        if (acc == 1.2345) bgbuf[0] = acc;
        return;
    }
    const int me = blockIdx.x / 8;
    const int myx = xcc_id();
    if (tid == 0) xcc[me] = myx;
    const __amdgpu_buffer_rsrc_t rt = rsrc(tiles), rf = rsrc(flags);
    int nbad = 0;
    __shared__ int fl;
    for (int h = me; h < hops; h += nwg) {
        if (h > 0) {
            if (tid == 0) {
                int spins = 0;
                while (ldflag<LAUX>(rf, (unsigned)(h - 1) * 128) == 0 && ++spins < (1 << 12)) __builtin_amdgcn_s_sleep(1);
                fl = spins;
                if (spins >= (1 << 12)) atomicAdd(bad, 1 << 20);
            }
            __syncthreads();
            const unsigned src = (unsigned)((h - 1) % nwg) * 8192;
            for (int q = 0; q < 2; ++q) {
                const d2 v = ld16<LAUX>(rt, src + (tid * 2 + 512 * q) * 8);
                nbad += v[0] != (double)(h - 1) * 4096.0 + tid * 2 + 512 * q;
                nbad += v[1] != (double)(h - 1) * 4096.0 + tid * 2 + 512 * q + 1;
            }
        }
        const unsigned dst = (unsigned)me * 8192;
        for (int q = 0; q < 2; ++q) {
            const double v = (double)h * 4096.0 + tid * 2 + 512 * q;
            st16<SAUX>(rt, dst + (tid * 2 + 512 * q) * 8, v, v + 1);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) stflag<FAUX>(rf, (unsigned)h * 128, 1 + myx);
        if (MODE == 2) {
            for (int q = 0; q < 2; ++q) {
                const double v = (double)h * 4096.0 + tid * 2 + 512 * q;
                st16<16>(rt, dst + (tid * 2 + 512 * q) * 8, v, v + 1);
            }
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    __syncthreads();
    if (me == (hops - 1) % nwg && tid == 0) stflag<16>(rsrc(stop), 0, 1);
}
int main() {
    double *tiles, *bgbuf; int *flags, *xcc, *bad, *stop;
    const int hops = 2000;
    const long bgn = 1L << 27;  // 1 GiB of doubles
    hipMalloc(&tiles, 8192 * 64); hipMalloc(&bad, 4); hipMalloc(&flags, 128 * hops); hipMalloc(&xcc, 4 * 64); hipMalloc(&stop, 4);
    hipMalloc(&bgbuf, bgn * 8); hipMemset(bgbuf, 0, bgn * 8);
    const char* names[4] = {"wt", "l2", "l2+wt", "l2f"};
    for (int bg : {0, 1, 2})
        for (int mode = 0; mode < 4; ++mode)
            for (int nwg : {2, 18}) {
                hipMemset(flags, 0, 128 * hops); hipMemset(bad, 0, 4); hipMemset(stop, 0, 4);
                hipMemset(tiles, 0, 8192 * 64);
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                const int grid = 220;
                if (mode == 0) k_ring<0><<<grid, 256>>>(tiles, flags, hops, nwg, bad, xcc, stop, bgbuf, bgn, bg);
                if (mode == 1) k_ring<1><<<grid, 256>>>(tiles, flags, hops, nwg, bad, xcc, stop, bgbuf, bgn, bg);
                if (mode == 2) k_ring<2><<<grid, 256>>>(tiles, flags, hops, nwg, bad, xcc, stop, bgbuf, bgn, bg);
                if (mode == 3) k_ring<3><<<grid, 256>>>(tiles, flags, hops, nwg, bad, xcc, stop, bgbuf, bgn, bg);
                hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                int hb, hx[64];
                hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
                hipMemcpy(hx, xcc, 4 * nwg, hipMemcpyDeviceToHost);
                std::printf("bg %d  %-6s %2d WGs: %.3f us/hop  xcc ids %d %d %d  wrong values %d  %s\n", bg, names[mode], nwg,
                            ms * 1e3 / hops, hx[0], hx[1], hx[nwg - 1], hb, hipGetErrorString(hipGetLastError()));
                std::fflush(stdout);
            }
    return 0;
}
