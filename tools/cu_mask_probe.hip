// Which CUs does a stream created with hipExtStreamCreateWithCUMask(mask) run on?  For a few masks: launch 4096
// single-wave workgroups that spin ~20 us each and record (XCC_ID, HW_ID's SE / CU fields); print the set of XCDs and
// the number of distinct (xcc, se, cu) triples used.  Build: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void probe(unsigned* out, long spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;           // XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);                  // HW_ID (all 32 bits)
    const long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 28) | (hw & 0x0fffffff);
}

int main() {
    const int nb = 4096;
    unsigned* d;
    hipMalloc(&d, nb * sizeof(unsigned));
    std::vector<unsigned> h(nb);
    struct M { const char* name; uint32_t w[8]; };
    M masks[] = {
        {"all", {0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff}},
        {"word0", {0xffffffff, 0, 0, 0, 0, 0, 0, 0}},
        {"word1", {0, 0xffffffff, 0, 0, 0, 0, 0, 0}},
        {"every8th_from0", {0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101, 0x01010101}},
        {"every8th_from1", {0x02020202, 0x02020202, 0x02020202, 0x02020202, 0x02020202, 0x02020202, 0x02020202, 0x02020202}},
        {"low_byte_of_each_word", {0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff}},
    };
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m.w);
        if (e != hipSuccess) { printf("%s: create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nb * sizeof(unsigned), s);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(64), 0, s, d, 2000L);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, nb * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::set<unsigned> xccs, cus;
        int per_xcc[16] = {0};
        for (unsigned v : h) {
            const unsigned xcc = v >> 28, se = (v >> 13) & 7, cu = (v >> 8) & 15;   // HW_ID: CU_ID[11:8], SH[12], SE_ID[15:13]
            xccs.insert(xcc);
            cus.insert((xcc << 8) | (se << 5) | ((v >> 12 & 1) << 4) | cu);
            per_xcc[xcc]++;
        }
        printf("%-22s xcds:", m.name);
        for (unsigned x : xccs) printf(" %u(%d)", x, per_xcc[x]);
        printf("  distinct CUs: %zu\n", cus.size());
        hipStreamDestroy(s);
    }
    return 0;
}
