// Measures the cost of a dependent kernel boundary (same stream, hipGraph replay and eager).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_touch(double* a, int n) {   // every WG reads+writes 2 KB that the previous kernel wrote
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[(i + 4096) % n] + 1.0;
}
template <class F> double run(const char* name, F launch, int nk, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < nk; ++i) launch();
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::printf("%-44s graph: %.2f us/kernel", name, ms * 1e3 / (10 * nk));
    hipEventRecord(e0, st);
    for (int r = 0; r < 10 * nk; ++r) launch();
    hipEventRecord(e1, st); hipStreamSynchronize(st);
    hipEventElapsedTime(&ms, e0, e1);
    std::printf("   eager: %.2f us/kernel\n", ms * 1e3 / (10 * nk));
    return ms;
}
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double* a; int n = 1 << 22; hipMalloc(&a, sizeof(double) * n); hipMemset(a, 0, sizeof(double) * n);
    run("empty 1 WG x 64", [&] { k_empty<<<1, 64, 0, st>>>(nullptr); }, 200, st);
    run("empty 256 WG x 256", [&] { k_empty<<<256, 256, 0, st>>>(nullptr); }, 200, st);
    run("empty 1040 WG x 256", [&] { k_empty<<<1040, 256, 0, st>>>(nullptr); }, 200, st);
    run("touch 544 threads (4 KB)", [&] { k_touch<<<3, 256, 0, st>>>(a, 544); }, 200, st);
    run("touch 256 WG x 256 (0.5 MB)", [&] { k_touch<<<256, 256, 0, st>>>(a, 65536); }, 200, st);
    run("touch 4096 WG x 256 (8 MB)", [&] { k_touch<<<4096, 256, 0, st>>>(a, 1 << 20); }, 200, st);
    return 0;
}
