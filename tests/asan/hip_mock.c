/* Stand-in for the HIP runtime, for SANITIZER runs of the library's HOST code on a machine without a GPU
 * (tests/asan/run_asan.sh).  Device memory is host memory (calloc), copies are memcpy, kernels do not run, graphs record
 * nothing.  What is exercised is everything around the kernels: descriptor checks, layout conversions, ELL assembly,
 * reference counting, staging buffers, graph bookkeeping, error paths -- the ~1.3 k lines of host C++ in csrc/ under
 * AddressSanitizer + UBSan.  LD_PRELOADed, so its (unversioned) definitions win over libamdhip64's.
 * Test infrastructure only: nothing in the product links or loads this file. */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
typedef struct { unsigned x, y, z; } dim3;
#define OK 0

static int capturing = 0;
hipError_t hipGetDeviceCount(int* c) { *c = 1; return OK; }
hipError_t hipSetDevice(int d) { (void)d; return OK; }
hipError_t hipDeviceSynchronize(void) { return OK; }
hipError_t hipGetLastError(void) { return OK; }
const char* hipGetErrorString(hipError_t e) { (void)e; return "hip mock"; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? OK : 2; }
hipError_t hipFree(void* p) { free(p); return OK; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags) { (void)flags; *p = calloc(n ? n : 1, 1); return *p ? OK : 2; }
hipError_t hipHostFree(void* p) { free(p); return OK; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned flags) { (void)flags; *d = h; return OK; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int kind) { (void)kind; memcpy(d, s, n); return OK; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int kind, hipStream_t st) { (void)kind; (void)st; if (!capturing) memcpy(d, s, n); return OK; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, int kind, hipStream_t st) {
    (void)kind; (void)st;
    for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, w);
    return OK;
}
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return OK; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { (void)st; if (!capturing) memset(d, v, n); return OK; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned f) { (void)f; *s = malloc(8); return OK; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return OK; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return OK; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(8); return OK; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return OK; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return OK; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return OK; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)a; (void)b; *ms = 0.f; return OK; }
hipError_t hipFuncSetAttribute(const void* f, int attr, int v) { (void)f; (void)attr; (void)v; return OK; }
hipError_t hipStreamBeginCapture(hipStream_t s, int mode) { (void)s; (void)mode; capturing = 1; return OK; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) { (void)s; capturing = 0; *g = malloc(8); return OK; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void* a, void* b, size_t n) { (void)g; (void)a; (void)b; (void)n; *e = malloc(8); return OK; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return OK; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return OK; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t s) { (void)e; (void)s; return OK; }
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t shmem, hipStream_t s) {
    (void)f; (void)g; (void)b; (void)args; (void)shmem; (void)s; return OK;
}
void** __hipRegisterFatBinary(const void* data) { (void)data; static void* h; return &h; }
void __hipUnregisterFatBinary(void** h) { (void)h; }
void __hipRegisterFunction(void** h, const void* hf, char* df, const char* dn, unsigned tl, void* a, void* b, void* c, void* d, int* e) {
    (void)h; (void)hf; (void)df; (void)dn; (void)tl; (void)a; (void)b; (void)c; (void)d; (void)e;
}
void __hipRegisterVar(void** h, void* var, char* hv, char* dv, int ext, size_t size, int constant, int global) {
    (void)h; (void)var; (void)hv; (void)dv; (void)ext; (void)size; (void)constant; (void)global;
}
static dim3 cfg_g, cfg_b; static size_t cfg_sh; static hipStream_t cfg_st;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { cfg_g = g; cfg_b = b; cfg_sh = sh; cfg_st = st; return OK; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) { *g = cfg_g; *b = cfg_b; *sh = cfg_sh; *st = cfg_st; return OK; }
