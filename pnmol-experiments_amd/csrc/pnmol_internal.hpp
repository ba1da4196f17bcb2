// Private to the library: what the translation units under csrc/ share.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <string>

struct pnmol_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // Lifetime rule of the C ABI (include/pnmol_hip.h, "Lifetimes"): a handle keeps its parent alive.  `children` counts
    // the live pnmol_filter / pnmol_sqrt_filter objects of this ctx; pnmol_ctx_destroy refuses (-1) while it is not zero.
    std::atomic<int> children{0};
};
