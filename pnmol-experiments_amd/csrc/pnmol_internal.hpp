// Private to the library: what the translation units under csrc/ share.  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

struct pnmol_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
};
