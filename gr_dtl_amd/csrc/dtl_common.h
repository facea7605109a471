// dtl_common.h -- shared host-side helpers for the MI355X (gfx950) gr-dtl RX library.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

// Status codes come from the public C ABI header.
#include "../../include/dtl_rx.h"

namespace dtl {

void set_last_error(const std::string& msg);
const char* last_error();

struct HipError {
    hipError_t e;
    const char* what;
};

#define DTL_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::dtl::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));    \
            return DTL_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

template <typename T>
inline int dev_alloc_copy(T** dptr, const T* host, size_t count)
{
    DTL_HIP_CHECK(hipMalloc((void**)dptr, (count ? count : 1) * sizeof(T)));
    if (count) DTL_HIP_CHECK(hipMemcpy(*dptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    return DTL_OK;
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

} // namespace dtl
