// dtl_common.h -- shared host-side helpers for the MI355X (gfx950) gr-dtl RX library.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

// Status codes come from the public C ABI header.
#include "../../include/dtl_rx.h"

namespace dtl {

// The error text of the calling thread lives in a fixed thread-local buffer: setting it never allocates and never
// throws, so it can be called from a catch handler for std::bad_alloc.
void set_last_error(const char* msg) noexcept;
void set_last_error(const std::string& msg) noexcept;
const char* last_error() noexcept;

struct HipError {
    hipError_t e;
    const char* what;
};

#define DTL_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::dtl::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));    \
            (void)hipGetLastError(); /* reported: take it out of the runtime's "last error" slot, or the next     \
                                        hipGetLastError() behind a launch -- of ANY handle -- reports it again */ \
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
