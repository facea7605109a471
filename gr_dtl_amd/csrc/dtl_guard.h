// dtl_guard.h -- the exception barrier of the C ABI (include/dtl_rx.h): included by every file that defines extern "C"
// entry points.
#pragma once

#include "dtl_common.h"

#include <cstdio>
#include <new>
#include <stdexcept>
#include <system_error>

namespace dtl {
// Every extern "C" entry point runs its body through this: no C++ exception leaves the library (the caller may be C, a
// ctypes binding or a GNU Radio scheduler thread -- an exception there is std::terminate in the caller's process).
// The reference throws out of work() instead (lib/dtl/ofdm_adaptive_frame_equalizer_vcvc_impl.cc:148-150,
// lib/dtl/tb_decoder.cc:58-60); here those conditions are status codes, and what is caught below is the C++ runtime's own
// failures: host memory (std::vector / std::string growth), a thread that cannot be started.
template <typename F>
inline int guarded(const char* who, F&& body) noexcept
{
    char text[960];
    try {
        return body();
    } catch (const std::bad_alloc&) {
        std::snprintf(text, sizeof(text), "%s: out of host memory (std::bad_alloc)", who);
        set_last_error(text);
        return DTL_ERR_CAPACITY;
    } catch (const std::length_error& e) {
        std::snprintf(text, sizeof(text), "%s: size beyond what a host container can hold (%s)", who, e.what());
        set_last_error(text);
        return DTL_ERR_CAPACITY;
    } catch (const std::system_error& e) {
        std::snprintf(text, sizeof(text), "%s: system error (%s)", who, e.what());
        set_last_error(text);
        return DTL_ERR_HIP;
    } catch (const std::exception& e) {
        std::snprintf(text, sizeof(text), "%s: unexpected exception (%s)", who, e.what());
        set_last_error(text);
        return DTL_ERR_INTERNAL;
    } catch (...) {
        std::snprintf(text, sizeof(text), "%s: unexpected exception", who);
        set_last_error(text);
        return DTL_ERR_INTERNAL;
    }
}
} // namespace dtl

#define DTL_API_BEGIN(who) return ::dtl::guarded(who, [&]() -> int {
#define DTL_API_END });
