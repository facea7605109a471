// dtl_ablate.h -- timing experiments on the kernels (tools/ablate.py).
//
// A library built with -DDTL_ABLATE=<k> leaves ONE piece of work out of ONE kernel: its RESULTS ARE WRONG BY CONSTRUCTION,
// only kernel times mean something ("what does this piece cost?").  The shipped library is built with DTL_ABLATE == 0:
// every `if constexpr (ablated(k))` in the kernels is false there and compiles to nothing.  tools/ablate.py builds the
// variants into gr_dtl_amd/csrc/_obj/ablate_<k>.so and runs a probe against each through DTL_HIP_LIB -- the shipped
// libdtl_hip.so is never touched.
//
//   frame equaliser (frame_eq_kernel)   1 no derotation      2 no butterflies        3 no equaliser recursion
//   sync kernel (sync_edges_kernel)    11 no window test    12 no strip totals either
//   decoder (ldpc_bp_np_kernel)        21 no syndrome check 22 no output stage      23 no soft-bit loads
//                                      24 channel ratios without the exponential    25 no tame / wild test
//                                      26 no iteration at all (first syndrome check only)
//   demapper (demap_kernel)            31 no exponentials   32 no logarithms        33 no stores   34 no symbol loads
//   header kernel (header_kernel)      41 no integer-offset search   42 no derotation   43 no butterflies
//                                      44 no header parse (CRC, fields)
#pragma once

#ifndef DTL_ABLATE
#define DTL_ABLATE 0
#endif

namespace dtl {
constexpr int kAblate = DTL_ABLATE;
__host__ __device__ constexpr bool ablated(int k) { return kAblate == k; }
} // namespace dtl
