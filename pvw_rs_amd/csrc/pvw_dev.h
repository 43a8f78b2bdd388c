// pvw_dev.h -- what the kernel translation units (pvw_mac / pvw_poly / pvw_decrypt / pvw_decode_kernels / pvw_gemm .hip)
// share on the device side.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "pvw_arith.h"

namespace pvw {

typedef u64 v2u64 __attribute__((ext_vector_type(2)));  // one 16-byte lane access
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

}  // namespace pvw

// launchers: the ring degree l is a template parameter of every kernel that keeps a polynomial limb in registers
#define PVW_DISPATCH_ELL(ell, ...)                       \
  switch (ell) {                                         \
    case 8:  { constexpr int E = 8;  __VA_ARGS__; } break;   \
    case 16: { constexpr int E = 16; __VA_ARGS__; } break;   \
    case 32: { constexpr int E = 32; __VA_ARGS__; } break;   \
    case 64: { constexpr int E = 64; __VA_ARGS__; } break;   \
    default: return hipErrorInvalidValue;                \
  }
