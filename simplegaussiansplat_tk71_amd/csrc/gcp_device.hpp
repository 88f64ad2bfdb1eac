// gcp_device.hpp — device/host helpers shared by the HIP sources of libgrouped_cumprod_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcp {

typedef long long i64;

// DPP helpers (gfx9 encodings: row_shr:n = 0x110+n, wave_shr:1 = 0x138, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143).  bound_ctrl = 0: a lane whose source is out of range (or whose row is
// masked off) keeps `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                         CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Wave-wide inclusive prefix sum of ints.
__device__ __forceinline__ int wave_incl_scan_i(int v) {
  v += dpp_i<0x111, 0xf>(0, v);
  v += dpp_i<0x112, 0xf>(0, v);
  v += dpp_i<0x114, 0xf>(0, v);
  v += dpp_i<0x118, 0xf>(0, v);
  v += dpp_i<0x142, 0xa>(0, v);
  v += dpp_i<0x143, 0xc>(0, v);
  return v;
}

// host side (defined in gcp_scan.hip)
int hip_fail(hipError_t e);

}  // namespace gcp

#define GCP_HIP(call)                                        \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return ::gcp::hip_fail(e_);        \
  } while (0)
