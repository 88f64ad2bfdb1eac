// stream_ceiling.hip — NOT part of the product.  Pure streaming kernels with the same HBM traffic mix as
// the scans (2 reads + 1 write = 12 B/element, 4 reads + 1 write = 20 B/element), 16 B per lane, no
// dependencies between elements: the bandwidth an ideal kernel of that shape reaches on this device.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int ROWS, bool NT, bool NTL = false>
__global__ __launch_bounds__(256) void k12(const float* __restrict__ x, const int* __restrict__ k, float* __restrict__ y, long long n4) {
  const long long base = (long long)blockIdx.x * 256 * ROWS + threadIdx.x;
  f4 a[ROWS]; i4 b[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) { const long long i = base + r * 256; if (i < n4) { a[r] = NTL ? __builtin_nontemporal_load((const f4*)x + i) : ((const f4*)x)[i]; b[r] = NTL ? __builtin_nontemporal_load((const i4*)k + i) : ((const i4*)k)[i]; } }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const long long i = base + r * 256;
    if (i < n4) {
      f4 o; o.x = a[r].x * (b[r].x & 1 ? 1.0f : 0.5f); o.y = a[r].y * (b[r].y & 1 ? 1.0f : 0.5f);
      o.z = a[r].z * (b[r].z & 1 ? 1.0f : 0.5f); o.w = a[r].w * (b[r].w & 1 ? 1.0f : 0.5f);
      if (NT) __builtin_nontemporal_store(o, (f4*)y + i); else ((f4*)y)[i] = o;
    }
  }
}

template <int ROWS, bool NT, bool NTL = false>
__global__ __launch_bounds__(256) void k20(const float* __restrict__ x, const float* __restrict__ c, const float* __restrict__ g,
                                           const int* __restrict__ k, float* __restrict__ y, long long n4) {
  const long long base = (long long)blockIdx.x * 256 * ROWS + threadIdx.x;
  f4 a[ROWS], cc[ROWS], gg[ROWS]; i4 b[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const long long i = base + r * 256;
    if (i < n4) {
      if (NTL) { a[r] = __builtin_nontemporal_load((const f4*)x + i); cc[r] = __builtin_nontemporal_load((const f4*)c + i); gg[r] = __builtin_nontemporal_load((const f4*)g + i); b[r] = __builtin_nontemporal_load((const i4*)k + i); }
      else { a[r] = ((const f4*)x)[i]; cc[r] = ((const f4*)c)[i]; gg[r] = ((const f4*)g)[i]; b[r] = ((const i4*)k)[i]; }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const long long i = base + r * 256;
    if (i < n4) {
      f4 o = gg[r] * cc[r] + a[r];
      o.x += (b[r].x & 1); o.y += (b[r].y & 1); o.z += (b[r].z & 1); o.w += (b[r].w & 1);
      if (NT) __builtin_nontemporal_store(o, (f4*)y + i); else ((f4*)y)[i] = o;
    }
  }
}

extern "C" int ceiling12(const float* x, const int* k, float* y, long long n, int nt, void* stream) {
  const long long n4 = n / 4; constexpr int R = 4;
  const unsigned grid = (unsigned)((n4 + 256 * R - 1) / (256 * R));
  if (nt == 2) hipLaunchKernelGGL((k12<R, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, k, y, n4);
  else if (nt) hipLaunchKernelGGL((k12<R, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, k, y, n4);
  else hipLaunchKernelGGL((k12<R, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, k, y, n4);
  return (int)hipGetLastError();
}
extern "C" int ceiling20(const float* x, const float* c, const float* g, const int* k, float* y, long long n, int nt, void* stream) {
  const long long n4 = n / 4; constexpr int R = 4;
  const unsigned grid = (unsigned)((n4 + 256 * R - 1) / (256 * R));
  if (nt == 2) hipLaunchKernelGGL((k20<R, true, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, c, g, k, y, n4);
  else if (nt) hipLaunchKernelGGL((k20<R, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, c, g, k, y, n4);
  else hipLaunchKernelGGL((k20<R, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, c, g, k, y, n4);
  return (int)hipGetLastError();
}
