// stream_variants.hip — NOT part of the product.  Which launch shape streams 2 reads + 1 write fastest on MI355X?
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

template <int ROWS, bool NTL, bool NTS, bool PERSIST>
__global__ __launch_bounds__(256) void kv(const float* __restrict__ x, const int* __restrict__ k, float* __restrict__ y, long long n4) {
  const long long chunk = 256LL * ROWS;
  const long long nchunks = (n4 + chunk - 1) / chunk;
  for (long long c = blockIdx.x; c < nchunks; c += PERSIST ? gridDim.x : nchunks) {
    const long long base = c * chunk + threadIdx.x;
    f4 a[ROWS]; i4 b[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const long long i = base + r * 256;
      if (i < n4) {
        a[r] = NTL ? __builtin_nontemporal_load((const f4*)x + i) : ((const f4*)x)[i];
        b[r] = NTL ? __builtin_nontemporal_load((const i4*)k + i) : ((const i4*)k)[i];
      }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const long long i = base + r * 256;
      if (i < n4) {
        f4 o; o.x = a[r].x * (b[r].x & 1 ? 1.0f : 0.5f); o.y = a[r].y * (b[r].y & 1 ? 1.0f : 0.5f);
        o.z = a[r].z * (b[r].z & 1 ? 1.0f : 0.5f); o.w = a[r].w * (b[r].w & 1 ? 1.0f : 0.5f);
        if (NTS) __builtin_nontemporal_store(o, (f4*)y + i); else ((f4*)y)[i] = o;
      }
    }
  }
}

template <int ROWS, bool NTL, bool NTS, bool PERSIST>
static int launch(const float* x, const int* k, float* y, long long n, int grid_persist, hipStream_t s) {
  const long long n4 = n / 4;
  const long long nchunks = (n4 + 256LL * ROWS - 1) / (256LL * ROWS);
  const unsigned grid = PERSIST ? (unsigned)grid_persist : (unsigned)nchunks;
  hipLaunchKernelGGL((kv<ROWS, NTL, NTS, PERSIST>), dim3(grid), dim3(256), 0, s, x, k, y, n4);
  return (int)hipGetLastError();
}

extern "C" int stream_variant(int id, const float* x, const int* k, float* y, long long n, int grid_persist, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  switch (id) {
    case 0: return launch<4, false, true, false>(x, k, y, n, 0, s);
    case 1: return launch<2, false, true, false>(x, k, y, n, 0, s);
    case 2: return launch<8, false, true, false>(x, k, y, n, 0, s);
    case 3: return launch<16, false, true, false>(x, k, y, n, 0, s);
    case 4: return launch<4, true, true, false>(x, k, y, n, 0, s);
    case 5: return launch<8, true, true, false>(x, k, y, n, 0, s);
    case 6: return launch<4, false, true, true>(x, k, y, n, grid_persist, s);
    case 7: return launch<8, false, true, true>(x, k, y, n, grid_persist, s);
    case 8: return launch<4, true, true, true>(x, k, y, n, grid_persist, s);
    case 9: return launch<1, false, true, false>(x, k, y, n, 0, s);
  }
  return -1;
}
