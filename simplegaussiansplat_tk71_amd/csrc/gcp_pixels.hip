// gcp_pixels.hip — the per-pixel carry of the reference's chunked calls (the callers either side of
// `_create_alpha_brend` / `grad_cumsum` in `_forward_batch` / `_backward_batch`).
//
// Every chunk of the reference's forward ends with `_create_alpha_brend_min(rects, T)` (gs_model.py:582-586, called at
// :609 / :615): `torch.unique(rects, dim=0)` — a lexicographic sort of the M-row list — and a `scatter_reduce(amin)` over
// its inverse; every chunk of the backward with `create_grad_alphabrend_min(rects, grad)` (:724-730, called at :639 /
// :643): the same with the pair's own index as the value, i.e. the FIRST pair of every pixel.  Both are a minimum per
// pixel, and a minimum needs no order: every pair takes the minimum with its pixel's cell in an image-sized table
// (1921 x 1081 cells at 1080p: 8.3 MB, held by the Infinity Cache and in part by the 4 MB L2 of each XCD) and the table is read out column by column — (x, y) ascending, which is
// the row order `torch.unique(dim=0)` returns.  One pass over the M-sized list (12 B per pair read, nothing M-sized
// written), then K-sized work; integer minima of order-preserving images of the floats: the result does not depend on the
// order in which the pairs arrive, bit for bit what the reference returns.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gcp_device.hpp"
#include "grouped_cumprod_hip.h"

namespace {
using namespace gcp;

constexpr unsigned kNoPair = 0xffffffffu;  // a cell no pair has touched
constexpr int kPairTile = 4096;            // pairs per block: 16 steps of 256 threads

// fp32 -> unsigned with the same order (a < b  <=>  enc(a) < enc(b), -0 below +0).  NaN -> 0, the smallest image, which
// decodes to a NaN: a pixel with a NaN among its values comes out NaN, as `amin` has it.  0xffffffff is no value's image
// (it would be that of the NaN 0x7fffffff, which goes to 0 like every NaN): it marks the untouched cell.
__device__ __forceinline__ unsigned enc_f32(float v) {
  const unsigned b = __float_as_uint(v);
  if (v != v) return 0u;
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

template <bool I64>
__device__ __forceinline__ int2 pair_at(const void* __restrict__ rects, i64 i) {
  if (!I64) return reinterpret_cast<const int2*>(rects)[i];
  typedef long long ll2 __attribute__((ext_vector_type(2), aligned(8)));
  const ll2 q = reinterpret_cast<const ll2*>(rects)[i];
  const bool ok = ((unsigned long long)q.x | (unsigned long long)q.y) < 0x80000000ull;
  return ok ? make_int2((int)q.x, (int)q.y) : make_int2(-1, -1);
}

// ---- coordinate range of a list (for callers that do not pass the image size) -------------------------------------------
template <bool I64>
__global__ __launch_bounds__(256) void k_pixels_range(const void* __restrict__ rects, i64 n, int* __restrict__ out /*{max x, max y, min}*/) {
  __shared__ int s_x[4], s_y[4], s_n[4];
  int mx = 0, my = 0, mn = 0x7fffffff;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const int2 r = pair_at<I64>(rects, i);
    mx = max(mx, r.x); my = max(my, r.y); mn = min(mn, min(r.x, r.y));
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, __shfl_xor(mx, o)); my = max(my, __shfl_xor(my, o)); mn = min(mn, __shfl_xor(mn, o));
  }
  if ((threadIdx.x & 63) == 0) { s_x[threadIdx.x >> 6] = mx; s_y[threadIdx.x >> 6] = my; s_n[threadIdx.x >> 6] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(out, max(max(s_x[0], s_x[1]), max(s_x[2], s_x[3])));
    atomicMax(out + 1, max(max(s_y[0], s_y[1]), max(s_y[2], s_y[3])));
    atomicMin(out + 2, min(min(s_n[0], s_n[1]), min(s_n[2], s_n[3])));
  }
}

// ---- pass 1: every pair takes the minimum with its pixel's cell ------------------------------------------------------------
// INDEX: the value of pair i is i itself (create_grad_alphabrend_min: the first pair of every pixel).
// A pair LOOKS at its cell first and leaves it alone when the cell already holds something as small (a stale look can only
// show a larger value than the cell holds: the atomic then decides).  How many atomics that spares depends on the order the
// blocks run in against the order of the values: indices grow along the list, so the blocks take it front to back and nearly
// every later pair of a pixel finds an earlier one's index in the cell; transmittances FALL along a pixel's list (what
// _create_alpha_brend_min is called on, gs_model.py:609), so with values the blocks take the list back to front.  Any other
// values: same result, more atomics.  All sixteen looks of a thread are issued before the first atomic (a look behind an
// atomic waits for it: the compiler cannot move a load over an atomic that may hit the same address).
#ifndef GCP_PIXELS_LOOK
#define GCP_PIXELS_LOOK 1      // measurement switches (tools/build_variant.py): 0 = every pair issues its atomic
#endif
#ifndef GCP_PIXELS_FILTER_FROM
#define GCP_PIXELS_FILTER_FROM (1 << 25)  // pairs from which the looks go to the 16-bit filter table (below: to the cells themselves)
#endif
#ifndef GCP_PIXELS_REVERSE
#define GCP_PIXELS_REVERSE 1   // 0 = values front to back as well
#endif
template <bool I64, bool INDEX, bool FILTER>
__global__ __launch_bounds__(256) void k_pixels_min(const void* __restrict__ rects, const float* __restrict__ values, i64 n, int w1, int h1,
                                                    unsigned* __restrict__ cell, unsigned short* __restrict__ filter, int* __restrict__ info) {
  constexpr int kSteps = kPairTile / 256;  // 16
  const i64 tile = (!INDEX && GCP_PIXELS_REVERSE) ? (i64)gridDim.x - 1 - blockIdx.x : (i64)blockIdx.x;
  // lane l of a wave takes pair (wave's 64 * step) + l: the 64 cells one instruction looks at are those of 64 CONSECUTIVE
  // pairs — five box rows of thirteen neighbouring cells, some seven cache lines (with four consecutive pairs per lane, the
  // shape of the wide loads, every instruction touched the rows of 256 pairs: 1.0 instead of 0.6 ms at 1.65e8 pairs)
  const i64 base = tile * kPairTile + threadIdx.x;
  const bool full = (tile + 1) * kPairTile <= n;  // block-uniform
  int2 e[kSteps];
  float v[kSteps];
  // straight-line: all loads of the thread are issued before anything waits (indices behind the end are clamped to the last
  // pair, their values never used)
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const i64 p = base + s * 256;
    const i64 q = (full || p < n) ? p : n - 1;
    e[s] = pair_at<I64>(rects, q);
    if (!INDEX) v[s] = values[q];
  }
  // the cell of every pair (a pair behind the end or outside the image looks at cell 0 and never writes), its image, the looks
  unsigned at[kSteps], u[kSteps], seen[kSteps];
  unsigned live = 0u;
  bool bad = false;
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    const i64 i = base + s * 256;
    const int2 r = e[s];
    const bool in_list = full || i < n;
    const bool inside = (unsigned)r.x < (unsigned)w1 && (unsigned)r.y < (unsigned)h1;
    bad |= in_list && !inside;
    const bool ok = in_list && inside;
    live |= (ok ? 1u : 0u) << s;
    at[s] = ok ? (unsigned)r.y * (unsigned)w1 + (unsigned)r.x : 0u;  // (cells <= 2^28)
    u[s] = INDEX ? (unsigned)i : enc_f32(v[s]);
  }
  if (FILTER) {
    // The looks go to a table of 16-bit words, half the size of the cells (4.2 instead of 8.3 MB at 1080p, against 4 MB of L2
    // per XCD): filter[c] is an UPPER bound of the top half of cell c — 0xffff at first; whoever sends a value to the cell
    // writes the value's top half to the filter afterwards (a plain store: whichever of two racing stores lands last, it is the
    // top half of a value that has been, or is being, sent to the cell, hence no smaller than the cell's).  filter[c] < top(u)
    // therefore means: something smaller than u is in the cell or on its way there — u is not needed.  Equal top halves decide
    // nothing: the atomic goes out.  Measured at 1.65e8 pairs: 0.59 -> 0.53 ms for transmittances, 0.51 -> 0.46 for the
    // first-pair index, 0.71 -> 0.80 for values in no order (more ties); at 1.6e7 pairs the second table costs more than it
    // saves: FILTER from 2^25 pairs.
#pragma unroll
    for (int s = 0; s < kSteps; ++s) seen[s] = __hip_atomic_load(filter + at[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const unsigned top = u[s] >> 16;
      if (((live >> s) & 1u) && seen[s] >= top) {
        atomicMin(cell + at[s], u[s]);
        if (seen[s] > top) __hip_atomic_store(filter + at[s], (unsigned short)top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else {
    if (GCP_PIXELS_LOOK) {
#pragma unroll
      for (int s = 0; s < kSteps; ++s) seen[s] = __hip_atomic_load(cell + at[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      if (((live >> s) & 1u) && (!GCP_PIXELS_LOOK || seen[s] > u[s])) atomicMin(cell + at[s], u[s]);
    }
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) {
    if (!__hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(info + 1, 1);
  }
}

// ---- pass 2: the table read out in (x, y) order ------------------------------------------------------------------------------
// The result lists the touched cells column by column ((x, y) ascending: the row order of torch.unique(dim=0)), the table is
// stored row by row.  A block takes a PATCH of 16 columns x 128 rows: it reads the patch with 64-byte row segments into LDS
// (a thread that walked a column of the table itself fetched a 64-byte sector for every 4-byte cell), and each wave walks
// four of the patch's columns there, its lanes over consecutive rows.  The (column, row band) pieces are contiguous runs of
// the result in the order column-major, band-minor: one count per piece, ONE exclusive scan over all pieces, ranked writes —
// neighbouring lanes write neighbouring rows of the result.
constexpr int kColTile = 16, kBandRows = 128, kColStride = 17;  // (stride 17: a column walk touches 64 different banks)
template <bool WRITE, bool I64, bool INDEX>
__global__ __launch_bounds__(256) void k_pixels_readout(const unsigned* __restrict__ cell, int w1, int h1, int bands, int* __restrict__ cnt,
                                                        const int* __restrict__ off, i64 capacity, void* __restrict__ out_xy,
                                                        float* __restrict__ out_val, int* __restrict__ info) {
  __shared__ unsigned s_tile[kBandRows * kColStride];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kColTile, band = blockIdx.y, y0 = band * kBandRows;
  unsigned u[kBandRows * kColTile / 256];
#pragma unroll
  for (int j = 0; j < kBandRows * kColTile / 256; ++j) {  // eight loads per thread, all issued before the first LDS store
    const int i = j * 256 + threadIdx.x, r = i >> 4, xx = i & 15;
    const bool in = x0 + xx < w1 && y0 + r < h1;
    u[j] = cell[in ? (size_t)(y0 + r) * w1 + x0 + xx : 0];
    if (!in) u[j] = kNoPair;
  }
#pragma unroll
  for (int j = 0; j < kBandRows * kColTile / 256; ++j) {
    const int i = j * 256 + threadIdx.x;
    s_tile[(i >> 4) * kColStride + (i & 15)] = u[j];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < kColTile / 4; ++c) {
    const int xx = w + 4 * c;  // wave w: columns w, w + 4, w + 8, w + 12 of the patch
    if (x0 + xx >= w1) break;  // wave-uniform
    const int piece = (x0 + xx) * bands + band;
    const unsigned v0 = s_tile[lane * kColStride + xx], v1 = s_tile[(lane + 64) * kColStride + xx];
    const unsigned long long m0 = __ballot(v0 != kNoPair), m1 = __ballot(v1 != kNoPair);
    if (!WRITE) {
      if (lane == 0) cnt[piece] = __builtin_popcountll(m0) + __builtin_popcountll(m1);
      continue;
    }
    const i64 o = off[piece];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const unsigned v = half ? v1 : v0;
      const i64 at = o + (half ? __builtin_popcountll(m0) : 0) + __builtin_popcountll((half ? m1 : m0) & below);
      if (v != kNoPair && at < capacity) {
        const int y = y0 + lane + 64 * half;
        if (I64) {
          reinterpret_cast<long long*>(out_xy)[2 * at] = x0 + xx;
          reinterpret_cast<long long*>(out_xy)[2 * at + 1] = y;
        } else {
          reinterpret_cast<int2*>(out_xy)[at] = make_int2(x0 + xx, y);
        }
        out_val[at] = INDEX ? (float)v : dec_f32(v);  // (INDEX: the float the reference carries the index in, gs_model.py:728)
      }
    }
  }
  if (WRITE && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) info[0] = off[(i64)w1 * bands];  // the scan's total
}

inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

struct PixelsWs {
  size_t cell, filter, cnt, off, scan, total;
  i64 blocks;
};
inline PixelsWs pixels_ws(i64 w1, i64 h1) {
  PixelsWs w;
  const i64 cells = w1 * h1;
  w.blocks = w1 * ((h1 + kBandRows - 1) / kBandRows);  // (column, row band) pieces of the read-out
  size_t o = 0;
  w.cell = o; o += align256((size_t)cells * sizeof(unsigned));
  w.filter = o; o += align256((size_t)cells * sizeof(unsigned short));
  w.cnt = o; o += align256((size_t)(w.blocks + 1) * sizeof(int));
  w.off = o; o += align256((size_t)(w.blocks + 1) * sizeof(int));
  w.scan = o; o += align256(gcp_scan_i32_workspace_bytes(w.blocks));
  w.total = o;
  return w;
}

constexpr i64 kMaxCells = 1ll << 28;  // 1 GB of cells: far beyond any image of the reference's key (x < 10000)

}  // namespace

extern "C" {

int gcp_pixels_range(const void* rects_xy, int32_t rects_are_int64, int64_t n, int32_t* out3, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || !out3) return GCP_ERR_INVALID_ARGUMENT;
  GCP_HIP(hipMemsetAsync(out3, 0, 2 * sizeof(int), stream));
  GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)(out3 + 2), 0x7fffffff, 1, stream));
  if (n == 0) return GCP_OK;
  if (!rects_xy) return GCP_ERR_INVALID_ARGUMENT;
  i64 blocks = (n + 4095) / 4096;
  if (blocks > 2048) blocks = 2048;
  if (rects_are_int64) hipLaunchKernelGGL((k_pixels_range<true>), dim3((unsigned)blocks), dim3(256), 0, stream, rects_xy, (i64)n, out3);
  else hipLaunchKernelGGL((k_pixels_range<false>), dim3((unsigned)blocks), dim3(256), 0, stream, rects_xy, (i64)n, out3);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

size_t gcp_pixels_min_workspace_bytes(int32_t width, int32_t height) {
  if (width < 0 || height < 0 || ((i64)width + 1) * ((i64)height + 1) > kMaxCells) return 0;
  return pixels_ws((i64)width + 1, (i64)height + 1).total;
}

int gcp_pixels_min(const void* rects_xy, int32_t rects_are_int64, const float* values, int64_t n, int32_t width, int32_t height,
                   void* out_xy, float* out_val, int64_t capacity, int32_t* info, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || n > 0x7fffffffLL || width < 0 || height < 0 || capacity < 0 || !info) return GCP_ERR_INVALID_ARGUMENT;
  const i64 w1 = (i64)width + 1, h1 = (i64)height + 1;
  if (w1 * h1 > kMaxCells) return GCP_ERR_INVALID_ARGUMENT;
  GCP_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int), stream));  // unique pixels, coordinate-outside-the-image flag, (2 unused)
  if (n == 0) return GCP_OK;
  if (!rects_xy || !ws || (capacity > 0 && (!out_xy || !out_val))) return GCP_ERR_INVALID_ARGUMENT;
  const PixelsWs L = pixels_ws(w1, h1);
  if (ws_bytes < L.total || ((uintptr_t)ws & 255u)) return GCP_ERR_WORKSPACE;
  char* const base = (char*)ws;
  unsigned* const cell = (unsigned*)(base + L.cell);
  int* const cnt = (int*)(base + L.cnt);
  int* const off = (int*)(base + L.off);
  GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)cell, (int)kNoPair, (size_t)(w1 * h1), stream));
  unsigned short* const filter = (unsigned short*)(base + L.filter);
  const char* ff = getenv("GCP_PIXELS_FILTER_FROM");  // read per call: the tests switch it inside one process
  const bool use_filter = n >= ((ff && *ff) ? (int64_t)atoll(ff) : (int64_t)GCP_PIXELS_FILTER_FROM);
  if (use_filter) GCP_HIP(hipMemsetD16Async((hipDeviceptr_t)filter, (unsigned short)0xffff, (size_t)(w1 * h1), stream));
  const unsigned pair_blocks = (unsigned)((n + kPairTile - 1) / kPairTile);
  const bool wide = rects_are_int64 != 0, index = values == nullptr;
#define GCP_PIXELS_MIN(I64, INDEX)                                                                                                    \
  do {                                                                                                                                \
    if (use_filter) hipLaunchKernelGGL((k_pixels_min<I64, INDEX, true>), dim3(pair_blocks), dim3(256), 0, stream, rects_xy, values,  \
                                       (i64)n, (int)w1, (int)h1, cell, filter, info);                                                \
    else hipLaunchKernelGGL((k_pixels_min<I64, INDEX, false>), dim3(pair_blocks), dim3(256), 0, stream, rects_xy, values, (i64)n,    \
                            (int)w1, (int)h1, cell, filter, info);                                                                    \
  } while (0)
  if (wide) { if (index) GCP_PIXELS_MIN(true, true); else GCP_PIXELS_MIN(true, false); }
  else { if (index) GCP_PIXELS_MIN(false, true); else GCP_PIXELS_MIN(false, false); }
#undef GCP_PIXELS_MIN
  GCP_HIP(hipGetLastError());
  const int bands = (int)((h1 + kBandRows - 1) / kBandRows);
  const dim3 rgrid((unsigned)((w1 + kColTile - 1) / kColTile), (unsigned)bands);
  hipLaunchKernelGGL((k_pixels_readout<false, false, false>), rgrid, dim3(256), 0, stream, (const unsigned*)cell, (int)w1, (int)h1, bands, cnt,
                     (const int*)nullptr, (i64)0, (void*)nullptr, (float*)nullptr, (int*)nullptr);
  GCP_HIP(hipGetLastError());
  const int st = gcp_exclusive_scan_i32(cnt, off, L.blocks, base + L.scan, gcp_scan_i32_workspace_bytes(L.blocks), stream_);
  if (st != GCP_OK) return st;
#define GCP_PIXELS_OUT(I64, INDEX)                                                                                                     \
  hipLaunchKernelGGL((k_pixels_readout<true, I64, INDEX>), rgrid, dim3(256), 0, stream, (const unsigned*)cell, (int)w1, (int)h1, bands, \
                     (int*)nullptr, (const int*)off, (i64)capacity, out_xy, out_val, info)
  if (wide) { if (index) GCP_PIXELS_OUT(true, true); else GCP_PIXELS_OUT(true, false); }
  else { if (index) GCP_PIXELS_OUT(false, true); else GCP_PIXELS_OUT(false, false); }
#undef GCP_PIXELS_OUT
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

}  // extern "C"
