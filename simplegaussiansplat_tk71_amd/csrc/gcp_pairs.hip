// gcp_pairs.hip — the rect list of the reference, taken apart again (rows a5 / a6).
//
// `_create_alpha_brend(rects, anti_opacity, flag)` (reference: gs_model.py:544-566) receives the splat-pixel pairs of one
// camera as a flat list of pixel coordinates and sorts it by pixel — two radix sorts, a gather and a scan over M pairs.
// But the list is never arbitrary: `_create_rects` (gs_model.py:480-482 -> Utilities.make_rect_points_parallel,
// uitility.py:336-366) writes it as a concatenation of row-major boxes, one per Gaussian in depth order, and that
// structure is enough to avoid the sort: cut the list back into rectangles, bin the rectangles into 16x16 tiles (K ~ 3
// entries each) and let every pixel walk its tile's list (gcp_pairs_scan_boxes, gcp_raster.hip).  This file does the
// cutting, for ANY list:
//   a ROW      is a maximal run of consecutive elements (x, y), (x + 1, y), (x + 2, y), ...
//   a RECTANGLE is a maximal run of consecutive rows with the same first x and the same length whose y grows by one.
// By construction every rectangle is exactly the row-major expansion of [x0, x1] x [y0, y1], the rectangles in list order
// reproduce the list, and no pixel occurs twice inside one — which is all the tile walk needs: the per-pixel order of the
// pairs is the order of the rectangles.  Two Gaussians whose boxes happen to continue each other come out as one rectangle
// (their pixels are disjoint: nothing changes); a list that is NOT made of boxes comes out as a great many tiny rectangles,
// the caller sees their number and takes the general route (stable radix sort) instead.
//
// Both cuts are stable stream compactions without atomics on the data path: per-tile counts, one exclusive scan of the
// counts, ranked writes.  The first cut reads the M-sized list ONCE: the counting pass parks each tile's few row records
// (about 4096 / 13 of them) in a slot region of the tile's own, and the write pass moves them to their final places.
// (A single-pass version with a decoupled look-back over the tile counts was built first and measured at 1.8 ms for
// 1.65e8 pairs: 1500 tiles start together, every one of them has to add up the counts of all the others in front of it,
// 64 per fabric round trip — the chain, not the data, set the pace.  This form takes 0.4 ms.)

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gcp_device.hpp"
#include "grouped_cumprod_hip.h"

namespace {
using namespace gcp;

constexpr int kRowsPerThread = 4;                 // 16 elements per thread
constexpr int kElemTile = 1024 * kRowsPerThread;  // 4096 elements per block
constexpr int kSlotRowsMax = kElemTile;           // row records a tile may park at most: every element may start a row (the carry rows
                                                  // of a chunked call, gs_model.py:611, are single pixels)
constexpr int kRowTile = 1024 * kRowsPerThread;   // rows per block in the second cut, 16 per thread

// Where tile t parks its row records.  A list of boxes parks ~ 4096 / (box width) records per tile, so the tiles of the
// boxes get `slot_rows` slots each (512 by default: 1 B of scratch per pair instead of 8; a tile that needs more marks the
// list "not boxes at this capacity" and the caller repeats with more or sorts); only the tiles that hold the single-pixel
// carry rows of a chunked call — [0, front_tiles) when they lead the list (gs_model.py:611), [back_tile0, n_tiles) when
// they trail it (:636) — get one slot per element.  0 <= front_tiles <= back_tile0 <= n_tiles.
// A tile that needs more than its slots (a stretch of narrow boxes in a list of wide ones) parks the excess in a POOL shared
// by all tiles: one atomic add on the pool's fill per overflowing tile reserves its run there (pool_off[tile]); where the
// records lie does not matter — the second pass moves them to places that depend on the counts alone — so the result stays
// deterministic.  Only when the pool runs out too is the list handed back (kNotBoxesSlots).
struct SlotLayout {
  i64 front_tiles, back_tile0;
  int slot_rows;
  i64 pool_rows;  // capacity of the shared pool, in records (0: none)
  __host__ __device__ i64 base(i64 t) const {
    const i64 a = t < front_tiles ? t : front_tiles;
    const i64 m = (t < back_tile0 ? t : back_tile0) - front_tiles;
    const i64 c = t - back_tile0;
    return a * kSlotRowsMax + (m > 0 ? m : 0) * slot_rows + (c > 0 ? c : 0) * kSlotRowsMax;
  }
  __host__ __device__ int cap(i64 t) const { return (t < front_tiles || t >= back_tile0) ? kSlotRowsMax : slot_rows; }
};

inline SlotLayout slot_layout(i64 n, i64 carry_front, i64 carry_back, int slot_rows, i64 pool_rows = 0) {
  const i64 n_tiles = (n + kElemTile - 1) / kElemTile;
  SlotLayout L;
  L.pool_rows = pool_rows > 0 ? pool_rows : 0;
  L.slot_rows = slot_rows < 1 ? 1 : (slot_rows > kSlotRowsMax ? kSlotRowsMax : slot_rows);
  carry_front = carry_front < 0 ? 0 : (carry_front > n ? n : carry_front);
  carry_back = carry_back < 0 ? 0 : (carry_back > n ? n : carry_back);
  L.front_tiles = (carry_front + kElemTile - 1) / kElemTile;
  L.back_tile0 = carry_back > 0 ? (n - carry_back) / kElemTile : n_tiles;
  if (L.front_tiles > n_tiles) L.front_tiles = n_tiles;
  if (L.back_tile0 < L.front_tiles) L.back_tile0 = L.front_tiles;
  if (L.back_tile0 > n_tiles) L.back_tile0 = n_tiles;
  return L;
}

// info[4] of the first cut: why the list cannot be walked as boxes
constexpr int kNotBoxesRange = 1;     // a coordinate the walk cannot take (x >= 10000: the reference's key merges pixels; y >= 2^17)
constexpr int kNotBoxesSlots = 2;     // more rows than there is room for — in a tile's slots and the pool, or in the caller's row arrays

// Ranks of the set bits of m[0..ROWS) inside the block: thread t of wave w holds the flags of the 4 consecutive items
// w * 256 * ROWS + r * 256 + lane * 4 + k.  rank[r] = flags in front of the thread's row r; returns the block's count.
template <int ROWS>
__device__ __forceinline__ int block_ranks(const unsigned (&m)[ROWS], int (&rank)[ROWS], int* s_w) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int wtot = 0;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int c = __builtin_popcount(m[r]);
    const int inc = wave_incl_scan_i(c);
    rank[r] = wtot + inc - c;
    wtot += __builtin_amdgcn_readlane(inc, 63);
  }
  if (lane == 0) s_w[w] = wtot;
  __syncthreads();
  int woff = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < w) woff += s_w[j];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) rank[r] += woff;
  return s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// ---- cut 1, pass 1: row starts of one tile -> the tile's slots; count; coordinate range ---------------------------------
// One (x, y) of the list as int32s.  I64: the list holds int64 coordinates — what the reference's own
// make_rect_points_parallel returns (uitility.py:336-366: `start + ix` with ix from torch.arange) — read where they lie, 16
// bytes per element, instead of being converted by a pass of its own; a coordinate outside [0, 2^31) comes back as -1 and
// the list is refused like any with negative coordinates.
template <bool I64>
__device__ __forceinline__ int2 rect_at(const void* __restrict__ rects, i64 i) {
  if (!I64) return reinterpret_cast<const int2*>(rects)[i];
  typedef long long ll2 __attribute__((ext_vector_type(2), aligned(8)));  // a view of an int64 tensor is 8-byte aligned, no more
  const ll2 q = reinterpret_cast<const ll2*>(rects)[i];
  const bool ok = ((unsigned long long)q.x | (unsigned long long)q.y) < 0x80000000ull;
  return ok ? make_int2((int)q.x, (int)q.y) : make_int2(-1, -1);
}

// FULL (compile-time: two straight-line bodies, chosen per block): the tile lies wholly inside the list — every tile but the last
template <bool I64, bool FULL>
__device__ __forceinline__ void rect_rows_local_tile(const void* __restrict__ rects, i64 n, const SlotLayout& L, int2* __restrict__ slots,
                                                     int2* __restrict__ pool, unsigned long long* __restrict__ pool_fill,
                                                     int* __restrict__ pool_off, int* __restrict__ cnt, int* __restrict__ info,
                                                     int* s_w, int* s_mx, int* s_my, int* s_mn, long long& s_pool) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 tile = blockIdx.x;
  const i64 base = tile * kElemTile + (i64)w * (256 * kRowsPerThread);
  int2 e[kRowsPerThread][4];
  unsigned m[kRowsPerThread];
  int mx = 0, my = 0, mn = 0x7fffffff;
  const bool vec = (((uintptr_t)rects) & 15u) == 0;
  int2 before[kRowsPerThread];  // the element in front of each row of the wave (what its lane 0 compares with), fetched with the row loads
  constexpr bool full = FULL;
  if (full) {
    // straight-line: all loads of the lane — its sixteen elements and, at a wave-uniform address (one broadcast line), the
    // element in front of each of its rows — are issued before anything waits.  (Behind a bounds test per element, or with
    // lane 0 fetching its neighbour inside the compare loop, the compiler waits load by load: gfx950 counts loads in one
    // in-order counter and cannot know how many a branch issued.)
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const i64 p = base + r * 256 + lane * 4;
      if (I64) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e[r][k] = rect_at<true>(rects, p + k);
      } else if (vec) {
        const int4 a = *reinterpret_cast<const int4*>(reinterpret_cast<const int2*>(rects) + p);
        const int4 b = *reinterpret_cast<const int4*>(reinterpret_cast<const int2*>(rects) + p + 2);
        e[r][0] = make_int2(a.x, a.y); e[r][1] = make_int2(a.z, a.w); e[r][2] = make_int2(b.x, b.y); e[r][3] = make_int2(b.z, b.w);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) e[r][k] = rect_at<false>(rects, p + k);
      }
      const i64 p0 = base + r * 256;
      const int2 q = rect_at<I64>(rects, p0 > 0 ? p0 - 1 : 0);
      before[r] = p0 > 0 ? q : make_int2(0x7ffffff0, 0x7ffffff0);  // (nothing continues the first element of the list)
    }
  } else {
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const i64 p = base + r * 256 + lane * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) e[r][k] = (p + k < n) ? rect_at<I64>(rects, p + k) : make_int2(0x7ffffff0, 0x7ffffff0);
      const i64 p0 = base + r * 256;
      before[r] = (p0 > 0 && p0 <= n) ? rect_at<I64>(rects, p0 - 1) : make_int2(0x7ffffff0, 0x7ffffff0);
    }
  }
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    // the element in front of the lane's four: the left neighbour's last one; lane 0 of a row takes the fetched one
    int2 prev;
    prev.x = dpp_i<0x138, 0xf>(0x7ffffff0, e[r][3].x);
    prev.y = dpp_i<0x138, 0xf>(0x7ffffff0, e[r][3].y);
    if (lane == 0) prev = before[r];
    m[r] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (full || p + k < n) {
        const bool cont = e[r][k].x == prev.x + 1 && e[r][k].y == prev.y;
        m[r] |= (cont ? 0u : 1u) << k;
        prev = e[r][k];
        mx = max(mx, e[r][k].x); my = max(my, e[r][k].y); mn = min(mn, min(e[r][k].x, e[r][k].y));
      }
    }
  }
  int rank[kRowsPerThread];
  const int count = block_ranks<kRowsPerThread>(m, rank, s_w);
  int2* const mine = slots + L.base(tile);
  const int cap = L.cap(tile);
  bool fits = count <= cap;  // block-uniform
  int2* extra = nullptr;     // where the records beyond the tile's own slots go
  if (!fits && L.pool_rows > 0) {
    if (threadIdx.x == 0) {
      // (look before the atomic: a list that is not made of boxes overflows in EVERY tile, and the pool is full after the
      // first few hundred of them)
      long long at = -1;
      if (__hip_atomic_load(pool_fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)L.pool_rows) {
        at = (long long)atomicAdd(pool_fill, (unsigned long long)(count - cap));
        if (at + (count - cap) > L.pool_rows) at = -1;
      }
      s_pool = at;
      if (at >= 0) pool_off[tile] = (int)at;
    }
    __syncthreads();
    if (s_pool >= 0) { fits = true; extra = pool + s_pool; }
  }
  if (count <= cap) {
    // the usual case, block-uniform: everything goes to the tile's own slots — a wave-uniform base and a 32-bit slot number,
    // nothing to choose per record (the kernel spends two thirds of its time issuing vector ALU instructions: 16 elements
    // per thread, each with its compare chain, its share of the coordinate range and this masked store)
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const int p32 = (int)(base + r * 256) + lane * 4;  // (n < 2^31)
      unsigned o = (unsigned)rank[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((m[r] >> k) & 1u) {
          mine[o] = make_int2(p32 + k, e[r][k].x | (e[r][k].y << 14));  // x < 10000 < 2^14, y < 2^17 (key < 2^31)
          ++o;
        }
      }
    }
  } else if (fits) {
#pragma unroll
    for (int r = 0; r < kRowsPerThread; ++r) {
      const i64 p = base + r * 256 + lane * 4;
      int o = rank[r];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((m[r] >> k) & 1u) {
          const int2 rec = make_int2((int)(p + k), e[r][k].x | (e[r][k].y << 14));
          if (o < cap) mine[o] = rec;
          else extra[o - cap] = rec;
          ++o;
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {  // coordinate range of the list (integer max / min: order-independent)
    mx = max(mx, __shfl_xor(mx, o)); my = max(my, __shfl_xor(my, o)); mn = min(mn, __shfl_xor(mn, o));
  }
  if (lane == 0) { s_mx[w] = mx; s_my[w] = my; s_mn[w] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cnt[tile] = fits ? count : 0;
    // (look before the atomic, as below: a list that is not made of boxes overflows in EVERY tile)
    if (!fits && !(__hip_atomic_load(info + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kNotBoxesSlots)) atomicOr(info + 4, kNotBoxesSlots);
    // Look before the atomic: 40 000 blocks hitting three words with an atomic each serialise at the L2 (1.1 of this
    // kernel's 1.4 ms when first written that way); the range settles after the first few blocks and the rest only read.
    const int bx = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])), by = max(max(s_my[0], s_my[1]), max(s_my[2], s_my[3]));
    const int bn = min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3]));
    // x >= 10000: the reference's pixel key y * 10000 + x (gs_model.py:538-541) then runs DIFFERENT pixels together —
    // (10000, 0) and (0, 1) share key 10000 — and only the key-based sort route reproduces its groups; the walk groups by
    // pixel.  y >= 2^17: does not fit the packed slot record (nor the reference's int32 key).
    if ((bx >= 10000 || by >= (1 << 17)) && !(__hip_atomic_load(info + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kNotBoxesRange))
      atomicOr(info + 4, kNotBoxesRange);
    if (bx > __hip_atomic_load(info + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(info + 1, bx);
    if (by > __hip_atomic_load(info + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(info + 2, by);
    if (bn < __hip_atomic_load(info + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(info + 3, bn);
  }
}

template <bool I64>
__global__ __launch_bounds__(256) void k_rect_rows_local(const void* __restrict__ rects, i64 n, const SlotLayout L,
                                                         int2* __restrict__ slots /*tile t: [L.base(t), + L.cap(t)): {index, x | y << 14}*/,
                                                         int2* __restrict__ pool, unsigned long long* __restrict__ pool_fill,
                                                         int* __restrict__ pool_off /*[tiles]*/, int* __restrict__ cnt,
                                                         int* __restrict__ info /*[5]: rows, max x, max y, min, not-boxes flags*/) {
  __shared__ int s_w[4], s_mx[4], s_my[4], s_mn[4];
  __shared__ long long s_pool;
  if (((i64)blockIdx.x + 1) * kElemTile <= n)
    rect_rows_local_tile<I64, true>(rects, n, L, slots, pool, pool_fill, pool_off, cnt, info, s_w, s_mx, s_my, s_mn, s_pool);
  else
    rect_rows_local_tile<I64, false>(rects, n, L, slots, pool, pool_fill, pool_off, cnt, info, s_w, s_mx, s_my, s_mn, s_pool);
}

// The rows of the first cut as the later stages read them: split — the public form, row_start int[rows + 1] (+ sentinel n)
// and row_xy int2[rows] — or packed, the one-call cut's scratch form: the 8-byte slot records as they are,
// {first element, x | y << 14}, and a sentinel record {n, 0}: 8 B per row instead of 12.
struct Rows {
  const int* start;
  const int2* xy;
  const int2* packed;
  __device__ __forceinline__ int first(i64 r) const { return packed ? packed[r].x : start[r]; }
  __device__ __forceinline__ int2 at(i64 r) const {
    if (!packed) return xy[r];
    const int q = packed[r].y;
    return make_int2(q & 0x3fff, (int)((unsigned)q >> 14));
  }
};

// ---- cut 1, pass 2: the parked records to their final places -----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rect_rows_gather(const int2* __restrict__ slots, const SlotLayout L, const int2* __restrict__ pool,
                                                          const int* __restrict__ pool_off, const int* __restrict__ off /*[tiles + 1]*/, i64 n,
                                                          i64 n_tiles, i64 row_capacity, int* __restrict__ row_start,
                                                          int2* __restrict__ row_xy, int2* __restrict__ row_packed /*or: the packed form*/,
                                                          int* __restrict__ info) {
  const i64 tile = blockIdx.x;
  const int flags = info[4];  // written by the launch before this one
  if (flags != 0 || (i64)off[n_tiles] + 1 > row_capacity) {
    // not a list the walk can take (coordinates, a tile with more rows than slots) or more rows than the caller made room
    // for: nothing is written
    if (tile == 0 && threadIdx.x == 0) { info[0] = off[n_tiles]; if (flags == 0) info[4] = kNotBoxesSlots; }
    return;
  }
  const int o0 = off[tile], c = off[tile + 1] - o0;
  const int2* const mine = slots + L.base(tile);
  const int cap = L.cap(tile);
  const int2* const extra = c > cap ? pool + pool_off[tile] : nullptr;
  for (int i = threadIdx.x; i < c; i += 256) {
    const int2 q = i < cap ? mine[i] : extra[i - cap];
    if (row_packed) { row_packed[o0 + i] = q; continue; }
    row_start[o0 + i] = q.x;
    row_xy[o0 + i] = make_int2(q.y & 0x3fff, (int)((unsigned)q.y >> 14));
  }
  if (tile == n_tiles - 1 && threadIdx.x == 0) {
    info[0] = off[n_tiles];
    if (row_packed) row_packed[off[n_tiles]] = make_int2((int)n, 0);
    else row_start[off[n_tiles]] = (int)n;  // sentinel: the end of the last row
  }
}

// ---- cut 2: rows -> rectangles (count, then write) -------------------------------------------------------------------------
// rows_dev: NULL (n_rows is the host's count), or the first cut's info on the device — the grid then covers the row
// CAPACITY, the count is info[0], and a list the first cut refused (info[4] != 0) has no rows at all.
template <bool WRITE, bool PACKED>
__global__ __launch_bounds__(256) void k_rows_rectangles(const Rows rows, i64 n_rows,
                                                         const int* __restrict__ rows_dev, int* __restrict__ cnt, const int* __restrict__ off,
                                                         int* __restrict__ rect_row, int* __restrict__ info) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 tile = blockIdx.x;
  if (rows_dev) n_rows = rows_dev[4] ? 0 : (i64)rows_dev[0];
  if (tile * kRowTile >= n_rows && !(tile == 0)) {  // a block of the capacity grid behind the last row: nothing to find
    if (!WRITE && threadIdx.x == 0) cnt[tile] = 0;
    return;
  }
  // 16 rows per thread as four groups of four consecutive ones (the item order of block_ranks): of rows p - 1 .. p + 4 the
  // first element, and of p - 1 .. p + 3 the first pixel — all loads of the thread are issued before the first compare
  const i64 base = tile * kRowTile + (i64)w * (256 * kRowsPerThread);
  int3 q[kRowsPerThread][6];
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    const i64 p = base + r * 256 + lane * 4;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      // straight-line loads (a load behind a branch makes every later one wait for it, the lesson of the walk): an index
      // outside the rows is clamped to one inside, its value never used
      i64 j = p - 1 + k;
      j = j < 0 ? 0 : (j > n_rows ? n_rows : j);
      if (PACKED) {
        const int2 v = rows.packed[j];  // (the sentinel {n, 0} sits at n_rows)
        q[r][k] = make_int3(v.x, v.y & 0x3fff, (int)((unsigned)v.y >> 14));
      } else {
        const int2 xy = rows.xy[j < n_rows ? j : (n_rows > 0 ? n_rows - 1 : 0)];
        q[r][k] = make_int3(rows.start[j], xy.x, xy.y);
      }
    }
  }
  unsigned m[kRowsPerThread];
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    m[r] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (p + k < n_rows) {
        // row p + k continues the rectangle of the row in front of it: same first column, same length, the next line
        const bool cont = p + k > 0 && q[r][k + 1].y == q[r][k].y && q[r][k + 2].x - q[r][k + 1].x == q[r][k + 1].x - q[r][k].x &&
                          q[r][k + 1].z == q[r][k].z + 1;
        m[r] |= (cont ? 0u : 1u) << k;
      }
    }
  }
  int rank[kRowsPerThread];
  const int count = block_ranks<kRowsPerThread>(m, rank, s_w);
  if (!WRITE) {
    if (threadIdx.x == 0) cnt[tile] = count;
    return;
  }
  const int o0 = off[tile];
#pragma unroll
  for (int r = 0; r < kRowsPerThread; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    int o = o0 + rank[r];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((m[r] >> k) & 1u) rect_row[o++] = (int)(p + k);  // (as many rectangles as rows at most: the caller sized it so)
  }
  if (threadIdx.x == 0 && tile == (n_rows > 0 ? (n_rows - 1) / kRowTile : 0)) {  // the block that holds the last row
    info[0] = o0 + count;
    rect_row[o0 + count] = (int)n_rows;  // sentinel
  }
}

// rectangle b = rows [rect_row[b], rect_row[b + 1]): its box, and where its pairs start in the list
// rects_dev: NULL (n_rects is the host's count), or the second cut's info on the device: the grid covers `capacity` + 1
// entries and the count is rects_dev[0].  tile_cnt (with rects_dev): the number of 16x16 tiles every rectangle touches —
// the boxes lie inside [0, max x] x [0, max y] by construction, so the binning's clamp is the identity and its counting
// pass (gcp_bin_tiles_count) is this line — zeros behind the last rectangle, and their 64-bit total.
__global__ __launch_bounds__(256) void k_rectangle_boxes(const int* __restrict__ rect_row, const Rows rows, i64 n_rects, i64 n,
                                                         const int* __restrict__ rects_dev,
                                                         i64 capacity, int* __restrict__ start_xy, int* __restrict__ end_xy,
                                                         int* __restrict__ box_off, int* __restrict__ tile_cnt,
                                                         unsigned long long* __restrict__ block_sum /*[gridDim.x], with tile_cnt*/) {
  __shared__ int s_w[4];
  if (rects_dev) n_rects = rects_dev[0];
  const bool fits = !rects_dev || n_rects <= capacity;  // more rectangles than the caller made room for: flagged by the finishing kernel
  const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  if (fits && b < n_rects) {
    const int r0 = rect_row[b], r1 = rect_row[b + 1];
    int2 q;
    int f0, f1;
    if (rows.packed) {  // (wave-uniform; both records with one 8-byte load each)
      const int2 v0 = rows.packed[r0], v1 = rows.packed[r0 + 1];
      q = make_int2(v0.y & 0x3fff, (int)((unsigned)v0.y >> 14));
      f0 = v0.x; f1 = v1.x;
    } else {
      q = rows.xy[r0];
      f0 = rows.start[r0]; f1 = rows.start[r0 + 1];
    }
    const int x1 = q.x + (f1 - f0) - 1, y1 = q.y + (r1 - r0) - 1;
    start_xy[2 * b] = q.x;
    start_xy[2 * b + 1] = q.y;
    end_xy[2 * b] = x1;
    end_xy[2 * b + 1] = y1;
    box_off[b] = f0;
    c = ((x1 >> 4) - (q.x >> 4) + 1) * ((y1 >> 4) - (q.y >> 4) + 1);
  } else if (fits && b == n_rects) {
    box_off[b] = (int)n;
  }
  if (tile_cnt) {
    if (b < capacity) tile_cnt[b] = c;
    // the block's share of the 64-bit total (a rectangle touches a few thousand tiles at most: an int holds 256 of them)
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (unsigned long long)((long long)s_w[0] + s_w[1] + s_w[2] + s_w[3]);
  }
}

// info8 of gcp_rects_cut from what the stages left on the device
__global__ __launch_bounds__(256) void k_cut_finish(const int* __restrict__ info_rows /*[5]*/, const int* __restrict__ info_rects /*[2]*/,
                                                    const unsigned long long* __restrict__ block_sum, i64 n_blocks, i64 rect_capacity,
                                                    int* __restrict__ info8) {
  __shared__ unsigned long long s_part[256];
  unsigned long long part = 0;  // integer adds: the same total whatever the order
  for (i64 j0 = threadIdx.x; j0 < n_blocks; j0 += 8 * 256) {  // eight loads in flight per thread
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (j0 + u * 256 < n_blocks) ? block_sum[j0 + u * 256] : 0ull;
#pragma unroll
    for (int u = 0; u < 8; ++u) part += v[u];
  }
  s_part[threadIdx.x] = part;
  __syncthreads();
  if (threadIdx.x != 0) return;
  unsigned long long k = 0;
  for (int j = 0; j < 256; ++j) k += s_part[j];
  int flags = info_rows[4];
  const int n_rects = info_rects[0];
  if (flags == 0 && (i64)n_rects > rect_capacity) flags |= 4;
  if (flags == 0 && k > 0x7fffffffull) flags |= 8;
  info8[0] = info_rows[0]; info8[1] = info_rows[1]; info8[2] = info_rows[2]; info8[3] = info_rows[3];
  info8[4] = flags;
  info8[5] = n_rects;
  info8[6] = flags ? 0 : (int)k;
  info8[7] = 0;
}

inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" {

static size_t rows_ws_bytes(i64 n, const SlotLayout& L) {
  const i64 t = (n > 0 ? n + kElemTile - 1 : kElemTile) / kElemTile;
  return align256((size_t)(L.base(t) + 1) * sizeof(int2)) + align256((size_t)(L.pool_rows + 1) * sizeof(int2)) + 256 +
         3 * align256((size_t)(t + 1) * sizeof(int)) + gcp_scan_i32_workspace_bytes(t);
}

// every tile with one slot per element: a list of anything, 8 B of scratch per element
size_t gcp_rects_rows_workspace_bytes(int64_t n) { return rows_ws_bytes(n, slot_layout(n, n, 0, kSlotRowsMax)); }

// the first cut; `slots_out` (optional) receives the slot region's address — dead once this call's launches are through
// pool_fill_in: the one-call cut's own pool counter, cleared by it together with `info` (one fill for all its state words)
static int rects_rows_impl(const void* rects_xy, bool wide, int64_t n, const SlotLayout& L, int64_t row_capacity, int32_t* row_start,
                           int32_t* row_xy, int32_t* info, void* ws, size_t ws_bytes, void* stream_, void** slots_out = nullptr,
                           int2* row_packed = nullptr, unsigned long long* pool_fill_in = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || n > 0x7fffffffLL || row_capacity < 1 || !info) return GCP_ERR_INVALID_ARGUMENT;
  if (!pool_fill_in) GCP_HIP(hipMemsetAsync(info, 0, 5 * sizeof(int), stream));    // rows, max x, max y, (min), not-boxes flags
  GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)(info + 3), 0x7fffffff, 1, stream));   // min coordinate
  if (n == 0) return GCP_OK;
  if (!rects_xy || (!row_packed && (!row_start || !row_xy)) || !ws) return GCP_ERR_INVALID_ARGUMENT;
  const i64 n_tiles = (n + kElemTile - 1) / kElemTile;
  if (ws_bytes < rows_ws_bytes(n, L)) return GCP_ERR_WORKSPACE;
  char* p = (char*)ws;
  int2* slots = (int2*)p; p += align256((size_t)(L.base(n_tiles) + 1) * sizeof(int2));
  int2* pool = (int2*)p; p += align256((size_t)(L.pool_rows + 1) * sizeof(int2));
  unsigned long long* pool_fill = pool_fill_in ? pool_fill_in : (unsigned long long*)p; p += 256;
  int* pool_off = (int*)p; p += align256((size_t)(n_tiles + 1) * sizeof(int));
  int* cnt = (int*)p; p += align256((size_t)(n_tiles + 1) * sizeof(int));
  int* off = (int*)p; p += align256((size_t)(n_tiles + 1) * sizeof(int));
  if (slots_out) *slots_out = slots;
  if (L.pool_rows > 0 && !pool_fill_in) GCP_HIP(hipMemsetAsync(pool_fill, 0, sizeof(unsigned long long), stream));
  if (wide) hipLaunchKernelGGL((k_rect_rows_local<true>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rects_xy, (i64)n, L, slots, pool, pool_fill,
                               pool_off, cnt, info);
  else hipLaunchKernelGGL((k_rect_rows_local<false>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rects_xy, (i64)n, L, slots, pool, pool_fill,
                          pool_off, cnt, info);
  GCP_HIP(hipGetLastError());
  const int st = gcp_exclusive_scan_i32(cnt, off, n_tiles, p, gcp_scan_i32_workspace_bytes(n_tiles), stream_);
  if (st != GCP_OK) return st;
  hipLaunchKernelGGL(k_rect_rows_gather, dim3((unsigned)n_tiles), dim3(256), 0, stream, (const int2*)slots, L, (const int2*)pool,
                     (const int*)pool_off, (const int*)off, (i64)n, n_tiles, (i64)row_capacity, row_start, (int2*)row_xy, row_packed, info);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_rects_rows(const int32_t* rects_xy, int64_t n, int64_t row_capacity, int32_t* row_start, int32_t* row_xy, int32_t* info,
                   void* ws, size_t ws_bytes, void* stream) {
  return rects_rows_impl(rects_xy, false, n, slot_layout(n, n, 0, kSlotRowsMax), row_capacity, row_start, row_xy, info, ws, ws_bytes, stream);
}

int gcp_rects_rows_i64(const int64_t* rects_xy, int64_t n, int64_t row_capacity, int32_t* row_start, int32_t* row_xy, int32_t* info,
                       void* ws, size_t ws_bytes, void* stream) {
  return rects_rows_impl(rects_xy, true, n, slot_layout(n, n, 0, kSlotRowsMax), row_capacity, row_start, row_xy, info, ws, ws_bytes, stream);
}

// rows a list may have and still be taken for boxes: a list of boxes has ~ n / (box width) of them, a list of unrelated
// coordinates ~ n; the line is drawn at one row per two elements (whether the rectangles are worth walking is the
// caller's second test, on their count)
int64_t gcp_rects_rows_capacity(int64_t n) { return (n > 0 ? n : 0) / 2 + 2; }

size_t gcp_rows_rectangles_workspace_bytes(int64_t n_rows) {
  const int64_t t = (n_rows > 0 ? n_rows + kRowTile - 1 : kRowTile) / kRowTile;
  return 2 * align256((size_t)(t + 1) * sizeof(int)) + gcp_scan_i32_workspace_bytes(t);
}

// n_rows: the row count, or — with rows_dev, the first cut's info on the device — the row CAPACITY the grid has to cover
static int rows_rectangles_impl(const Rows rows, int64_t n_rows, const int* rows_dev, int32_t* rect_row,
                                int32_t* info, void* ws, size_t ws_bytes, void* stream_, bool info_cleared = false) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows < 0 || n_rows > 0x7fffffffLL || !info) return GCP_ERR_INVALID_ARGUMENT;
  if (!info_cleared) GCP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int), stream));  // rectangles, (unused)
  if (n_rows == 0) return GCP_OK;
  if ((!rows.packed && (!rows.start || !rows.xy)) || !rect_row || !ws) return GCP_ERR_INVALID_ARGUMENT;
  const i64 n_tiles = (n_rows + kRowTile - 1) / kRowTile;
  if (ws_bytes < gcp_rows_rectangles_workspace_bytes(n_rows)) return GCP_ERR_WORKSPACE;
  char* p = (char*)ws;
  int* cnt = (int*)p; p += align256((size_t)(n_tiles + 1) * sizeof(int));
  int* off = (int*)p; p += align256((size_t)(n_tiles + 1) * sizeof(int));
  if (rows.packed) hipLaunchKernelGGL((k_rows_rectangles<false, true>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rows, (i64)n_rows, rows_dev,
                                      cnt, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  else hipLaunchKernelGGL((k_rows_rectangles<false, false>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rows, (i64)n_rows, rows_dev,
                          cnt, (const int*)nullptr, (int*)nullptr, (int*)nullptr);
  GCP_HIP(hipGetLastError());
  const int st = gcp_exclusive_scan_i32(cnt, off, n_tiles, p, gcp_scan_i32_workspace_bytes(n_tiles), stream_);
  if (st != GCP_OK) return st;
  if (rows.packed) hipLaunchKernelGGL((k_rows_rectangles<true, true>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rows, (i64)n_rows, rows_dev,
                                      (int*)nullptr, (const int*)off, rect_row, info);
  else hipLaunchKernelGGL((k_rows_rectangles<true, false>), dim3((unsigned)n_tiles), dim3(256), 0, stream, rows, (i64)n_rows, rows_dev,
                          (int*)nullptr, (const int*)off, rect_row, info);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_rows_rectangles(const int32_t* row_start, const int32_t* row_xy, int64_t n_rows, int32_t* rect_row, int32_t* info, void* ws,
                        size_t ws_bytes, void* stream) {
  return rows_rectangles_impl(Rows{row_start, (const int2*)row_xy, nullptr}, n_rows, nullptr, rect_row, info, ws, ws_bytes, stream);
}

int gcp_rectangle_boxes(const int32_t* rect_row, const int32_t* row_start, const int32_t* row_xy, int64_t n_rects, int64_t n,
                        int32_t* start_xy, int32_t* end_xy, int32_t* box_off, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rects < 0 || n < 0 || !box_off) return GCP_ERR_INVALID_ARGUMENT;
  if (n_rects > 0 && (!rect_row || !row_start || !row_xy || !start_xy || !end_xy)) return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_rectangle_boxes, dim3((unsigned)((n_rects + 1 + 255) / 256)), dim3(256), 0, stream, rect_row,
                     Rows{row_start, (const int2*)row_xy, nullptr}, (i64)n_rects, (i64)n, (const int*)nullptr, (i64)n_rects, start_xy, end_xy,
                     box_off, (int*)nullptr, (unsigned long long*)nullptr);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

// ---- the whole cut in one call, nothing read back in between ------------------------------------------------------------------
// rows -> rectangles -> boxes -> tiles per box -> their prefix sums, every count handed on in device memory; the caller reads
// info8 ONCE and knows everything the binning and the walk need.  Scratch: the slot region (8 B x slot_rows per 4096-pair
// tile; the rectangle list of the second cut re-uses it), the rows (12 B each, at most one per slot), a few words per tile —
// with 512 slots per tile 2.5 B per pair, plus 28 B per rectangle of capacity in the caller's arrays.
// rows the one-call cut makes room for: as many as it has slots (an eighth of the list at 512 per tile, + the carry rows) —
// the pool only evens out between tiles
static i64 cut_row_capacity(i64 n, const SlotLayout& L) { return L.base((n + kElemTile - 1) / kElemTile) + 1; }
// the pool of the one-call cut: a thirty-second of the list (0.25 B per pair) for the tiles that exceed their slots
static i64 cut_pool_rows(i64 n) { return n / 32 + 1024; }

struct CutWs {
  size_t rows_ws, row_start, row_xy, rect_ws, tile_cnt, scan_ws, info, total;
};
static CutWs cut_ws_layout(i64 n, const SlotLayout& L, i64 rect_capacity) {
  const i64 rc = cut_row_capacity(n, L);
  CutWs w;
  size_t o = 0;
  w.rows_ws = o; o += align256(rows_ws_bytes(n, L));
  w.row_start = o;                                              // (unused: the rows stay packed)
  w.row_xy = o; o += align256((size_t)(rc + 1) * sizeof(int2));  // packed rows + sentinel
  w.rect_ws = o; o += align256(gcp_rows_rectangles_workspace_bytes(rc));
  w.tile_cnt = o; o += align256((size_t)(rect_capacity + 1) * sizeof(int));
  w.scan_ws = o; o += align256(gcp_scan_i32_workspace_bytes(rect_capacity + 1));
  w.info = o; o += 256;   // int[5] rows info, int[2] rectangles info at +32, the pool's fill counter at +64: one fill clears them
  w.total = o; o += align256((size_t)((rect_capacity + 256) / 256 + 1) * sizeof(unsigned long long));  // per-block sums of the tile counts
  return w;
}

size_t gcp_rects_cut_workspace_bytes(int64_t n, int64_t carry_front, int64_t carry_back, int32_t slot_rows, int64_t rect_capacity) {
  if (n < 0 || rect_capacity < 0) return 0;
  const CutWs w = cut_ws_layout(n, slot_layout(n, carry_front, carry_back, slot_rows, cut_pool_rows(n)), rect_capacity);
  return w.total + align256((size_t)((rect_capacity + 256) / 256 + 1) * sizeof(unsigned long long));
}

int gcp_rects_cut(const void* rects_xy, int32_t rects_are_int64, int64_t n, int64_t carry_front, int64_t carry_back, int32_t slot_rows,
                  int64_t rect_capacity, int32_t* start_xy, int32_t* end_xy, int32_t* box_off, int32_t* tile_off, int32_t* info8,
                  void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || n > 0x7fffffffLL || rect_capacity < 1 || rect_capacity > 0x7ffffff0LL || !info8) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) {
    GCP_HIP(hipMemsetAsync(info8, 0, 8 * sizeof(int), stream));
    GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)(info8 + 3), 0x7fffffff, 1, stream));
    if (box_off) GCP_HIP(hipMemsetAsync(box_off, 0, sizeof(int), stream));
    if (tile_off) GCP_HIP(hipMemsetAsync(tile_off, 0, sizeof(int), stream));
    return GCP_OK;
  }
  if (!rects_xy || !start_xy || !end_xy || !box_off || !tile_off || !ws) return GCP_ERR_INVALID_ARGUMENT;
  const SlotLayout L = slot_layout(n, carry_front, carry_back, slot_rows, cut_pool_rows(n));
  const CutWs w = cut_ws_layout(n, L, rect_capacity);
  if (ws_bytes < gcp_rects_cut_workspace_bytes(n, carry_front, carry_back, slot_rows, rect_capacity) || ((uintptr_t)ws & 255u)) return GCP_ERR_WORKSPACE;
  char* const base = (char*)ws;
  int2* const row_packed = (int2*)(base + w.row_xy);
  const Rows rows{nullptr, nullptr, row_packed};
  int* const info_rows = (int*)(base + w.info);
  int* const info_rects = info_rows + 8;
  unsigned long long* const block_sum = (unsigned long long*)(base + w.total);
  unsigned long long* const pool_fill = (unsigned long long*)(base + w.info + 64);
  GCP_HIP(hipMemsetAsync(info_rows, 0, 256, stream));  // every state word of the call (info8 itself is written whole at the end)
  int* const tile_cnt = (int*)(base + w.tile_cnt);
  const i64 row_cap = cut_row_capacity(n, L);
  void* slots = nullptr;
  int st = rects_rows_impl(rects_xy, rects_are_int64 != 0, n, L, row_cap, nullptr, nullptr, info_rows, base + w.rows_ws, rows_ws_bytes(n, L), stream_, &slots,
                           row_packed, pool_fill);
  if (st != GCP_OK) return st;
  // the rectangle list (one int per rectangle, at most one per row, + the sentinel) re-uses the slot region: its records
  // have all been moved to the rows by now (8 B per slot there, 4 B per row here)
  int* const rect_row = (int*)slots;
  st = rows_rectangles_impl(rows, row_cap, info_rows, rect_row, info_rects, base + w.rect_ws, gcp_rows_rectangles_workspace_bytes(row_cap), stream_, true);
  if (st != GCP_OK) return st;
  const i64 box_blocks = (rect_capacity + 1 + 255) / 256;
  hipLaunchKernelGGL(k_rectangle_boxes, dim3((unsigned)box_blocks), dim3(256), 0, stream, (const int*)rect_row, rows, (i64)0, (i64)n,
                     (const int*)info_rects, (i64)rect_capacity, start_xy, end_xy, box_off, tile_cnt, block_sum);
  GCP_HIP(hipGetLastError());
  st = gcp_exclusive_scan_i32(tile_cnt, tile_off, rect_capacity, base + w.scan_ws, gcp_scan_i32_workspace_bytes(rect_capacity + 1), stream_);
  if (st != GCP_OK) return st;
  hipLaunchKernelGGL(k_cut_finish, dim3(1), dim3(256), 0, stream, (const int*)info_rows, (const int*)info_rects, (const unsigned long long*)block_sum,
                     box_blocks, (i64)rect_capacity, info8);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

}  // extern "C"
