// gcp_raster.hip — tile binning and fused per-pixel alpha blending.
//
// What the reference does around its scan (reference: gs_model.py:598-663, :666-692, :786-820):
// expand every Gaussian's box into M splat-pixel pairs, compute the Gaussian kernel per pair,
// radix-sort the M pixel keys, scan, un-sort (second radix sort), compact, blend, scatter-add
// into the image with atomics — ~20 passes over M-length arrays — and the same again (recomputed)
// in backward, plus a pair->Gaussian scatter_reduce.
//
// MI355X-first restatement, same results, no M-length array at all:
//   f2  Bin Gaussians (given in depth order) into 16x16-pixel tiles: count tiles per box, prefix
//       sum, emit (tile, gaussian) entries Gaussian-major, STABLE LSD radix sort on the tile id
//       (8-bit digits; ranks from wave ballots, waves ordered by an LDS prefix => deterministic), so every
//       tile's list is in depth order.  K entries (~3 per Gaussian) instead of M pairs (~166).
//   f1  One 256-thread block per tile, one pixel per lane.  The tile's list is staged through LDS
//       256 entries at a time; every lane walks it in depth order keeping its own transmittance
//       T (the exclusive grouped cumprod of the scan path, sequential per pixel), box-tests,
//       evaluates g = exp(-0.5 d Λ d^T), and accumulates colour.  Pixels are owned by lanes:
//       no atomics, deterministic.  A wave skips an entry when no lane is inside its box.
//       Backward walks each tile list BACK TO FRONT in chunks of kStageBwd entries: the exclusive suffix
//       sum S_k of (dL/dI . p) the reference gets from a flipped grouped cumsum (gs_model.py:716-722)
//       is carried in the normalised form S_k / (1 - o_k g_k) = T_k * R_k with the recurrence
//       R_{k-1} = R_k + o_k g_k ((dL/dI . l_k) - R_k)  — a convex combination: no accumulated sum is
//       subtracted — and T_k (the exclusive transmittance) comes from the entry behind it,
//       T_k = T_{k+1} / (1 - o_k g_k), restarted at every chunk from the checkpoint the forward kernel
//       saved for the chunk's end (T per pixel every kStageBwd entries, and behind the whole list); the
//       one chunk per pixel in which T underflows is recomputed front to back instead.  Every gradient
//       term is T_k times a bounded quantity, so its round-off is relative to the transmittance of
//       ITS OWN layer whatever the depth.
//       Per-pair gradients collapse to 7 per-lane values;
//       they are summed along each pixel row of the tile (16 lanes = one DPP row, 4 fused
//       v_add_f32_dpp) into LDS, one thread per entry folds the 16 rows (dy is constant along a
//       row) into the entry's slot in Gaussian-major order, and a last kernel sums each Gaussian's
//       few slots.  (An entry-parallel variant with the pair values parked in LDS was measured
//       slower, 1.88 vs 1.69 ms: it cannot skip the waves an entry does not touch.)
//       No float atomics anywhere => bitwise reproducible.
//   The per-pixel CSR (pixel offsets, pair->Gaussian, pair->rect index) that the scan API
//   consumes is exported by the same traversal (k_pixel_count / k_pixel_fill): bit-exact with
//   torch.sort(stable=True) of the reference's pixel keys (gs_model.py:546-547).
//
// Reference semantics kept: integer inclusive boxes (uitility.py:336-366), depth order = input
// order, pair dropped when its INCLUSIVE product is exactly 0 (gs_model.py:560,:575-578),
// image layout (H+1, W+1, 3) (gs_model.py:505), single chunk.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "gcp_device.hpp"
#include "grouped_cumprod_hip.h"

namespace {
using namespace gcp;

#ifndef GCP_STAGE_BWD
#define GCP_STAGE_BWD 32
#endif
constexpr int kTile = 16;           // tile edge in pixels; 256 pixels = one block, 4 rows per wave
// Measurement builds only (tools/build_variant.py -DGCP_TILE_SX=5|6): tiles of 32 x 8 / 64 x 4 pixels for the binning and the
// tile-list walk — the blend kernels are NOT valid in such a build.  DESIGN.md §3.4 "the walk's access shape".
#ifndef GCP_TILE_SX
#define GCP_TILE_SX 4
#endif
// -DGCP_TILE_SX=5 -DGCP_TILE_SY=4: "super-tiles" of 32 x 16 pixels walked by 512-thread blocks — eight waves of 4 rows x 16
// columns each, two side by side — so that both pieces of a box row that crosses a 16-pixel column boundary are read and
// written by one block in the same round.
#ifndef GCP_TILE_SY
#define GCP_TILE_SY (8 - GCP_TILE_SX)
#endif
constexpr int kTileSX = GCP_TILE_SX, kTileSY = GCP_TILE_SY;
constexpr int kTileW = 1 << kTileSX, kTileH = 1 << kTileSY;
constexpr int kWalkThreads = kTileW * kTileH;           // one lane per pixel of the walk's tile
constexpr bool kWalkSuper = kWalkThreads == 512;
#ifndef GCP_WALK_DENSE
#define GCP_WALK_DENSE 0  // measurement builds: the dense walk below (one wave per tile of any shape up to 1024 pixels)
#endif
static_assert(GCP_WALK_DENSE || kWalkThreads == 256 || (kTileSX == 5 && kTileSY == 4), "walk tiles: 256 pixels, or 32 x 16 super-tiles");
constexpr int kStage = 256;         // list entries staged per LDS round (forward)
constexpr int kStageBwd = GCP_STAGE_BWD;       // (backward; LDS also holds the per-pixel-row partial sums)
constexpr int kCkpt = kStageBwd;               // the forward saves every pixel's transmittance every kCkpt list entries
static_assert(kStage % kCkpt == 0 && 64 % kCkpt == 0 && kCkpt <= 32, "checkpoints fall on hit-word boundaries");
constexpr int kGradVals = 9;        // per (tile, Gaussian) slot: go, gl0..2, S(c dx), S(c dy), S(c dx dx), S(c dx dy), S(c dy dy)
constexpr int kRowVals = 7;         // per pixel row in LDS: go, gl0..2, S(c), S(c dx), S(c dx dx)   (dy is constant along a row)
constexpr int kRowSlots = 8;        // LDS slots per pixel row (the transposed reduction below leaves 8 values in 8 lane classes)
constexpr int kScanChunk = 2048;    // ints per prefix-sum block

// sum over each 16-lane DPP row (= one pixel row of the tile); valid in lanes 15, 31, 47, 63
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_f<0x111, 0xf>(0.0f, v);
  v += dpp_f<0x112, 0xf>(0.0f, v);
  v += dpp_f<0x114, 0xf>(0.0f, v);
  v += dpp_f<0x118, 0xf>(0.0f, v);
  return v;
}

// One step of the transposed reduction: lanes with `second` false keep value a, the others keep b; each adds its
// partner's copy of the value it keeps (partner = DPP pattern CTRL, which must flip the class bit).
template <int CTRL>
__device__ __forceinline__ float xchg_sum(bool second, float a, float b) {
  const float keep = second ? b : a, send = second ? a : b;
  return keep + dpp_f<CTRL, 0xf>(0.0f, send);
}

struct TileGrid { int tx, ty; };
inline TileGrid tile_grid(int W, int H) { return {(W + 1 + kTileW - 1) / kTileW, (H + 1 + kTileH - 1) / kTileH}; }

// ------------------------------------------------------------------------------------------
// Exclusive prefix sum of int32 (out has n+1 entries, out[n] = total).  Two small launches.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan_256(int v, int* s_w, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_incl_scan_i(v);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  int woff = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int t = s_w[j]; if (j < w) woff += t; tot += t; }
  total = tot;
  __syncthreads();
  return woff + inc - v;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const int* in, int* bsum, i64 n) {
  __shared__ int s_w[4];
  const i64 base = (i64)blockIdx.x * kScanChunk + (i64)threadIdx.x * 8;
  int s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) if (base + k < n) s += in[base + k];
  int total;
  block_excl_scan_256(s, s_w, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// second (last) launch: every block first reduces the block sums in front of it (<= a few thousand ints,
// L2-resident) to get its own offset — cheaper than a third launch for the block-sum scan
__global__ __launch_bounds__(256) void k_scan_apply(const int* in, const int* bsum, int* out, i64 n, i64 nb) {
  __shared__ int s_w[4];
  int part = 0;
  for (i64 j = threadIdx.x; j < (i64)blockIdx.x; j += 256) part += bsum[j];
  int boff;
  block_excl_scan_256(part, s_w, boff);  // boff = sum of all parts = offset of this block
  const i64 base = (i64)blockIdx.x * kScanChunk + (i64)threadIdx.x * 8;
  int v[8];
  int s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  int total;
  int ex = block_excl_scan_256(s, s_w, total) + boff;
#pragma unroll
  for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
  if ((i64)blockIdx.x == nb - 1 && threadIdx.x == 0) out[n] = boff + total;
}

// host: ws needs ceil(n/2048) ints
int launch_excl_scan(const int* in, int* out, i64 n, int* ws, hipStream_t stream) {
  if (n <= 0) {
    GCP_HIP(hipMemsetAsync(out, 0, sizeof(int), stream));
    return GCP_OK;
  }
  const i64 nb = (n + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(256), 0, stream, in, ws, n);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, in, (const int*)ws, out, n, nb);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

// ------------------------------------------------------------------------------------------
// f2: tile binning
// ------------------------------------------------------------------------------------------
struct Box { int x0, y0, x1, y1; };
__device__ __forceinline__ bool load_box(const int* start, const int* end, i64 g, int W, int H, Box& b) {
  b.x0 = max(start[2 * g], 0);
  b.y0 = max(start[2 * g + 1], 0);
  b.x1 = min(end[2 * g], W);
  b.y1 = min(end[2 * g + 1], H);
  return b.x1 >= b.x0 && b.y1 >= b.y0;
}

__global__ __launch_bounds__(256) void k_tile_count(const int* start, const int* end, i64 n, int W, int H, int* cnt,
                                                     unsigned long long* total64) {
  __shared__ unsigned long long s_total;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  // grid-stride: few blocks, so the 64-bit total (which lets the host refuse a K that does not fit the int32
  // prefix sums) costs a few hundred atomics, not one per 256 Gaussians
  unsigned long long wide = 0;
  for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (i64)gridDim.x * blockDim.x) {
    Box b;
    int c = 0;
    if (load_box(start, end, g, W, H, b))
      c = ((b.x1 >> kTileSX) - (b.x0 >> kTileSX) + 1) * ((b.y1 >> kTileSY) - (b.y0 >> kTileSY) + 1);
    cnt[g] = c;
    wide += (unsigned long long)c;
  }
  if (wide) atomicAdd(&s_total, wide);  // integer adds: order-independent, deterministic
  __syncthreads();
  if (threadIdx.x == 0 && s_total) atomicAdd(total64, s_total);
}

// `capacity` / `info` (capture-safe binning, gcp_bin_tiles): a Gaussian whose entries do not fit below `capacity` is
// left out together with everything behind it; info[0] = entries actually listed, info[1] = 1 if anything was left out.
// A box over many tiles (a background splat: 8 100 of them at 1080p) is emitted by its whole wave, lane l taking entries l,
// l + 64, ... — one thread writing them all kept the launch waiting (0.16 -> 0.49 ms for twenty such boxes).
constexpr int kEmitWide = 128;
__global__ __launch_bounds__(256) void k_tile_emit(const int* start, const int* end, i64 n, int W, int H, int tiles_x,
                                                   const int* off, unsigned* key, unsigned* val, i64 capacity, int* info,
                                                   const unsigned long long* total64) {
  const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool live = g < n;
  if (info) {
    if (*total64 > 0x7fffffffull) {  // the int32 prefix sums have wrapped: nothing can be listed
      if (g == 0) { info[0] = 0; info[1] = 1; }
      return;
    }
    if (live) {
      const i64 lo = off[g], hi = off[g + 1];
      if (g == 0 && (i64)off[n] <= capacity) { info[0] = off[n]; info[1] = 0; }
      if (lo <= capacity && hi > capacity) { info[0] = (int)lo; info[1] = 1; }  // the one Gaussian that straddles the bound
      if (hi > capacity) live = false;
    }
  }
  Box b = {0, 0, -1, -1};
  if (live) live = load_box(start, end, g, W, H, b);
  const int tx0 = b.x0 >> kTileSX, ty0 = b.y0 >> kTileSY;
  const int ntx = live ? (b.x1 >> kTileSX) - tx0 + 1 : 0, nty = live ? (b.y1 >> kTileSY) - ty0 + 1 : 0;
  const int e0 = live ? off[g] : 0;
  const bool wide = ntx * nty >= kEmitWide;
  if (live && !wide) {
    int e = e0;
    for (int ty = ty0; ty < ty0 + nty; ++ty)
      for (int tx = tx0; tx < tx0 + ntx; ++tx) {
        key[e] = (unsigned)(ty * tiles_x + tx);
        val[e] = (unsigned)g;
        ++e;
      }
  }
  for (unsigned long long todo = __ballot(wide); todo; todo &= todo - 1ull) {  // wave-uniform: one wide box at a time
    const int owner = __builtin_ctzll(todo);
    const int wtx0 = __builtin_amdgcn_readlane(tx0, owner), wty0 = __builtin_amdgcn_readlane(ty0, owner);
    const int wntx = __builtin_amdgcn_readlane(ntx, owner), wcount = wntx * __builtin_amdgcn_readlane(nty, owner);
    const int we0 = __builtin_amdgcn_readlane(e0, owner);
    const unsigned wg = (unsigned)(g - lane + owner);
    for (int i = lane; i < wcount; i += 64) {
      const int ty = i / wntx, tx = i - ty * wntx;
      key[we0 + i] = (unsigned)((wty0 + ty) * tiles_x + wtx0 + tx);
      val[we0 + i] = wg;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Stable LSD radix sort of (key, payload) pairs, 8-bit digits — used for the (tile, Gaussian) entries of the
// binning and, as gcp_sort_pairs_u32, for M-sized pixel-key arrays (the sort of the reference's
// _create_alpha_brend, gs_model.py:546-547).  Per pass: per-block digit histogram laid out [digit][block], ONE
// linear exclusive scan of it (= first output slot of every (digit, block)), then a stable scatter of 4096-key
// blocks.  Inside a block wave w owns keys [1024w, 1024w+1024): (A) every wave counts its digits, (B) a prefix
// over digits and waves gives every (wave, digit) its first slot, (C) each wave ranks its keys 64 at a time in
// order — peers with the same digit from 8 ballots, rank = earlier lanes among the peers — and parks them in
// LDS in digit order, (D) the block copies LDS out, so every digit run is one coalesced global write.
// Deterministic: no atomic ever decides a slot.
// ------------------------------------------------------------------------------------------
constexpr int kSortChunk = 4096;    // keys per radix-sort block

// RECTS: `key` points at the reference's rect list, int32 (x, y) per pair, and the key is y * 10000 + x
// (gs_model.py:538-541) computed on the fly — the key array of `unique()` never exists in memory.
template <bool RECTS>
__device__ __forceinline__ unsigned load_key(const unsigned* key, i64 i) {
  if (!RECTS) return key[i];
  const int2 r = reinterpret_cast<const int2*>(key)[i];
  return (unsigned)(r.y * 10000 + r.x);
}

// Chunk (4096 keys) of this block.  Blocks are dealt round-robin over the 8 XCDs; with the remap XCD x takes the x-th
// CONTIGUOUS eighth of the chunks, so the blocks that run on one XCD at the same time hold neighbouring chunks: their
// runs inside every digit bucket are adjacent in the destination, and the partial cache lines at the run ends merge in
// that XCD's L2 instead of being written back half-filled from two.  -1: no such chunk (the grid is rounded up to 8).
__device__ __forceinline__ i64 sort_chunk(i64 b, i64 nblk, int xcd_remap) {
  if (!xcd_remap) return b < nblk ? b : -1;
  const i64 per = (nblk + 7) >> 3;
  const i64 c = (b & 7) * per + (b >> 3);
  return ((b >> 3) < per && c < nblk) ? c : -1;
}
inline unsigned sort_grid(i64 nblk, int xcd_remap) { return (unsigned)(xcd_remap ? ((nblk + 7) >> 3) * 8 : nblk); }

// n_dev (optional): the number of keys lives on the device (capture-safe binning); n is then only the bound the grid was
// sized for, and blocks past the real count contribute empty histograms / copy nothing
template <bool RECTS>
__global__ __launch_bounds__(256) void k_sort_hist(const unsigned* key, i64 n, int shift, int* hist, int nblk, const int* n_dev,
                                                   int xcd_remap) {
  __shared__ int h[256];
  const i64 chunk = sort_chunk(blockIdx.x, nblk, xcd_remap);
  if (chunk < 0) return;
  if (n_dev) n = min(n, (i64)*n_dev);
  h[threadIdx.x] = 0;
  __syncthreads();
  const i64 base = chunk * kSortChunk;
#pragma unroll 4
  for (int i = threadIdx.x; i < kSortChunk; i += 256)
    if (base + i < n) atomicAdd(&h[(load_key<RECTS>(key, base + i) >> shift) & 255u], 1);
  __syncthreads();
  hist[(i64)threadIdx.x * nblk + chunk] = h[threadIdx.x];
}

template <bool FIRST, bool RECTS = false>  // FIRST: the payload is the element's own index
__global__ __launch_bounds__(256) void k_sort_scatter(const unsigned* key, const unsigned* val, unsigned* key_out,
                                                       unsigned* val_out, i64 n, int shift, const int* hist_excl,
                                                       int nblk, const int* n_dev, int xcd_remap) {
  const i64 chunk = sort_chunk(blockIdx.x, nblk, xcd_remap);
  if (chunk < 0) return;
  if (n_dev) n = min(n, (i64)*n_dev);
  __shared__ unsigned s_key[kSortChunk];
  __shared__ unsigned s_val[kSortChunk];
  __shared__ int off[4][256];   // (A) per-wave digit counts -> (B) first LDS slot of (wave, digit)
  __shared__ int gdelta[256];   // global slot = LDS slot + gdelta[digit]
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int d = lane; d < 256; d += 64) off[w][d] = 0;
  __syncthreads();
  const i64 bbase = chunk * kSortChunk;
  const i64 wbase = bbase + (i64)w * (kSortChunk / 4);
  constexpr int kSteps = kSortChunk / 4 / 64;  // 16
  unsigned k[kSteps], v[kSteps];
#pragma unroll
  for (int st = 0; st < kSteps; ++st) {  // (A)
    const i64 i = wbase + st * 64 + lane;
    const bool valid = i < n;
    k[st] = valid ? load_key<RECTS>(key, i) : 0u;
    v[st] = FIRST ? (unsigned)i : (valid ? val[i] : 0u);
    if (valid) atomicAdd(&off[w][(k[st] >> shift) & 255u], 1);  // counts only: order-independent
  }
  __syncthreads();
  {  // (B) thread d: digit total -> block-wide exclusive prefix over digits -> per-wave LDS bases
    const int c0 = off[0][tid], c1 = off[1][tid], c2 = off[2][tid], c3 = off[3][tid];
    int total;
    const int lstart = block_excl_scan_256(c0 + c1 + c2 + c3, s_w, total);
    off[0][tid] = lstart;
    off[1][tid] = lstart + c0;
    off[2][tid] = lstart + c0 + c1;
    off[3][tid] = lstart + c0 + c1 + c2;
    gdelta[tid] = hist_excl[(i64)tid * nblk + chunk] - lstart;
  }
  __syncthreads();
#pragma unroll
  for (int st = 0; st < kSteps; ++st) {  // (C) stable ranks, staged into LDS in digit order
    const i64 i = wbase + st * 64 + lane;
    const bool valid = i < n;
    const unsigned d = (k[st] >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int rank = __builtin_popcountll(peers & ((1ull << lane) - 1ull));
    const int pos = off[w][d];
    if (valid && rank == 0) off[w][d] = pos + __builtin_popcountll(peers);
    if (valid) {
      s_key[pos + rank] = k[st];
      s_val[pos + rank] = v[st];
    }
  }
  __syncthreads();
  const int nvalid = (int)((n - bbase < kSortChunk) ? (n - bbase) : kSortChunk);
  for (int i = tid; i < nvalid; i += 256) {  // (D) coalesced copy-out: consecutive lanes, consecutive slots
    const unsigned kk = s_key[i];
    const i64 g = (i64)i + gdelta[(kk >> shift) & 255u];
    key_out[g] = kk;
    val_out[g] = s_val[i];
  }
}

// ------------------------------------------------------------------------------------------
// The same pass for M-sized key arrays (gcp_sort_pairs_u32 / gcp_sort_rects), organised for throughput.  The two kernels
// above spend their time waiting: a block lives for one 4096-key chunk, four of them fit a CU, and each goes load ->
// barrier -> count -> barrier -> rank -> barrier -> copy-out with nothing in flight behind it (measured: 0.9 ms per
// pass at 1.65e8 keys of which 0.1 ms is the scattered stores; the histogram pass another 0.33 ms).  Here one block
// owns a SUPER-CHUNK of up to kSortSub consecutive chunks:
//   * histogram: one column per super-chunk (8x fewer scattered counter writes, a 5 MB table that one tiny scan
//     handles), 16-byte loads, and counts aggregated over runs of equal digits in neighbouring lanes before they reach
//     the LDS atomic (pixel keys arrive partially ordered: without this the upper digits serialise 64-way);
//   * scatter: the digit offsets of the block's first chunk come from the scanned table, the following chunks
//     continue from them (+= the chunk's own counts); 8 waves per block (8 instead of 16 dependent steps per wave and
//     chunk, 24 instead of 16 waves per CU), every key ranked ONCE (the counting phase keeps the ranks), and the
//     staging writes made independent of each other.
// Same result as the chunk-at-a-time kernels bit for bit: integer arithmetic only, and the one fetch-add whose RETURN value
// is used (a digit's running count inside a wave) has a single lane per address and instruction, in program order.
// ------------------------------------------------------------------------------------------
#ifndef GCP_SORT_BIG
#define GCP_SORT_BIG 4096  // (8192 with 16 waves and one block per CU: passes 0 / 1 no faster, measured)
#endif
constexpr int kBigChunk = GCP_SORT_BIG;  // keys staged through LDS at a time by the M-sized sort
constexpr int kSortSub = 8;  // chunks per super-chunk at most

// digit counts of one wave's 64 keys into `h`: lanes with the same digit as their left neighbour are counted by the
// leftmost lane of the run (one LDS atomic per run).  d >= 256 marks a lane without a key.
__device__ __forceinline__ void count_runs(int* h, unsigned d, int lane) {
  const unsigned dl = (unsigned)dpp_i<0x138, 0xf>((int)0xffffffffu, (int)d);  // wave_shr:1, lane 0 sees "no key"
  const bool head = (lane == 0) || (d != dl);
  const unsigned long long heads = __ballot(head);
  if (head && d < 256u) {
    const unsigned long long above = (lane == 63) ? 0ull : (heads >> (lane + 1));
    const int len = above ? (__builtin_ctzll(above) + 1) : (64 - lane);
    atomicAdd(&h[d], len);
  }
}

// Digit of a pass: (key >> shift) & dmask.  With idw > 0 the M-sized sort runs on COMPACT pixel ids y * idw + x (idw =
// image width + 1) instead of the reference's y * 10000 + x — the same order, 21 instead of 24 significant bits at
// 1920x1080, so three passes of 7 bits: half as many digit runs per chunk, twice as long, and the scattered stores of a
// pass come that much closer to whole cache lines.  The last pass turns the ids back into the reference's keys.
struct SortPass { int shift; unsigned dmask; int idw; int restore; float inv_idw; };

template <bool RECTS>
__device__ __forceinline__ unsigned load_sort_key(const unsigned* key, i64 i, int idw) {
  if (!RECTS) return key[i];
  const int2 r = reinterpret_cast<const int2*>(key)[i];
  return (unsigned)(r.y * (idw ? idw : 10000) + r.x);
}

// id -> y * 10000 + x with y = id / idw, x = id % idw (ids < 2^24: exact in fp32; one correction step either way)
__device__ __forceinline__ unsigned restore_key(unsigned id, int idw, float inv_idw) {
  unsigned y = (unsigned)((float)id * inv_idw);
  int x = (int)id - (int)y * idw;
  if (x < 0) { --y; x += idw; }
  else if (x >= idw) { ++y; x -= idw; }
  return y * 10000u + (unsigned)x;
}

template <bool RECTS>
__global__ __launch_bounds__(256) void k_sort_hist2(const unsigned* key, i64 n, const SortPass ps, int* hist, int nsuper, int sub, int xcd_remap,
                                                    int vec /*the source is 16-byte aligned*/) {
  const int shift = ps.shift;
  const unsigned dmask = ps.dmask;
  const int kmul = ps.idw ? ps.idw : 10000;
  __shared__ int h[256];
  const i64 super = sort_chunk(blockIdx.x, nsuper, xcd_remap);
  if (super < 0) return;
  const int lane = threadIdx.x & 63;
  h[threadIdx.x] = 0;
  __syncthreads();
  const i64 base = super * (i64)sub * kBigChunk;
  const i64 end = min(n, base + (i64)sub * kBigChunk);
  // 4 consecutive keys per lane per step; the order inside the super-chunk does not matter for a histogram
  for (i64 p0 = base; p0 < end; p0 += 1024) {  // block-uniform trip count: count_runs needs all 64 lanes of every wave
    const i64 p = p0 + (i64)threadIdx.x * 4;
    unsigned k4[4];
    if (vec && p + 3 < end) {
      if (RECTS) {
        const int4 a = *reinterpret_cast<const int4*>(reinterpret_cast<const int2*>(key) + p);
        const int4 b = *reinterpret_cast<const int4*>(reinterpret_cast<const int2*>(key) + p + 2);
        k4[0] = (unsigned)(a.y * kmul + a.x); k4[1] = (unsigned)(a.w * kmul + a.z);
        k4[2] = (unsigned)(b.y * kmul + b.x); k4[3] = (unsigned)(b.w * kmul + b.z);
      } else {
        const uint4 a = *reinterpret_cast<const uint4*>(key + p);
        k4[0] = a.x; k4[1] = a.y; k4[2] = a.z; k4[3] = a.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) k4[j] = (k4[j] >> shift) & dmask;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) k4[j] = (p + j < end) ? ((load_sort_key<RECTS>(key, p + j, ps.idw) >> shift) & dmask) : 256u;
    }
    // element j of every lane: neighbouring lanes hold keys 4 apart — still mostly equal upper digits
#pragma unroll
    for (int j = 0; j < 4; ++j) count_runs(h, k4[j], lane);
  }
  __syncthreads();
  hist[(i64)threadIdx.x * nsuper + super] = h[threadIdx.x];
}

// Peers of every lane = the lanes of the wave that hold the same 8-bit digit, as a 64-bit mask in two dwords.  Per digit
// bit: one ballot, and the lanes that DIFFER in that bit are OR-ed into a mismatch mask (ballot ^ own bit replicated) —
// 5 VALU per bit instead of the 8 of a select-and-AND formulation; this loop is what the scatter kernel's time is made
// of (it issues 2/3 of its vector instructions).
__device__ __forceinline__ void digit_peers(unsigned d, unsigned long long valid, unsigned& lo, unsigned& hi) {
  unsigned acc_lo = 0u, acc_hi = 0u;
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int sb = ((int)(d << (31 - b))) >> 31;  // 0 / -1: this lane's bit b, replicated
    const unsigned long long m = __ballot(sb != 0);
    acc_lo |= (unsigned)m ^ (unsigned)sb;
    acc_hi |= (unsigned)(m >> 32) ^ (unsigned)sb;
  }
  lo = (unsigned)valid & ~acc_lo;
  hi = (unsigned)(valid >> 32) & ~acc_hi;
}

constexpr int kSortWaves = kBigChunk / 512;  // 8 steps of 64 keys per wave and chunk

template <bool FIRST, bool RECTS>
__global__ __launch_bounds__(64 * kSortWaves) void k_sort_scatter2(const unsigned* key, const unsigned* val, unsigned* key_out,
                                                                    unsigned* val_out, i64 n, const SortPass ps, const int* hist_excl,
                                                                    int nsuper, int sub, int xcd_remap) {
  const i64 super = sort_chunk(blockIdx.x, nsuper, xcd_remap);
  if (super < 0) return;
  const int shift = ps.shift;
  const unsigned dmask = ps.dmask;
  __shared__ unsigned s_key[kBigChunk];
  __shared__ unsigned s_val[kBigChunk];
  __shared__ int off[kSortWaves][256];  // (A) per-wave digit counts -> (B) first LDS slot of (wave, digit)
  __shared__ int gdelta[256];           // global slot = LDS slot + gdelta[digit]
  __shared__ int gbase[256];            // first global slot of the CURRENT chunk's keys of every digit
  __shared__ int s_w[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int kThreadsS = 64 * kSortWaves;
  constexpr int kWaveKeys = kBigChunk / kSortWaves;  // 512
  constexpr int kSteps = kWaveKeys / 64;              // 8
  const i64 sbase = super * (i64)sub * kBigChunk;
  const int nsub = (int)min((i64)sub, (n - sbase + kBigChunk - 1) / kBigChunk);
  if (tid < 256) gbase[tid] = hist_excl[(i64)tid * nsuper + super];
  for (int c = 0; c < nsub; ++c) {
    const i64 bbase = sbase + (i64)c * kBigChunk;
    const i64 wbase = bbase + (i64)w * kWaveKeys;
    unsigned k[kSteps], v[kSteps];
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const i64 i = wbase + st * 64 + lane;
      const bool valid = i < n;
      k[st] = valid ? load_sort_key<RECTS>(key, i, ps.idw) : 0u;  // (non-temporal loads here: no difference, measured)
      v[st] = FIRST ? (unsigned)i : (valid ? val[i] : 0u);
    }
    for (int d = lane; d < 256; d += 64) off[w][d] = 0;
    __syncthreads();  // (also: the previous chunk's copy-out has read s_key / s_val / gdelta)
    // (A) ranks.  Per step of 64 keys: the peers of every lane; its rank among them (lanes below it); the leader (lowest
    // peer) adds the peer count to the wave's counter of that digit and gets back how many keys of the digit the wave's
    // EARLIER steps held — one lane per address and instruction, the steps of a wave in program order, so the returned
    // value does not depend on any arbitration — and hands it to its peers.  slot[st] = position among the wave's keys of
    // this digit: nothing is left to do in (C) but to add the digit's base.
    int slot[kSteps];
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {
      const bool valid = wbase + st * 64 + lane < n;
      const unsigned d = (k[st] >> shift) & dmask;
      unsigned plo, phi;
      digit_peers(d, __ballot(valid), plo, phi);
      const int rank = (int)__builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
      const int cnt = __builtin_popcount(plo) + __builtin_popcount(phi);
      int before = 0;
      if (valid && rank == 0) before = atomicAdd(&off[w][d], cnt);
      const int leader = plo ? __builtin_ctz(plo) : (32 + __builtin_ctz(phi | 0x80000000u));
      before = __builtin_amdgcn_ds_bpermute(leader << 2, before);
      slot[st] = before + rank;
    }
    __syncthreads();
    {  // (B) thread d < 256: digit total -> block-wide exclusive prefix over digits -> per-wave LDS bases; global bases move on
      int cw[kSortWaves], sum = 0;
      if (tid < 256) {
#pragma unroll
        for (int j = 0; j < kSortWaves; ++j) { cw[j] = off[j][tid]; sum += cw[j]; }
      }
      const int inc = wave_incl_scan_i(sum);
      if (lane == 63 && w < 4) s_w[w] = inc;
      __syncthreads();
      if (tid < 256) {
        int run = inc - sum;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < w) run += s_w[j];
        const int gb = gbase[tid];
        gdelta[tid] = gb - run;
        gbase[tid] = gb + sum;
#pragma unroll
        for (int j = 0; j < kSortWaves; ++j) { off[j][tid] = run; run += cw[j]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < kSteps; ++st) {  // (C) staged into LDS in digit order: independent reads and writes
      if (wbase + st * 64 + lane < n) {
        const int pos = off[w][(k[st] >> shift) & dmask] + slot[st];
        s_key[pos] = k[st];
        s_val[pos] = v[st];
      }
    }
    __syncthreads();
    const int nvalid = (int)((n - bbase < kBigChunk) ? (n - bbase) : kBigChunk);
    for (int i = tid; i < nvalid; i += kThreadsS) {  // (D) coalesced copy-out: consecutive lanes, consecutive slots
      const unsigned kk = s_key[i];
      const i64 g = (i64)i + gdelta[(kk >> shift) & dmask];
      if ((unsigned long long)g < (unsigned long long)n) {  // always true; keeps a broken histogram from becoming a wild store
        key_out[g] = ps.restore ? restore_key(kk, ps.idw, ps.inv_idw) : kk;
        val_out[g] = s_val[i];
      }
    }
  }
}

// tile_start[t] = first sorted entry whose tile id is >= t, for t in [0, n_tiles]
__global__ void k_tile_bounds(const unsigned* key, i64 K, int n_tiles, int* tile_start, const int* n_dev) {
  if (n_dev) K = min(K, (i64)*n_dev);
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > K) return;
  const int prev = (i == 0) ? -1 : (int)key[i - 1];
  const int cur = (i == K) ? n_tiles : (int)key[i];
  for (int t = prev + 1; t <= cur; ++t) tile_start[t] = (int)i;
}

// ------------------------------------------------------------------------------------------
// f1: fused blend
// ------------------------------------------------------------------------------------------
struct BlendArgs {
  const int* start;      // [N,2] x,y inclusive
  const int* end;        // [N,2]
  const float* mean;     // [N,2] x,y
  const float* vinv;     // [N,2,2]
  const float* opacity;  // [N]
  const float* l_d;      // [N,3]
  const int* tile_start; // [n_tiles+1]
  const unsigned* tile_list;  // [K] gaussian id, depth order inside each tile
  int W, H, tiles_x;
};

template <int STAGE>
struct Staged {
  int4 box[STAGE];     // x0, y0, x1-x0, y1-y0 (clamped to the image)
  float4 geo[STAGE];   // mx, my, opacity, box mask as bits (0-15: tile columns inside the box, 16-31: tile rows)
  float4 vin[STAGE];   // Λ' = -0.5*log2(e) * Λ, Λ = [[a,b],[c,d]]: a b c d (forward) or a, b + c, d, - (backward)
  float4 col[STAGE];   // l0 l1 l2, 1/opacity (0 if opacity == 0)
  // hits[w][c]: bit j set = staged entry 64*c + j reaches into the four pixel rows of wave w.  A wave walks the set
  // bits of its own words (scalar s_ff1 / s_andn2) and never sees the entries that miss it.
  unsigned long long hits[4][(STAGE + 63) / 64];
};

template <int STAGE, bool QUAD3>
__device__ __forceinline__ void stage_entries(const BlendArgs& a, Staged<STAGE>& s, int first, int cnt, int tile_x0, int tile_y0) {
  for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
    const i64 g = a.tile_list[first + j];
    Box b;
    load_box(a.start, a.end, g, a.W, a.H, b);
    s.box[j] = make_int4(b.x0, b.y0, b.x1 - b.x0, b.y1 - b.y0);
    // the box as two 16-bit masks over the tile's columns and rows: membership of a pixel is ONE and + ONE compare
    // against the lane's own two bits
    const int c0 = max(b.x0 - tile_x0, 0), c1 = min(b.x1 - tile_x0, kTile - 1);
    const int r0 = max(b.y0 - tile_y0, 0), r1 = min(b.y1 - tile_y0, kTile - 1);
    const unsigned cm = (c1 >= c0) ? ((2u << c1) - (1u << c0)) : 0u;
    const unsigned rm = (r1 >= r0) ? ((2u << r1) - (1u << r0)) : 0u;
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {  // entries j of one 64-lane staging wave form one word per target wave
      const unsigned long long touched = __ballot(((rm >> (4 * w2)) & 0xfu) != 0u);
      if ((threadIdx.x & 63) == 0) s.hits[w2][j >> 6] = touched;
    }
    const float op = a.opacity[g];
    s.geo[j] = make_float4(a.mean[2 * g], a.mean[2 * g + 1], op, __uint_as_float(cm | (rm << 16)));
    // Λ pre-scaled by -0.5*log2(e): g = exp(-0.5 d Λ d^T) becomes ONE v_exp_f32 of d Λ' d^T.  The extra rounding of
    // Λ' moves g by < 1e-7 absolute (relative 6e-8*|log2 g|, and g decays as fast as that factor grows).
    constexpr float kS = -0.5f * 1.44269504088896341f;
    // QUAD3 (backward): a, b + c, d — the quadratic form in five VALU instead of six; the forward keeps a, b, c, d and the
    // association of the reference's two matmuls: its per-entry chain is latency-bound and the shorter form is 6 % slower there
    s.vin[j] = QUAD3 ? make_float4(kS * a.vinv[4 * g], kS * a.vinv[4 * g + 1] + kS * a.vinv[4 * g + 2], kS * a.vinv[4 * g + 3], 0.0f)
                     : make_float4(kS * a.vinv[4 * g], kS * a.vinv[4 * g + 1], kS * a.vinv[4 * g + 2], kS * a.vinv[4 * g + 3]);
    s.col[j] = make_float4(a.l_d[3 * g], a.l_d[3 * g + 1], a.l_d[3 * g + 2], op != 0.0f ? 1.0f / op : 0.0f);
  }
}

// the set bits of a wave-uniform 64-bit word, lowest first, on the scalar unit
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

#ifndef GCP_DPP_ASM
#define GCP_DPP_ASM 1
#endif
#ifndef GCP_BLEND_FMA
#define GCP_BLEND_FMA 1
#endif
#if GCP_BLEND_FMA
// The blend kernels are VALU-issue bound: let a*b+c contract into v_fma_f32 here (the library is otherwise built
// with -ffp-contract=off).  One rounding instead of two per contraction; results stay within the 1e-5 bar.
#define GCP_FP_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define GCP_FP_CONTRACT
#endif

// Transmittance checkpoints: slot q of tile t holds every pixel's transmittance in front of list entry first + q kCkpt,
// for q = 0 .. ceil(n / kCkpt) — the last one is the transmittance behind the whole list — 256 floats each (one per
// pixel of the tile, thread order).  Tile t's slots start at first / kCkpt + 2 t: consecutive tiles never overlap
// (floor(first/c) + ceil(n/c) + 1 <= floor((first+n)/c) + 2), so K / kCkpt + 2 n_tiles + 2 slots hold them all without a
// separate prefix sum.
__device__ __forceinline__ i64 ckpt_slot0(int first, int tile) { return (i64)(first / kCkpt) + 2 * (i64)tile; }
inline size_t ckpt_floats(i64 n_tile_pairs, int n_tiles) {
  return (size_t)((n_tile_pairs > 0 ? n_tile_pairs : 0) / kCkpt + 2 * (i64)n_tiles + 2) * 256u;
}

template <bool CKPT>
__global__ __launch_bounds__(256) void k_blend_fwd(const BlendArgs a, float* __restrict__ image, float* __restrict__ t_ckpt) {
  GCP_FP_CONTRACT
  __shared__ Staged<kStage> s;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave id in an SGPR: the row test below is scalar
  const int tile = blockIdx.x;
  const int px = (tile % a.tiles_x) * kTile + (lane & 15);
  const int py = (tile / a.tiles_x) * kTile + w * 4 + (lane >> 4);
  const float fx = (float)px, fy = (float)py;
  const unsigned lane_bits = (1u << (lane & 15)) | (1u << (16 + w * 4 + (lane >> 4)));
  const int first = a.tile_start[tile], last = a.tile_start[tile + 1];
  float* const ck = CKPT ? t_ckpt + ckpt_slot0(first, tile) * 256 + threadIdx.x : nullptr;
  float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  for (int base = first; base < last; base += kStage) {
    const int cnt = __builtin_amdgcn_readfirstlane(min(kStage, last - base));  // scalar loop bound
    __syncthreads();
    stage_entries<kStage, false>(a, s, base, cnt, (tile % a.tiles_x) * kTile, (tile / a.tiles_x) * kTile);
    __syncthreads();
    // Only the entries whose rows reach this wave (hits[w]), in list order.  Every listed entry is used, so all
    // three of its LDS records are read together, ahead of the membership branch.  (Reading one entry ahead of the
    // blend, with two register sets, changes nothing: 93 % of the launch is VALU issue.)
    auto blend = [&](const float4& ge, const float4& vi, const float4& co) {
      if ((__float_as_uint(ge.w) & lane_bits) == lane_bits) {  // (the branch-free form of the backward is 4 % slower here)
        const float dx = fx - ge.x, dy = fy - ge.y;
        // (d Λ) d^T with the association of the reference's two matmuls (gs_model.py:495)
        const float t0 = dx * vi.x + dy * vi.z;
        const float t1 = dx * vi.y + dy * vi.w;
        const float g = __builtin_amdgcn_exp2f(t0 * dx + t1 * dy);  // = exp(-0.5 (d Λ) d^T), gs_model.py:495
        const float anti = 1.0f - ge.z * g;             // gs_model.py:535
        const float incl = T * anti;                    // inclusive grouped cumprod
        if (incl != 0.0f) {                             // gs_model.py:560: dropped when exactly 0
          const float wgt = T * ge.z * g;               // gs_model.py:500
          c0 += wgt * co.x; c1 += wgt * co.y; c2 += wgt * co.z;
        }
        T = incl;
      }
    };
    const int chunks = (cnt + 63) >> 6;
    for (int c = 0; c < chunks; ++c) {
      const unsigned long long hits64 = uniform64(s.hits[w][c]);
      if (!CKPT && !hits64) continue;
#pragma unroll
      for (int sub = 0; sub < 64 / kCkpt; ++sub) {
        const int e0 = c * 64 + sub * kCkpt;  // first staged entry of this checkpoint interval
        if (CKPT) {
          if (e0 >= cnt) break;  // wave-uniform
          // the transmittance entering list entry base + e0 (the backward restarts its front-to-back pass from here)
          ck[(i64)((base - first + e0) / kCkpt) * 256] = T;
        }
        unsigned hits = (unsigned)(hits64 >> (sub * kCkpt)) & (unsigned)((1ull << kCkpt) - 1ull);
        // wave-uniform: every pixel of the strip is behind an exact zero (an opaque layer, or a product that underflowed
        // hundreds of layers deep) — T stays 0 and nothing more reaches the image: the rest of the list costs this wave its
        // checkpoints only
        if (__ballot(T != 0.0f) == 0ull) hits = 0u;
        while (hits) {
          const int k = e0 + __builtin_ctz(hits);
          hits &= hits - 1;
          const float4 ge = s.geo[k], vi = s.vin[k], co = s.col[k];
          asm volatile("" :: "v"(vi.x), "v"(co.x));  // keep the reads ahead of the branch
          blend(ge, vi, co);
        }
      }
    }
  }
  if (CKPT) ck[(i64)((last - first + kCkpt - 1) / kCkpt) * 256] = T;  // behind the whole list
  if (px <= a.W && py <= a.H) {
    float* o = image + ((i64)py * (a.W + 1) + px) * 3;
    o[0] = c0; o[1] = c1; o[2] = c2;
  }
}

// Backward: per (tile, entry) partial sums, written to the entry's Gaussian-major slot.
//
// Per pixel, with k running over the list entries whose box holds the pixel, T_k the exclusive transmittance,
// a_k = o_k g_k, c_k = (dL/dI . l_k), p_k = T_k a_k l_k (gs_model.py:500):
//   S_k = sum_{j>k} (dL/dI . p_j)                    exclusive suffix sum, gs_model.py:716-722
//   S_k / (1 - a_k) = T_k R_k,   R_{k-1} = a_k c_k + (1 - a_k) R_k = R_k + a_k (c_k - R_k),   R_last = 0
//   dL/do_k    = T_k g_k (c_k - R_k)                 gs_model.py:733-740   (= gp/o - (g/anti) S)
//   "common"_k = T_k a_k (c_k - R_k)                 gs_model.py:747-748, :757-758   (= gp - (a/anti) S)
//   dL/dl_k    = dL/dI T_k a_k                       (true gradient; the reference's is channel-collapsed, Q2)
// The list is walked back to front, one staged chunk of kStageBwd entries at a time, in ONE pass: the transmittance in
// front of an entry comes from the one behind it, T_k = T_{k+1} / (1 - a_k) (v_rcp_f32, 1 ulp), restarted at every chunk
// from the checkpoint the forward kernel saved for the chunk's END — so the quotients never chain further than one chunk
// (<= 32 roundings, ~2e-6 relative) and nothing is subtracted or accumulated in T.  A quotient cannot undo an
// underflow: a chunk in which some pixel's transmittance falls below FLT_MIN (at most one chunk per pixel) is handled by
// its wave with T_k recomputed front to back from the chunk's START checkpoint for every entry instead (exact; the product
// up to an entry's group of eight is formed once per group: ~160 entry evaluations per chunk, two more registers).  g_k = 0 where the pixel is outside the box or the pair was dropped
// (gs_model.py:560) — such an entry then contributes exactly nothing.  Every gradient term is T_k times a convex
// combination of the c_j: its round-off is relative to the layer's own transmittance at any depth.
// T behind staged entry j for this lane's pixel, given T in front of it: T * (1 - o_j g_j) inside the entry's box, T outside
template <int STAGE>
__device__ __forceinline__ float slow_factor(const Staged<STAGE>& s, int j, float fx, float fy, unsigned lane_bits, float T) {
  const float4 gj = s.geo[j];
  const float4 vj = s.vin[j];
  const float dxj = fx - gj.x, dyj = fy - gj.y;
  const float g_j = __builtin_amdgcn_exp2f(dxj * (vj.x * dxj + vj.y * dyj) + (vj.z * dyj) * dyj);
  return ((__float_as_uint(gj.w) & lane_bits) == lane_bits) ? T * (1.0f - gj.z * g_j) : T;
}

__global__ __launch_bounds__(256) void k_blend_bwd(const BlendArgs a, const int* __restrict__ tile_off,
                                                   const float* __restrict__ t_ckpt,
                                                   const float* __restrict__ grad_image,
                                                   float* __restrict__ partial /*[K][kGradVals]*/) {
  GCP_FP_CONTRACT
  __shared__ Staged<kStageBwd> s;
  // [entry][pixel row of the tile * kRowSlots + value]; one word of padding per entry: the fold below reads with one
  // thread per entry, and a stride of 128 words would put all of them on one LDS bank
#ifndef GCP_PART_PAD
#define GCP_PART_PAD 1
#endif
  constexpr int kPartStride = 16 * kRowSlots + GCP_PART_PAD;
  __shared__ float s_part[kStageBwd][kPartStride];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave id in an SGPR: the row test below is scalar
  const int tile = blockIdx.x;
  const int ttx = tile % a.tiles_x, tty = tile / a.tiles_x;
  const int px = ttx * kTile + (lane & 15);
  const int py = tty * kTile + w * 4 + (lane >> 4);
  const float fx = (float)px, fy = (float)py;
  const int first = a.tile_start[tile], last = a.tile_start[tile + 1];
  float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;
  if (px <= a.W && py <= a.H) {
    const i64 o = ((i64)py * (a.W + 1) + px) * 3;
    g0 = grad_image[o]; g1 = grad_image[o + 1]; g2 = grad_image[o + 2];
  }
  const unsigned lane_bits = (1u << (lane & 15)) | (1u << (16 + w * 4 + (lane >> 4)));
  const bool b1 = lane & 2;
#if !GCP_DPP_ASM
  const bool b3 = lane & 8, b2 = lane & 4;
#endif
  float* const row_slot = &s_part[0][(w * 4 + (lane >> 4)) * kRowSlots + ((lane & 2) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 8) ? 1 : 0)];
  const float* const ck = t_ckpt + ckpt_slot0(first, tile) * 256 + threadIdx.x;
  static_assert(kStageBwd <= 32, "one 32-bit word of hits per wave");
  const int nchunks = (last - first + kStageBwd - 1) / kStageBwd;
  float R = 0.0f;  // R_k of the deepest entry handled so far (the suffix behind the end of the list is empty)
  for (int q = nchunks - 1; q >= 0; --q) {
    const int base = first + q * kStageBwd;
    const int cnt = __builtin_amdgcn_readfirstlane(min(kStageBwd, last - base));  // scalar loop bound
    // transmittance behind this chunk (the next chunk's start, or the end of the list) and in front of it; issued ahead
    // of the staging: their latency hides behind the barrier
    const float T_end = ck[(i64)(q + 1) * 256];
    const float T_start = (q > 0) ? ck[(i64)q * 256] : 1.0f;
    __syncthreads();
    stage_entries<kStageBwd, true>(a, s, base, cnt, ttx * kTile, tty * kTile);
    __syncthreads();
    // only the entries whose rows reach this wave, deepest first (the fold below skips this wave's rows for the others,
    // so nothing needs zeroing)
    const unsigned hits = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s.hits[w][0]);
    // wave-uniform: some pixel of the strip underflows inside this chunk — quotients cannot be trusted for it
    const bool slow = __ballot(T_end < 1.17549435e-38f && T_start != 0.0f) != 0ull;
    float Tn = T_end;  // transmittance behind the entry in hand
    float T_group = T_start;
    int slow_group = -1;
    auto walk_chunk = [&](auto slow_tag) {
    constexpr bool kSlow = decltype(slow_tag)::value;
    unsigned h = hits;
    while (h) {
      const int k = 31 - __builtin_clz(h);
      h &= ~(1u << k);
      {
        // straight-line for all 64 lanes; lanes outside the box / dropped pairs are zeroed with selects
        const float4 ge = s.geo[k];
        const float4 vi = s.vin[k];
        const float4 co = s.col[k];
        const bool in = (__float_as_uint(ge.w) & lane_bits) == lane_bits;
        const float dx = fx - ge.x, dy = fy - ge.y;
        // (d Λ) d^T (gs_model.py:495) as a dx^2 + (b + c) dx dy + d dy^2
        const float gv = __builtin_amdgcn_exp2f(dx * (vi.x * dx + vi.y * dy) + (vi.z * dy) * dy);
        const float og = ge.z * gv;
        const float anti = 1.0f - og;                       // gs_model.py:535
        float Tk, incl;
        if (!kSlow) {
          incl = Tn;                                          // what the forward pass left behind this pair
          Tk = (Tn == 0.0f) ? 0.0f : Tn * __builtin_amdgcn_rcpf(anti);
        } else {
          // T_k front to back from the chunk's start checkpoint: the same products in the same order for every entry, but not
          // from scratch for each — the product up to the entry's group of eight slots is formed once per group (the walk
          // visits the groups back to front: 0 + 8 + 16 + 24 slots) and only the group's own earlier slots per entry
          // (<= 7): about 160 entry evaluations per chunk instead of 496.  A scene whose pixels underflow at different
          // depths takes this branch in many chunks of a wave (blend backward 0.81 -> 2.0 ms with the Gaussians crowding the
          // image centre, before this).
          const unsigned below_group = (1u << (k & ~7)) - 1u;
          if ((k & ~7) != slow_group) {  // wave-uniform
            slow_group = k & ~7;
            T_group = T_start;
            for (unsigned hh = hits & below_group; hh; hh &= hh - 1) T_group = slow_factor(s, __builtin_ctz(hh), fx, fy, lane_bits, T_group);
          }
          Tk = T_group;
          for (unsigned hh = hits & ((1u << k) - 1u) & ~below_group; hh; hh &= hh - 1) Tk = slow_factor(s, __builtin_ctz(hh), fx, fy, lane_bits, Tk);
          incl = Tk * anti;
        }
        const bool keep = in & (incl != 0.0f);                // dropped when the inclusive product is exactly 0 (gs_model.py:560)
        Tn = in ? Tk : Tn;
        const float tg = keep ? Tk * gv : 0.0f;
        const float c = g0 * co.x + g1 * co.y + g2 * co.z;   // dL/dI . l
        const float d = c - R;
        float r_o = tg * d;
        const float wgt = tg * ge.z;                            // T o g (gs_model.py:500 without l)
        float r_c = wgt * d;
        float r_l0 = g0 * wgt, r_l1 = g1 * wgt, r_l2 = g2 * wgt;
        float r_cx = r_c * dx;
        float r_xx = r_cx * dx;
        R = keep ? R + og * d : R;                              // R_{k-1} = R_k + a_k (c_k - R_k)
        const int kc = k;
        // Seven 16-lane row sums by a transposed butterfly: at each step a lane keeps half of its values and hands the
        // other half to its partner, so the live registers halve.  Partners: 15-i, 7-i (within each half), i^2, i^1;
        // afterwards lane i holds the row sum of value ((i>>1)&1)*4 + ((i>>2)&1)*2 + ((i>>3)&1) (slot 7 is a dummy).
        // The first two steps split the lanes by bit 3 and bit 2, i.e. by DPP bank: two bank-masked v_add_f32_dpp
        // writing one destination do "keep + partner's copy" for both classes without a select (7 + 4 VALU); the last
        // two need selects (3 + 1).  15 VALU instead of 7 x 4 = 28.
        float q0, q1, q2, q3, p0, p1;
#if GCP_DPP_ASM
        // s_nop 1: a DPP source written by the preceding VALU instruction needs two wait states
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %4, %4 row_mirror row_mask:0xf bank_mask:0x3\n\t"    // lanes 0-7 : go
            "v_add_f32_dpp %1, %6, %6 row_mirror row_mask:0xf bank_mask:0x3\n\t"    //             gl1
            "v_add_f32_dpp %2, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"    //             S(c)
            "v_add_f32_dpp %3, %10, %10 row_mirror row_mask:0xf bank_mask:0xf\n\t"  // all lanes : S(c dx dx)
            "v_add_f32_dpp %0, %5, %5 row_mirror row_mask:0xf bank_mask:0xc\n\t"    // lanes 8-15: gl0
            "v_add_f32_dpp %1, %7, %7 row_mirror row_mask:0xf bank_mask:0xc\n\t"    //             gl2
            "v_add_f32_dpp %2, %9, %9 row_mirror row_mask:0xf bank_mask:0xc\n\t"    //             S(c dx)
            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
            : "v"(r_o), "v"(r_l0), "v"(r_l1), "v"(r_l2), "v"(r_c), "v"(r_cx), "v"(r_xx));
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"  // bit 2 clear: from q0 / q2
            "v_add_f32_dpp %1, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"  // bit 2 set  : from q1 / q3
            "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
            : "=&v"(p0), "=&v"(p1)
            : "v"(q0), "v"(q1), "v"(q2), "v"(q3));
#else
        q0 = xchg_sum<0x140>(b3, r_o, r_l0);     // b3 = 0 lanes: go     | b3 = 1: gl0
        q1 = xchg_sum<0x140>(b3, r_l1, r_l2);    //               gl1    |         gl2
        q2 = xchg_sum<0x140>(b3, r_c, r_cx);     //               S(c)   |         S(c dx)
        q3 = r_xx + dpp_f<0x140, 0xf>(0.0f, r_xx);  // S(c dx dx) in both classes; the b3 = 1 copy lands in slot 7
        p0 = xchg_sum<0x141>(b2, q0, q1);
        p1 = xchg_sum<0x141>(b2, q2, q3);
#endif
        const float o0 = xchg_sum<0x4e>(b1, p0, p1);
        const float tot = o0 + dpp_f<0xb1, 0xf>(0.0f, o0);
        row_slot[kc * kPartStride] = tot;  // lanes i and i^1 store the same word
      }
    }
    };
    // two copies of the loop, chosen per chunk: the common one carries nothing of the other's (with one loop and the test inside,
    // the common path ran 2-6 % slower after the slow one grew its per-group product)
    if (__ballot(T_start != 0.0f) == 0ull) {
      // every pixel of the strip enters the chunk behind an exact zero: T_k = 0 throughout, every term is 0 and R does not
      // move — the wave hands the fold its zeros and goes on (a scene that crowds one region has half its pairs there)
      for (unsigned hz = hits; hz; hz &= hz - 1) row_slot[__builtin_ctz(hz) * kPartStride] = 0.0f;
    } else if (slow) {
      walk_chunk(std::true_type{});
    } else {
      walk_chunk(std::false_type{});
    }
    __syncthreads();
    // one thread per entry: add the 16 pixel rows in fixed order, write the entry's Gaussian-major slot
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const i64 g = a.tile_list[base + j];
      const int4 bx = s.box[j];
      const int ntx = ((bx.x + bx.z) >> 4) - (bx.x >> 4) + 1;
      const i64 e = (i64)tile_off[g] + (i64)(tty - (bx.y >> 4)) * ntx + (ttx - (bx.x >> 4));
      float* out = partial + e * kGradVals;
      const float my = s.geo[j].y;
      float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f, cx = 0.0f, cy = 0.0f, xx = 0.0f, xy = 0.0f, yy = 0.0f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        if (!((s.hits[wv][0] >> j) & 1ull)) continue;  // that wave never wrote its rows for this entry
#pragma unroll
        for (int r = 4 * wv; r < 4 * wv + 4; ++r) {
          const float* d = &s_part[j][r * kRowSlots];
          const float dy = (float)(tty * kTile + r) - my;  // constant along the pixel row
          o0 += d[0]; o1 += d[1]; o2 += d[2]; o3 += d[3];
          cx += d[5]; cy += dy * d[4];
          xx += d[6]; xy += dy * d[5]; yy += dy * dy * d[4];
        }
      }
      out[0] = o0; out[1] = o1; out[2] = o2; out[3] = o3; out[4] = cx; out[5] = cy; out[6] = xx; out[7] = xy; out[8] = yy;
    }
  }
}

// per Gaussian: sum its tile slots in order, expand the moments into the four gradients.
// A Gaussian whose box covers many tiles (a background splat over the whole frame: 8 100 slots of 9 values) is not left to one
// thread — 73 000 dependent loads, 1.1 ms for twenty of them while the rest of the launch takes 0.04 — but summed by its whole
// wave: lane l takes slots e0 + l, e0 + l + 64, ... in order, and the 64 partial sums are added in a fixed butterfly.  Which
// Gaussians go that way depends on their slot count alone, so the result is the same from run to run.
constexpr int kReduceWide = 128;  // tile slots from which a Gaussian is summed by the wave
__global__ __launch_bounds__(256) void k_grad_reduce(const float* __restrict__ partial, const int* __restrict__ tile_off,
                                                     const int* __restrict__ tile_start, int n_tiles, const float* __restrict__ vinv, i64 n,
                                                     i64 capacity, float* grad_mean, float* grad_vinv, float* grad_opacity, float* grad_l) {
  const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float r[kGradVals];
#pragma unroll
  for (int v = 0; v < kGradVals; ++v) r[v] = 0.0f;
  // Only slots the blend kernel wrote are summed: a Gaussian's entries [tile_off[g], tile_off[g+1]) count iff they lie
  // inside what the binning LISTED (tile_start[n_tiles] entries: everything with exact binning; with a capture-safe
  // capacity the Gaussians that fit; nothing at all when the int32 prefix sums wrapped at > 2^31 entries — the offsets
  // behind the wrap are negative or decreasing, and even the ones before it point at slots nobody wrote).  Zeros otherwise.
  const i64 listed = min((i64)tile_start[n_tiles], capacity);
  i64 e0 = 0, e1 = 0;
  if (g < n) {
    e0 = tile_off[g];
    e1 = tile_off[g + 1];
    if (e0 < 0 || e1 < e0 || e1 > listed) e1 = e0 < 0 ? 0 : e0;
    if (e0 < 0) e0 = 0;
  }
  const bool wide = e1 - e0 >= kReduceWide;
  if (!wide) {
    for (i64 e = e0; e < e1; ++e)
#pragma unroll
      for (int v = 0; v < kGradVals; ++v) r[v] += partial[e * kGradVals + v];
  }
  for (unsigned long long todo = __ballot(wide); todo; todo &= todo - 1ull) {  // wave-uniform: one wide Gaussian at a time
    const int owner = __builtin_ctzll(todo);
    const i64 f0 = (i64)__builtin_amdgcn_readlane((int)e0, owner), f1 = (i64)__builtin_amdgcn_readlane((int)e1, owner);  // (entries < 2^31)
    float p[kGradVals];
#pragma unroll
    for (int v = 0; v < kGradVals; ++v) p[v] = 0.0f;
    for (i64 e = f0 + lane; e < f1; e += 64)
#pragma unroll
      for (int v = 0; v < kGradVals; ++v) p[v] += partial[e * kGradVals + v];
#pragma unroll
    for (int v = 0; v < kGradVals; ++v) {
      float t = p[v];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);  // a fixed butterfly: every lane ends with the same sum
      if (lane == owner) r[v] = t;
    }
  }
  if (g >= n) return;
  const float A = vinv[4 * g], B = vinv[4 * g + 1], C = vinv[4 * g + 2], D = vinv[4 * g + 3];
  grad_opacity[g] = r[0];
  grad_l[3 * g] = r[1]; grad_l[3 * g + 1] = r[2]; grad_l[3 * g + 2] = r[3];
  // sum common * (d Λ): x0 = dx a + dy c, x1 = dx b + dy d   (gs_model.py:745)
  grad_mean[2 * g] = r[4] * A + r[5] * C;
  grad_mean[2 * g + 1] = r[4] * B + r[5] * D;
  // -0.5 * sum common * d^T d   (gs_model.py:755-758)
  grad_vinv[4 * g] = -0.5f * r[6];
  grad_vinv[4 * g + 1] = -0.5f * r[7];
  grad_vinv[4 * g + 2] = -0.5f * r[7];
  grad_vinv[4 * g + 3] = -0.5f * r[8];
}

// ------------------------------------------------------------------------------------------
// per-pixel CSR export (what torch.sort(stable) + unique give the reference, gs_model.py:546-548)
// ------------------------------------------------------------------------------------------
template <bool FILL>
__global__ __launch_bounds__(256) void k_pixel_lists(const BlendArgs a, int* __restrict__ pixel_count,
                                                     const int* __restrict__ pixel_off,
                                                     const int* __restrict__ box_off, int* __restrict__ pair_gauss,
                                                     int* __restrict__ pair_index, int* __restrict__ pair_key) {
  __shared__ int4 s_box[kStage];
  __shared__ int s_g[kStage];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  const int px = (tile % a.tiles_x) * kTile + (lane & 15);
  const int py = (tile / a.tiles_x) * kTile + w * 4 + (lane >> 4);
  const bool in_img = px <= a.W && py <= a.H;
  const int first = a.tile_start[tile], last = a.tile_start[tile + 1];
  int n = 0;
  i64 o = 0;
  if (FILL && in_img) o = pixel_off[(i64)py * (a.W + 1) + px];
  for (int base = first; base < last; base += kStage) {
    const int cnt = min(kStage, last - base);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const i64 g = a.tile_list[base + j];
      Box b;
      load_box(a.start, a.end, g, a.W, a.H, b);
      s_box[j] = make_int4(b.x0, b.y0, b.x1, b.y1);
      s_g[j] = (int)g;
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const int4 bx = s_box[k];
      const bool in = (px >= bx.x) & (px <= bx.z) & (py >= bx.y) & (py <= bx.w);
      if (in) {
        if (FILL) {
          const int g = s_g[k];
          pair_gauss[o + n] = g;
          // position of this pixel in the Gaussian-major rect list (uitility.py:336-366):
          // row-major inside the box, boxes concatenated in depth order
          pair_index[o + n] = box_off[g] + (py - bx.y) * (bx.z - bx.x + 1) + (px - bx.x);
          if (pair_key) pair_key[o + n] = py * 10000 + px;  // the reference's pixel key (gs_model.py:538-541)
        }
        ++n;
      }
    }
  }
  if (!FILL && in_img) pixel_count[(i64)py * (a.W + 1) + px] = n;
}

// ------------------------------------------------------------------------------------------
// Rows a5 / a6 for a caller that still holds the BOXES its rect list was expanded from (gs_model.py:601 `_create_rects`
// feeds :607): the sort -> gather -> scan -> un-sort of _create_alpha_brend (gs_model.py:546-555) collapses into one walk
// of the depth-ordered tile lists.  One block per 16x16 tile, one pixel per lane; every lane walks the tile's list and,
// for the entries whose box holds its pixel, reads the pair's value at its Gaussian-major position
//   box_off[g] + (py - y0) * width_g + (px - x0)            (uitility.py:336-366)
// folds it into its running product / sum and writes the INCLUSIVE value back to the same position — exactly
// `output[torch.argsort(index)]` of gs_model.py:555, with every pixel scanned strictly front to back (the CPU path's own
// association; MODE 2: back to front = grad_cumsum's flipped scan, gs_model.py:716-722).  No M-sized sort, no M-sized
// index array: 8 B per pair.  Lanes of one pixel row read and write consecutive addresses (64 B per box row and tile).
// ------------------------------------------------------------------------------------------
constexpr int kWalkStage = 64;  // list entries staged per round: one hit word per wave
#ifndef GCP_WALK_BATCH
#define GCP_WALK_BATCH 8
#endif
#ifndef GCP_WALK_DBG
#define GCP_WALK_DBG 0  // measurement builds only (tools/build_variant.py): 1 = no loads, 2 = no stores — DESIGN.md §3.4's split
#endif
constexpr int kWalkBatch = GCP_WALK_BATCH;  // listed entries whose loads are in flight together

// A batch: kWalkBatch listed entries of one wave's hit word.  The staged records are read together and the values
// loaded together (walk_load); walk_fold then multiplies / adds them in list order and stores the running values.
// Straight-line code — no branch around a load: a lane outside the box, or a slot past the last hit (record -1: no bits
// set), reads pair 0 and discards it.  gfx950 counts loads and stores in ONE in-order counter, and behind a conditional
// load the compiler can only wait for "everything": every store of a batch then waited for the store before it.
// WIDE = false: pair positions are 32-bit byte offsets from the (wave-uniform) array bases — no 64-bit address
// arithmetic per pair.
struct WalkBatch {
  bool in[kWalkBatch];
  unsigned off[kWalkBatch];
  float v[kWalkBatch];
};
template <int MODE, bool WIDE>
__device__ __forceinline__ void walk_load(WalkBatch& b, unsigned long long& hits, const int4* __restrict__ ent,
                                          unsigned lane_bits, int ly, int lxo, const float* __restrict__ x) {
  int k[kWalkBatch];
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) {  // scalar: the next set bit, -1 when none is left
    if (MODE == 2) {
      k[u] = hits ? 63 - __builtin_clzll(hits) : -1;
      hits &= ~(1ull << (k[u] & 63));
    } else {
      k[u] = hits ? __builtin_ctzll(hits) : -1;
      hits &= hits - 1ull;
    }
  }
  int4 e[kWalkBatch];
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) e[u] = ent[k[u]];
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) {
    asm volatile("" :: "v"(e[u].x), "v"(e[u].y));  // the whole record is read ahead of the membership test
#if GCP_TILE_SX == 4
    b.in[u] = ((unsigned)e[u].z & lane_bits) == lane_bits;
#elif GCP_TILE_SX == 5
    // 32 x 8 tiles, two pixel rows per wave: z = the box over the tile's 32 columns, w = over its 8 rows
    b.in[u] = (((unsigned)e[u].z & lane_bits) != 0u) & (((unsigned)e[u].w & (1u << ly)) != 0u);
#else
    // 64 x 4 tiles, one pixel row per wave (the hit word has decided the row): (z, w) = the box over the tile's 64 columns
    b.in[u] = ((((int)(threadIdx.x & 32) ? (unsigned)e[u].w : (unsigned)e[u].z) & lane_bits) != 0u);
#endif
    const unsigned o = (unsigned)e[u].x + (unsigned)lxo + __umul24((unsigned)ly, (unsigned)e[u].y);
    b.off[u] = b.in[u] ? o : 0u;
#if (GCP_WALK_DBG & 1)
    b.v[u] = __int_as_float(0x3f7fff00 + b.off[u] % 7);
#else
    b.v[u] = WIDE ? x[b.off[u]] : *(const float*)((const char*)x + b.off[u]);
#endif
  }
}
// COUNT: how many of the values just written are exactly 0 — what the `!= 0` compaction that follows drops
// (gs_model.py:560) — per kCompactTile consecutive pairs.  One integer add per wave, list entry and tile that has any
// (none at all in a scene without opaque or underflowing layers): the sums do not depend on the order.
constexpr int kDropTileLog2 = 12;
template <bool WIDE>
__device__ __forceinline__ void walk_count_dropped(bool drop, unsigned off, int* __restrict__ dropped) {
  unsigned long long dm = __ballot(drop);
  if (!dm) return;
  const unsigned blk = off >> (WIDE ? kDropTileLog2 : kDropTileLog2 + 2);
  const int lane = (int)(threadIdx.x & 63);
  while (dm) {
    const int leader = __builtin_ctzll(dm);
    const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)blk, leader);
    const unsigned long long same = __ballot(drop && blk == b);
    if (lane == leader) atomicAdd(dropped + b, __builtin_popcountll(same));
    dm &= ~same;
  }
}
// OUT: what a pair receives.  kWalkInclusive: its inclusive value (gs_model.py:555); kWalkCount: the same, and the zeros
// written are counted per 4096 pairs; kWalkFinal: the FINAL value of _create_alpha_brend — inclusive / self (cumprod,
// gs_model.py:562) or inclusive - self (cumsum, :564): the pair's own value is in a register anyway, and it is the same
// fp32 division / subtraction the compaction pass would do on the stored inclusive value, so the bits are the same — while
// the `!= 0` test of :560 is taken on the inclusive value here: a pair whose inclusive value is exactly 0 clears its byte of
// `keep` (pre-set to 1 by the launcher) and is counted.  A scene that drops nothing is finished after this kernel.
constexpr int kWalkInclusive = 0, kWalkCount = 1, kWalkFinal = 2;
template <int MODE, bool WIDE, int OUT>
__device__ __forceinline__ void walk_fold(const WalkBatch& b, float* __restrict__ out, float& acc, int* __restrict__ dropped,
                                          unsigned char* __restrict__ keep) {
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) {
    bool drop = false;
    if (b.in[u]) {
      acc = (MODE == 0) ? acc * b.v[u] : acc + b.v[u];
      drop = acc == 0.0f;  // NaN is kept, as `!= 0` keeps it
      const float res = (OUT != kWalkFinal) ? acc : (MODE == 0 ? acc / b.v[u] : acc - b.v[u]);
#if (GCP_WALK_DBG & 2)
      if (acc == 12345.678f)
#endif
      {
        if (WIDE) out[b.off[u]] = res;
        else *(float*)((char*)out + b.off[u]) = res;
      }
    }
    if (OUT != kWalkInclusive) {
      if (OUT == kWalkFinal && __ballot(drop) != 0ull) {  // wave-uniform: no store instruction at all where nothing drops
        if (drop) keep[WIDE ? b.off[u] : (b.off[u] >> 2)] = 0;
      }
      walk_count_dropped<WIDE>(drop, b.off[u], dropped);
    }
  }
}

// Which tile a block of the walk takes.  Blocks are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), and tiles
// that share cache lines should share an L2 (see the kernel).  xcd_remap 1: XCD x takes the x-th contiguous eighth of the
// tiles — a band of tile rows.  xcd_remap >= 2: STRIPES of 2^(xcd_remap - 2) tile rows dealt round-robin to the XCDs, so that
// every XCD holds stripes from all over the image: a scene whose Gaussians crowd one region (bands: the XCDs of that region
// do most of the work while the others idle) is spread evenly, and all but the lines that straddle a stripe edge still
// share an L2.  0: tile = block.
__device__ __forceinline__ int walk_tile(unsigned b, int n_tiles, int tiles_x, int xcd_remap) {
  if (xcd_remap < 2) return (int)sort_chunk(b, n_tiles, xcd_remap);
  const int sh = xcd_remap - 2;                  // log2 of the tile rows per stripe
  const int x = (int)(b & 7u), j = (int)(b >> 3);  // XCD, and the block's number on it
  const int per_stripe = tiles_x << sh;
  const int stripe = (j / per_stripe) * 8 + x, within = j % per_stripe;
  const int tile = stripe * per_stripe + within;
  return tile < n_tiles ? tile : -1;
}
inline unsigned walk_grid(int n_tiles, int tiles_x, int xcd_remap) {
  if (xcd_remap < 2) return sort_grid(n_tiles, xcd_remap);
  const int per_stripe = tiles_x << (xcd_remap - 2);
  const int stripes = (n_tiles + per_stripe - 1) / per_stripe;
  return (unsigned)(((stripes + 7) / 8) * 8 * per_stripe);
}

// A wave all of whose 64 pixels have reached a product of exactly 0 (cumprod: it stays 0 — behind an opaque pair, or where the
// product has underflowed, hundreds of layers deep) has nothing left to compute: every further pair of its strip is dropped
// whatever its value.  The rest of the tile's list then costs it one byte per pair — the pair's `keep` byte is cleared and the
// drop counted — instead of a 4-byte load, a multiplication and a 4-byte store (whose result nobody reads: the compaction
// that follows moves kept values only).  Scenes that crowd one region drop a large share of their pairs this way.
template <bool WIDE>
__device__ __forceinline__ void walk_dead(unsigned long long hits, const int4* __restrict__ ent, unsigned lane_bits, int ly, int lxo,
                                          int* __restrict__ dropped, unsigned char* __restrict__ keep) {
  while (hits) {
    const int k = __builtin_ctzll(hits);
    hits &= hits - 1ull;
    const int4 e = ent[k];
#if GCP_TILE_SX == 4
    const bool in = ((unsigned)e.z & lane_bits) == lane_bits;
#elif GCP_TILE_SX == 5
    const bool in = (((unsigned)e.z & lane_bits) != 0u) & (((unsigned)e.w & (1u << ly)) != 0u);
#else
    const bool in = ((((int)(threadIdx.x & 32) ? (unsigned)e.w : (unsigned)e.z) & lane_bits) != 0u);
#endif
    const unsigned o = (unsigned)e.x + (unsigned)lxo + __umul24((unsigned)ly, (unsigned)e.y);
    if (in) keep[WIDE ? o : (o >> 2)] = 0;
    walk_count_dropped<WIDE>(in, in ? o : 0u, dropped);
  }
}

template <int MODE, bool WIDE, int OUT>  // MODE 0 cumprod, 1 cumsum, 2 reverse cumsum; WIDE: more than 2^30 pairs
// (pinned to eight waves per SIMD the byte-offset form fits 63 VGPRs without a spill — and runs no faster: 0.62 ms either way)
__global__ __launch_bounds__(kWalkThreads) void k_pairs_scan_boxes(const BlendArgs a, const int* __restrict__ box_off,
                                                                   const float* __restrict__ x, float* __restrict__ out,
                                                                   int* __restrict__ dropped, unsigned char* __restrict__ keep,
                                                                   int n_tiles, int xcd_remap) {
  // a staged entry: x = position of the tile's first pixel in the entry's box run (box_off + (tile_y0 - y0) * width +
  // (tile_x0 - x0), may lie before the run), y = box width — both in bytes unless WIDE —, z = the box as bits over the
  // tile's columns (0-15) and rows (16-31).  A lane's pair is x + row * y + column, and it is in the box when both of
  // its bits are set: membership is one AND and one compare.
  __shared__ int4 s_ent_[kWalkStage + 1];
  __shared__ unsigned long long s_hits[kWalkThreads / 64];
  int4* const s_ent = s_ent_ + 1;  // record -1: no bits set, what a batch reads for the slots past its last hit
  if (threadIdx.x == 0) s_ent[-1] = make_int4(0, 0, 0, 0);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // A box is one contiguous run of the pair arrays (row-major, uitility.py:336-366) but lies across up to 2 x 2 tiles: with
  // blocks dealt round-robin over the XCDs its pieces would be written from different L2s and reach memory as partial
  // lines (measured: 2.6x the algorithmic bytes).  An XCD therefore takes whole tile ROWS (walk_tile): the two pieces of a
  // box row, which share cache lines, always meet in one L2; only the line a box's rows 15 | 16 of a tile-row pair straddle
  // is written from two.
  const int tile = walk_tile(blockIdx.x, n_tiles, a.tiles_x, xcd_remap);
  if (tile < 0) return;
  const int tile_x0 = (tile % a.tiles_x) * kTileW, tile_y0 = (tile / a.tiles_x) * kTileH;
  constexpr int kUnit = WIDE ? 1 : 4;
  // pixel rows per wave: 4 (16 x 16 tiles, and super-tiles: two waves side by side), 2 (32 x 8), 1 (64 x 4)
  constexpr int kWaveCols = kWalkSuper ? 16 : kTileW, kWaveRows = 64 / kWaveCols, kWavesAcross = kTileW / kWaveCols;
  const int lx = (w % kWavesAcross) * kWaveCols + (lane & (kWaveCols - 1));
  const int lxo = lx * kUnit, ly = (w / kWavesAcross) * kWaveRows + lane / kWaveCols;
#if GCP_TILE_SX == 4
  const unsigned lane_bits = (1u << (lane & 15)) | (1u << (16 + ly));
#else
  const unsigned lane_bits = 1u << (lx & 31);
#endif
  const int first = a.tile_start[tile], last = a.tile_start[tile + 1];
  const int nrounds = (last - first + kWalkStage - 1) / kWalkStage;
  float acc = (MODE == 0) ? 1.0f : 0.0f;
  for (int q0 = 0; q0 < nrounds; ++q0) {
    const int q = (MODE == 2) ? (nrounds - 1 - q0) : q0;
    const int base = first + q * kWalkStage;
    const int cnt = min(kWalkStage, last - base);
    __syncthreads();
    if (w == 0) {  // wave 0 stages the round (a fifth wave staging one round ahead of the walkers: measured 6 % slower)
      unsigned rm = 0u, cmask32 = 0u;
      if (lane < cnt) {
        const i64 g = a.tile_list[base + lane];
        Box b;
        load_box(a.start, a.end, g, a.W, a.H, b);
        const int wd = b.x1 - b.x0 + 1;
        const int c0 = max(b.x0 - tile_x0, 0), c1 = min(b.x1 - tile_x0, kTileW - 1);
        const int r0 = max(b.y0 - tile_y0, 0), r1 = min(b.y1 - tile_y0, kTileH - 1);
        const unsigned long long cm = (c1 >= c0) ? ((2ull << c1) - (1ull << c0)) : 0ull;
        rm = (r1 >= r0 && cm) ? ((2u << r1) - (1u << r0)) : 0u;
        cmask32 = (unsigned)cm;
        // modulo 2^32: every pair of the list lies below 2^32 bytes (2^31 pairs when WIDE), whatever the tile's corner does
        const unsigned p0 = (unsigned)box_off[g] + (unsigned)(tile_y0 - b.y0) * (unsigned)wd + (unsigned)(tile_x0 - b.x0);
#if GCP_TILE_SX == 4
        s_ent[lane] = make_int4((int)(p0 * (unsigned)kUnit), wd * kUnit, (int)((unsigned)cm | (rm << 16)), 0);
#elif GCP_TILE_SX == 5
        s_ent[lane] = make_int4((int)(p0 * (unsigned)kUnit), wd * kUnit, (int)(unsigned)cm, (int)rm);
#else
        s_ent[lane] = make_int4((int)(p0 * (unsigned)kUnit), wd * kUnit, (int)(unsigned)cm, (int)(unsigned)(cm >> 32));
#endif
      }
#pragma unroll
      for (int w2 = 0; w2 < kWalkThreads / 64; ++w2) {
        const bool rows_hit = ((rm >> (kWaveRows * (w2 / kWavesAcross))) & ((1u << kWaveRows) - 1u)) != 0u;
        const bool cols_hit = !kWalkSuper || ((cmask32 >> (16 * (w2 % kWavesAcross))) & 0xffffu) != 0u;
        const unsigned long long touched = __ballot(rows_hit && cols_hit);
        if (lane == 0) s_hits[w2] = touched;
      }
    }
    __syncthreads();
    unsigned long long hits = uniform64(s_hits[w]);
    if (MODE == 0 && OUT == kWalkFinal && hits && __ballot(acc != 0.0f) == 0ull) {  // wave-uniform, decided once per round
      walk_dead<WIDE>(hits, s_ent, lane_bits, ly, lxo, dropped, keep);
      continue;
    }
    // two batches in flight: the loads of the next one are issued before the stores of the one in hand, so that waiting
    // for loaded values (in-order counter) never waits for the stores just issued
    if (hits) {
      WalkBatch A, B;
      walk_load<MODE, WIDE>(A, hits, s_ent, lane_bits, ly, lxo, x);
      for (;;) {
        if (!hits) { walk_fold<MODE, WIDE, OUT>(A, out, acc, dropped, keep); break; }
        walk_load<MODE, WIDE>(B, hits, s_ent, lane_bits, ly, lxo, x);
        walk_fold<MODE, WIDE, OUT>(A, out, acc, dropped, keep);
        if (!hits) { walk_fold<MODE, WIDE, OUT>(B, out, acc, dropped, keep); break; }
        walk_load<MODE, WIDE>(A, hits, s_ent, lane_bits, ly, lxo, x);
        walk_fold<MODE, WIDE, OUT>(B, out, acc, dropped, keep);
      }
    }
  }
}

#if GCP_WALK_DENSE
// ---- measurement build: the DENSE walk on tiles of kTileW x kTileH pixels (<= 1024) --------------------------------------
// One wave per tile; its lanes are laid over the PAIRS of an entry's box inside the tile, row-major, so that a box that is not
// cut by the tile's left / right edge is read and written as ONE contiguous run across its rows (consecutive box rows are
// adjacent in memory): with 64-pixel-wide tiles four boxes in five.  The pixels' running values live in LDS.
struct DenseBatch {
  bool in[kWalkBatch];
  unsigned off[kWalkBatch];
  int slot[kWalkBatch];
  float v[kWalkBatch];
};
struct DenseCursor {
  unsigned long long hits;
  int k, it, nit;
  __device__ __forceinline__ bool more() const { return hits != 0ull || (k >= 0 && it + 1 < nit); }
};
template <int MODE, bool WIDE>
__device__ __forceinline__ void dense_load(DenseBatch& b, DenseCursor& c, const int4& rec, int lane, const float* __restrict__ x) {
  constexpr int kUnit = WIDE ? 1 : 4;
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) {
    if (c.k >= 0 && c.it + 1 < c.nit) {
      ++c.it;
    } else {
      if (MODE == 2) {
        c.k = c.hits ? 63 - __builtin_clzll(c.hits) : -1;
        c.hits &= ~(1ull << (c.k & 63));
      } else {
        c.k = c.hits ? __builtin_ctzll(c.hits) : -1;
        c.hits &= c.hits - 1ull;
      }
      c.it = 0;
      c.nit = c.k < 0 ? 0 : ((__builtin_amdgcn_readlane(rec.z, c.k & 63) >> 20) & 31);
    }
    const int ep = __builtin_amdgcn_readlane(rec.x, c.k & 63), ewd = __builtin_amdgcn_readlane(rec.y, c.k & 63);
    const int geo = c.k < 0 ? 0 : __builtin_amdgcn_readlane(rec.z, c.k & 63);  // cw | pairs << 8 | parts << 20
    const int iw = __builtin_amdgcn_readlane(rec.w, c.k & 63);                  // 65536 / cw rounded up | first pixel slot << 17
    const int cw = geo & 255, tot = (geo >> 8) & 4095, inv = iw & 0x1ffff, sbase = (int)((unsigned)iw >> 17);
    const int i = lane + 64 * c.it;
    const int row = (int)(__umul24((unsigned)i, (unsigned)inv) >> 16);  // i / cw, exact for i < 1100 and cw <= 64
    const int col = i - row * cw;
    b.in[u] = i < tot;
    const unsigned o = (unsigned)ep + (unsigned)row * (unsigned)ewd + (unsigned)(col * kUnit);
    b.off[u] = b.in[u] ? o : 0u;
    b.slot[u] = sbase + row * kTileW + col;
    b.v[u] = WIDE ? x[b.off[u]] : *(const float*)((const char*)x + b.off[u]);
  }
}
template <int MODE, bool WIDE, int OUT>
__device__ __forceinline__ void dense_fold(const DenseBatch& b, float* __restrict__ out, float* __restrict__ acc_lds,
                                           int* __restrict__ dropped, unsigned char* __restrict__ keep) {
#pragma unroll
  for (int u = 0; u < kWalkBatch; ++u) {
    bool drop = false;
    if (b.in[u]) {
      float acc = acc_lds[b.slot[u]];
      acc = (MODE == 0) ? acc * b.v[u] : acc + b.v[u];
      acc_lds[b.slot[u]] = acc;
      drop = acc == 0.0f;
      const float res = (OUT != kWalkFinal) ? acc : (MODE == 0 ? acc / b.v[u] : acc - b.v[u]);
      if (WIDE) out[b.off[u]] = res;
      else *(float*)((char*)out + b.off[u]) = res;
    }
    if (OUT != kWalkInclusive) {
      if (OUT == kWalkFinal && __ballot(drop) != 0ull) {
        if (drop) keep[WIDE ? b.off[u] : (b.off[u] >> 2)] = 0;
      }
      walk_count_dropped<WIDE>(drop, b.off[u], dropped);
    }
  }
}
template <int MODE, bool WIDE, int OUT>
__global__ __launch_bounds__(256) void k_pairs_walk_dense(const BlendArgs a, const int* __restrict__ box_off, const float* __restrict__ x,
                                                          float* __restrict__ out, int* __restrict__ dropped,
                                                          unsigned char* __restrict__ keep, int n_tiles, int n_blocks, int xcd_remap) {
  constexpr int kPix = kTileW * kTileH;
  static_assert(kPix <= 1024 && kTileW <= 64, "dense walk: tiles of at most 1024 pixels, 64 wide");
  __shared__ float s_acc[4][kPix];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int chunk = (int)sort_chunk(blockIdx.x, n_blocks, xcd_remap);
  if (chunk < 0) return;
  const int tile = 4 * chunk + w;
  if (tile >= n_tiles) return;
  const int first = a.tile_start[tile], n = a.tile_start[tile + 1] - first;
  if (n <= 0) return;
  float* const acc = s_acc[w];
#pragma unroll
  for (int j = 0; j < kPix / 64; ++j) acc[lane + 64 * j] = (MODE == 0) ? 1.0f : 0.0f;
  const int tile_x0 = (tile % a.tiles_x) * kTileW, tile_y0 = (tile / a.tiles_x) * kTileH;
  constexpr int kUnit = WIDE ? 1 : 4;
  const int nrounds = (n + kWalkStage - 1) / kWalkStage;
  for (int q0 = 0; q0 < nrounds; ++q0) {
    const int q = (MODE == 2) ? (nrounds - 1 - q0) : q0;
    const int base = first + q * kWalkStage;
    const int cnt = min(kWalkStage, first + n - base);
    int4 rec = make_int4(0, 0, 0, 0);
    bool valid = false;
    if (lane < cnt) {
      const i64 g = a.tile_list[base + lane];
      Box b;
      load_box(a.start, a.end, g, a.W, a.H, b);
      const int wd = b.x1 - b.x0 + 1;
      const int c0 = max(b.x0 - tile_x0, 0), c1 = min(b.x1 - tile_x0, kTileW - 1);
      const int r0 = max(b.y0 - tile_y0, 0), r1 = min(b.y1 - tile_y0, kTileH - 1);
      const int cw = c1 - c0 + 1, ch = r1 - r0 + 1;
      valid = cw > 0 && ch > 0;
      if (valid) {
        const unsigned p0 = (unsigned)box_off[g] + (unsigned)(tile_y0 + r0 - b.y0) * (unsigned)wd + (unsigned)(tile_x0 + c0 - b.x0);
        const int tot = cw * ch;
        rec = make_int4((int)(p0 * (unsigned)kUnit), wd * kUnit, cw | (tot << 8) | (((tot + 63) >> 6) << 20),
                        (int)((65535u / (unsigned)cw + 1u) | ((unsigned)(r0 * kTileW + c0) << 17)));
      }
    }
    DenseCursor c;
    c.hits = __ballot(valid);
    c.k = -1; c.it = 0; c.nit = 0;
    if (c.hits) {
      DenseBatch A, B;
      dense_load<MODE, WIDE>(A, c, rec, lane, x);
      for (;;) {
        if (!c.more()) { dense_fold<MODE, WIDE, OUT>(A, out, acc, dropped, keep); break; }
        dense_load<MODE, WIDE>(B, c, rec, lane, x);
        dense_fold<MODE, WIDE, OUT>(A, out, acc, dropped, keep);
        if (!c.more()) { dense_fold<MODE, WIDE, OUT>(B, out, acc, dropped, keep); break; }
        dense_load<MODE, WIDE>(A, c, rec, lane, x);
        dense_fold<MODE, WIDE, OUT>(B, out, acc, dropped, keep);
      }
    }
  }
}
#endif  // GCP_WALK_DENSE

// Gaussian-major rect list (reference: Utilities.make_rect_points_parallel, uitility.py:336-366, called by
// _create_rects, gs_model.py:480-482): pair i of Gaussian g is pixel (x0 + i % w, y0 + i / w) of its box.
// Parallel over the BOXES, not over the pairs: a block takes kExpandBoxes consecutive Gaussians, each wave writes one box
// at a time — its lanes over consecutive pairs, 512 contiguous bytes per store instruction — and a box of more than
// kExpandBig pairs is written by the whole block.  Nothing is searched (one thread per pair had to find its Gaussian by a
// 20-step bisection of dependent loads in the box offsets: 2.2 ms for 1.65e8 pairs, 0.6 TB/s of stores), and the box's
// width is wave-uniform: i / w is one multiplication by its reciprocal and one correction step (exact: i < 2^24).
constexpr int kExpandBoxes = 64;
constexpr int kExpandBig = 8192;

template <bool BIG>
__device__ __forceinline__ void expand_box(const int* __restrict__ start, const int* __restrict__ end, const int* __restrict__ box_off, i64 g,
                                           i64 m, int W, int H, int t, int step, int2* __restrict__ rects, int* __restrict__ pair_gauss) {
  Box b;
  if (!load_box(start, end, g, W, H, b)) return;
  const int bw = b.x1 - b.x0 + 1;
  i64 size = (i64)bw * (b.y1 - b.y0 + 1);
  if ((size > kExpandBig) != BIG) return;
  const i64 off = box_off[g];
  if (off < 0 || off > m) return;  // (offsets that do not belong to these boxes: nothing is written outside the list)
  size = min(size, m - off);
  if (size < (1 << 24)) {
    const float rw = 1.0f / (float)bw;
    for (int l = t; l < (int)size; l += step) {
      int q = (int)((float)l * rw), r = l - q * bw;
      if (r < 0) { --q; r += bw; }
      else if (r >= bw) { ++q; r -= bw; }
      rects[off + l] = make_int2(b.x0 + r, b.y0 + q);
      if (pair_gauss) pair_gauss[off + l] = (int)g;
    }
  } else {
    for (i64 l = t; l < size; l += step) {
      const i64 q = l / bw;
      rects[off + l] = make_int2(b.x0 + (int)(l - q * bw), b.y0 + (int)q);
      if (pair_gauss) pair_gauss[off + l] = (int)g;
    }
  }
}

__global__ __launch_bounds__(256) void k_expand_rects(const int* __restrict__ start, const int* __restrict__ end, const int* __restrict__ box_off,
                                                      i64 n_gauss, i64 m, int W, int H, int2* __restrict__ rects /*[m]*/,
                                                      int* __restrict__ pair_gauss /*[m] or null*/) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const i64 b0 = (i64)blockIdx.x * kExpandBoxes, b1 = min(n_gauss, b0 + kExpandBoxes);
  for (i64 g = b0 + w; g < b1; g += 4) expand_box<false>(start, end, box_off, g, m, W, H, lane, 64, rects, pair_gauss);
  for (i64 g = b0; g < b1; ++g) expand_box<true>(start, end, box_off, g, m, W, H, (int)threadIdx.x, 256, rects, pair_gauss);
}

// ---- the index plumbing around the scan in _create_alpha_brend (gs_model.py:548, :555-564) --------------
// sorted values: dst[i] = src[index[i]]   (gs_model.py:548 `anti_opacity[index]`), int32 permutation
__global__ void k_gather_f32(const float* __restrict__ src, const int* __restrict__ index, float* __restrict__ dst, i64 n) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[index[i]];
}
// un-sort + the two element-wise steps that follow the compaction, fused (they commute with it):
//   full[index[i]] = inclusive[i] / self  (mode 0, gs_model.py:562)  or  inclusive[i] - self  (mode 1, :564)
//   keep[index[i]] = inclusive[i] != 0                                  (gs_model.py:560, :575-578)
// where self = sorted_x[i] is the pair's own input value (= anti_opacity[index[i]]).
__global__ void k_unsort_finish(const float* __restrict__ incl, const float* __restrict__ sorted_x,
                                const int* __restrict__ index, float* __restrict__ full, unsigned char* __restrict__ keep,
                                i64 n, int mode) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = incl[i], x = sorted_x[i];
  const int o = index[i];
  full[o] = mode == 0 ? v / x : v - x;
  keep[o] = v != 0.0f ? 1 : 0;
}

// max of the pixel keys and min of the coordinates of a rect list (integer max / min: order-independent)
__global__ __launch_bounds__(256) void k_rects_key_range(const int* rects, i64 n, int* out /*[2] = {max key, min coordinate}*/) {
  __shared__ int s_mx[4], s_mn[4];
  int mx = 0, mn = 0x7fffffff;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const int2 r = reinterpret_cast<const int2*>(rects)[i];
    mx = max(mx, r.y * 10000 + r.x);
    mn = min(mn, min(r.x, r.y));
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = max(mx, __shfl_xor(mx, o));
    mn = min(mn, __shfl_xor(mn, o));
  }
  if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_mn[threadIdx.x >> 6] = mn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(out, max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])));
    atomicMin(out + 1, min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3])));
  }
}

// ---- the tail of _create_alpha_brend (gs_model.py:557-564): `!= 0` mask, boolean-mask compaction, / self or - self ----
// One 256-thread block per tile of kCompactTile elements of the ORIGINAL pair order; wave w owns elements
// [1024 w, 1024 w + 1024) of the tile as 4 rows of 64 lanes x 4 consecutive elements (16-byte loads).  WRITE = false:
// the tile's kept count.  WRITE = true: every kept element's rank = tile offset (exclusive scan of the counts) + kept
// elements before it in the tile (per-lane popcounts -> wave prefix in DPP -> 4 LDS words), its value written to that
// slot — neighbouring lanes write neighbouring slots — and the mask as one packed word per lane.
#ifndef GCP_COMPACT_X4
#define GCP_COMPACT_X4 1
#endif
constexpr int kCompactTile = 1 << kDropTileLog2;
template <bool VEC, bool WRITE>
__global__ __launch_bounds__(256) void k_compact(const float* __restrict__ incl, const float* __restrict__ self, i64 n, int mode,
                                                 int* __restrict__ cnt, const int* __restrict__ off, float* __restrict__ values,
                                                 unsigned char* __restrict__ keep, int* __restrict__ count_dev) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 base = (i64)blockIdx.x * kCompactTile + (i64)w * 1024;
  float v[4][4], x[4][4];
  unsigned m[4];
  int c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    if (VEC && p + 3 < n) {
      const float4 a = *reinterpret_cast<const float4*>(incl + p);
      v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w;
      if (WRITE) {
        const float4 b = *reinterpret_cast<const float4*>(self + p);
        x[r][0] = b.x; x[r][1] = b.y; x[r][2] = b.z; x[r][3] = b.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[r][k] = (p + k < n) ? incl[p + k] : 0.0f;
        if (WRITE) x[r][k] = (p + k < n) ? self[p + k] : 1.0f;
      }
    }
    m[r] = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) m[r] |= (v[r][k] != 0.0f ? 1u : 0u) << k;  // NaN != 0 is true, as in torch (gs_model.py:577)
    c[r] = __builtin_popcount(m[r]);
  }
  // kept elements before this lane inside the wave's 1024: rows in order, lanes in order inside a row
  int before[4];
  int wtot = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int inc = wave_incl_scan_i(c[r]);
    before[r] = wtot + inc - c[r];
    wtot += __builtin_amdgcn_readlane(inc, 63);
  }
  if (lane == 0) s_w[w] = wtot;
  __syncthreads();
  if (!WRITE) {
    if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return;
  }
  int woff = off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < w) woff += s_w[j];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) count_dev[0] = off[blockIdx.x] + s_w[0] + s_w[1] + s_w[2] + s_w[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    int o = woff + before[r];
#if GCP_COMPACT_X4
    if (m[r] == 0xfu) {
      // all four kept (every lane of a stretch that drops nothing): one 16-byte store at a 4-byte-aligned slot — a wave
      // then writes 1 KB contiguous with one instruction instead of four strided ones
      typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
      float4_u q;
      q.x = mode == 0 ? v[r][0] / x[r][0] : v[r][0] - x[r][0];
      q.y = mode == 0 ? v[r][1] / x[r][1] : v[r][1] - x[r][1];
      q.z = mode == 0 ? v[r][2] / x[r][2] : v[r][2] - x[r][2];
      q.w = mode == 0 ? v[r][3] / x[r][3] : v[r][3] - x[r][3];
      *reinterpret_cast<float4_u*>(values + o) = q;
    } else
#endif
    {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if ((m[r] >> k) & 1u) {
          values[o] = mode == 0 ? v[r][k] / x[r][k] : v[r][k] - x[r][k];
          ++o;
        }
      }
    }
    if (VEC && p + 3 < n) {
      // bytes 0/1 per element, memory order: bit k of m -> byte k
      *reinterpret_cast<unsigned*>(keep + p) = (m[r] & 1u) | ((m[r] & 2u) << 7) | ((m[r] & 4u) << 14) | ((m[r] & 8u) << 21);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p + k < n) keep[p + k] = (unsigned char)((m[r] >> k) & 1u);
    }
  }
}

// ---- the compaction that is left when the walk has written final values: only where something was dropped ---------------
// per-tile kept counts from the keep bytes of [begin, end) (tile t = elements [begin + 4096 t, ...)): 1 B per element
__global__ __launch_bounds__(256) void k_count_keep(const unsigned char* __restrict__ keep, i64 n, int* __restrict__ cnt) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 p = (i64)blockIdx.x * kCompactTile + (i64)threadIdx.x * 16;
  int c = 0;
  if ((((uintptr_t)keep) & 15u) == 0 && p + 15 < n) {
    const uint4 q = *reinterpret_cast<const uint4*>(keep + p);
    c = __builtin_popcount(q.x & 0x01010101u) + __builtin_popcount(q.y & 0x01010101u) + __builtin_popcount(q.z & 0x01010101u) +
        __builtin_popcount(q.w & 0x01010101u);
  } else {
    for (int k = 0; k < 16; ++k) c += (p + k < n && keep[p + k]) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if (lane == 0) s_w[w] = c;
  __syncthreads();
  if (threadIdx.x == 0) cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// kept values of [begin, end) to their ranks (tile offsets from the exclusive scan of the counts + ranks inside the tile)
template <bool VEC>
__global__ __launch_bounds__(256) void k_compact_kept(const float* __restrict__ vals, const unsigned char* __restrict__ keep, i64 n,
                                                      const int* __restrict__ off, float* __restrict__ out) {
  __shared__ int s_w[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 base = (i64)blockIdx.x * kCompactTile + (i64)w * 1024;
  float v[4][4];
  unsigned m[4];
  int c[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const i64 p = base + r * 256 + lane * 4;
    m[r] = 0u;
    if (VEC && p + 3 < n) {
      const float4 a = *reinterpret_cast<const float4*>(vals + p);
      v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w;
      const unsigned q = *reinterpret_cast<const unsigned*>(keep + p);  // bytes 0 / 1, memory order
      m[r] = (q & 1u) | ((q >> 7) & 2u) | ((q >> 14) & 4u) | ((q >> 21) & 8u);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[r][k] = (p + k < n) ? vals[p + k] : 0.0f;
        m[r] |= ((p + k < n && keep[p + k]) ? 1u : 0u) << k;
      }
    }
    c[r] = __builtin_popcount(m[r]);
  }
  int before[4];
  int wtot = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int inc = wave_incl_scan_i(c[r]);
    before[r] = wtot + inc - c[r];
    wtot += __builtin_amdgcn_readlane(inc, 63);
  }
  if (lane == 0) s_w[w] = wtot;
  __syncthreads();
  int woff = off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < w) woff += s_w[j];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int o = woff + before[r];
    if (m[r] == 0xfu) {
      typedef float float4_u __attribute__((ext_vector_type(4), aligned(4)));
      float4_u q;
      q.x = v[r][0]; q.y = v[r][1]; q.z = v[r][2]; q.w = v[r][3];
      *reinterpret_cast<float4_u*>(out + o) = q;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((m[r] >> k) & 1u) out[o++] = v[r][k];
    }
  }
}

__global__ void k_box_sizes(const int* start, const int* end, i64 n, int W, int H, int* size) {
  const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  Box b;
  size[g] = load_box(start, end, g, W, H, b) ? (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) : 0;
}

inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" {

int gcp_tile_grid(int32_t width, int32_t height, int32_t* tiles_x, int32_t* tiles_y) {
  if (width < 0 || height < 0 || !tiles_x || !tiles_y) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid t = tile_grid(width, height);
  *tiles_x = t.tx;
  *tiles_y = t.ty;
  return GCP_OK;
}

size_t gcp_scan_i32_workspace_bytes(int64_t n) {
  return align256((size_t)((n + kScanChunk - 1) / kScanChunk + 1) * sizeof(int));
}

int gcp_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes, void* stream) {
  if (n < 0 || !out || (n > 0 && (!in || !ws))) return GCP_ERR_INVALID_ARGUMENT;
  if (n > 0 && ws_bytes < gcp_scan_i32_workspace_bytes(n)) return GCP_ERR_WORKSPACE;
  return launch_excl_scan(in, out, n, (int*)ws, (hipStream_t)stream);
}

int gcp_bin_tiles_count(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                        int32_t height, int32_t* tile_off, int64_t* n_tile_pairs_host, void* ws,
                        size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_gauss < 0 || width < 0 || height < 0 || !tile_off || !n_tile_pairs_host) return GCP_ERR_INVALID_ARGUMENT;
  *n_tile_pairs_host = 0;
  if (n_gauss == 0) {
    GCP_HIP(hipMemsetAsync(tile_off, 0, sizeof(int), stream));
    return GCP_OK;
  }
  if (!start_xy || !end_xy || !ws) return GCP_ERR_INVALID_ARGUMENT;
  const size_t need = 256 + align256((size_t)n_gauss * sizeof(int)) + gcp_scan_i32_workspace_bytes(n_gauss);
  if (ws_bytes < need) return GCP_ERR_WORKSPACE;
  unsigned long long* total64 = (unsigned long long*)ws;
  int* cnt = (int*)((char*)ws + 256);
  int* sws = (int*)((char*)ws + 256 + align256((size_t)n_gauss * sizeof(int)));
  GCP_HIP(hipMemsetAsync(total64, 0, sizeof(unsigned long long), stream));
  const i64 count_blocks = (n_gauss + 255) / 256 < 512 ? (n_gauss + 255) / 256 : 512;
  hipLaunchKernelGGL(k_tile_count, dim3((unsigned)count_blocks), dim3(256), 0, stream, start_xy, end_xy,
                     (i64)n_gauss, width, height, cnt, total64);
  GCP_HIP(hipGetLastError());
  const int st = launch_excl_scan(cnt, tile_off, n_gauss, sws, stream);
  if (st != GCP_OK) return st;
  unsigned long long total = 0;
  GCP_HIP(hipMemcpyAsync(&total, total64, sizeof(total), hipMemcpyDeviceToHost, stream));
  GCP_HIP(hipStreamSynchronize(stream));
  if (total > 0x7fffffffull) return GCP_ERR_INVALID_ARGUMENT;  // (tile, Gaussian) entries are indexed with int32
  *n_tile_pairs_host = (int64_t)total;
  return GCP_OK;
}

size_t gcp_bin_workspace_bytes(int64_t n_gauss, int64_t n_tile_pairs) {
  const int64_t k = n_tile_pairs > 0 ? n_tile_pairs : 1;
  const int64_t nblk = (k + kSortChunk - 1) / kSortChunk;
  size_t b = 0;
  b += 3 * align256((size_t)k * sizeof(unsigned));                 // key A, key B, val B
  b += 2 * align256((size_t)(256 * nblk + 1) * sizeof(int));       // hist, hist_excl
  b += gcp_scan_i32_workspace_bytes(256 * nblk);
  const size_t count_need = 256 + align256((size_t)(n_gauss > 0 ? n_gauss : 1) * sizeof(int)) +
                            gcp_scan_i32_workspace_bytes(n_gauss > 0 ? n_gauss : 1);
  return (b > count_need ? b : count_need) + 512;  // + room for gcp_bin_tiles' 64-bit total behind everything else
}

// emit + stable sort by tile + tile bounds; `info` non-null = capture-safe mode (K is the capacity, the real count is info[0])
static int bin_fill(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                    const int32_t* tile_off, int64_t K, int32_t* tile_start, int32_t* tile_list, int* info,
                    const unsigned long long* total64, void* ws, hipStream_t stream) {
  const TileGrid tg = tile_grid(width, height);
  const int n_tiles = tg.tx * tg.ty;
  const i64 nblk = (K + kSortChunk - 1) / kSortChunk;
  char* p = (char*)ws;
  unsigned* keyA = (unsigned*)p; p += align256((size_t)K * sizeof(unsigned));
  unsigned* keyB = (unsigned*)p; p += align256((size_t)K * sizeof(unsigned));
  unsigned* valB = (unsigned*)p; p += align256((size_t)K * sizeof(unsigned));
  int* hist = (int*)p; p += align256((size_t)(256 * nblk + 1) * sizeof(int));
  int* hist_ex = (int*)p; p += align256((size_t)(256 * nblk + 1) * sizeof(int));
  int* sws = (int*)p;
  unsigned* valA = (unsigned*)tile_list;  // the caller's output buffer doubles as one value buffer
  const int* n_dev = info;                // info[0] = entries listed

  hipLaunchKernelGGL(k_tile_emit, dim3((unsigned)((n_gauss + 255) / 256)), dim3(256), 0, stream, start_xy, end_xy,
                     (i64)n_gauss, width, height, tg.tx, tile_off, keyA, valA, (i64)K, info, total64);
  GCP_HIP(hipGetLastError());
  int bits = 1;
  while ((1 << bits) < n_tiles) ++bits;
  int passes = (bits + 7) / 8;
  if (passes & 1) ++passes;  // even number of passes: the result lands back in (keyA, valA = tile_list)
  unsigned *ks = keyA, *vs = valA, *kd = keyB, *vd = valB;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL((k_sort_hist<false>), dim3((unsigned)nblk), dim3(256), 0, stream, (const unsigned*)ks, K, shift, hist, (int)nblk, n_dev, 0);
    GCP_HIP(hipGetLastError());
    const int st = launch_excl_scan(hist, hist_ex, 256 * nblk, sws, stream);
    if (st != GCP_OK) return st;
    hipLaunchKernelGGL((k_sort_scatter<false>), dim3((unsigned)nblk), dim3(256), 0, stream, (const unsigned*)ks,
                       (const unsigned*)vs, kd, vd, K, shift, (const int*)hist_ex, (int)nblk, n_dev, 0);
    GCP_HIP(hipGetLastError());
    unsigned* t;
    t = ks; ks = kd; kd = t;
    t = vs; vs = vd; vd = t;
  }
  hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned)((K + 1 + 255) / 256)), dim3(256), 0, stream, ks, K, n_tiles,
                     tile_start, n_dev);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_bin_tiles_fill(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                       int32_t height, const int32_t* tile_off, int64_t n_tile_pairs, int32_t* tile_start,
                       int32_t* tile_list, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_gauss < 0 || n_tile_pairs < 0 || width < 0 || height < 0 || !tile_start) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  const int n_tiles = tg.tx * tg.ty;
  const i64 K = n_tile_pairs;
  if (K == 0) {
    GCP_HIP(hipMemsetAsync(tile_start, 0, (size_t)(n_tiles + 1) * sizeof(int), stream));
    return GCP_OK;
  }
  if (!start_xy || !end_xy || !tile_off || !tile_list || !ws) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_bin_workspace_bytes(n_gauss, K)) return GCP_ERR_WORKSPACE;
  return bin_fill(start_xy, end_xy, n_gauss, width, height, tile_off, K, tile_start, tile_list, nullptr, nullptr, ws, stream);
}

int gcp_bin_tiles(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                  int64_t capacity, int32_t* tile_off, int32_t* tile_start, int32_t* tile_list, int32_t* info, void* ws,
                  size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_gauss < 0 || width < 0 || height < 0 || capacity < 1 || capacity > 0x7fffffffLL || !tile_off || !tile_start || !tile_list ||
      !info || !ws)
    return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  const int n_tiles = tg.tx * tg.ty;
  if (n_gauss == 0) {
    GCP_HIP(hipMemsetAsync(tile_off, 0, sizeof(int), stream));
    GCP_HIP(hipMemsetAsync(tile_start, 0, (size_t)(n_tiles + 1) * sizeof(int), stream));
    GCP_HIP(hipMemsetAsync(info, 0, 2 * sizeof(int), stream));
    return GCP_OK;
  }
  if (!start_xy || !end_xy) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_bin_workspace_bytes(n_gauss, capacity)) return GCP_ERR_WORKSPACE;
  // count + prefix sum (as gcp_bin_tiles_count, without handing the total to the host); the 64-bit total sits in the
  // last 256 bytes of the workspace, which the sort buffers never reach (gcp_bin_workspace_bytes adds them)
  unsigned long long* total64 = (unsigned long long*)((char*)ws + ((ws_bytes - 256) & ~(size_t)255));
  int* cnt = (int*)((char*)ws + 256);
  int* sws = (int*)((char*)ws + 256 + align256((size_t)n_gauss * sizeof(int)));
  GCP_HIP(hipMemsetAsync(total64, 0, sizeof(unsigned long long), stream));
  const i64 count_blocks = (n_gauss + 255) / 256 < 512 ? (n_gauss + 255) / 256 : 512;
  hipLaunchKernelGGL(k_tile_count, dim3((unsigned)count_blocks), dim3(256), 0, stream, start_xy, end_xy, (i64)n_gauss, width,
                     height, cnt, total64);
  GCP_HIP(hipGetLastError());
  const int st = launch_excl_scan(cnt, tile_off, n_gauss, sws, stream);
  if (st != GCP_OK) return st;
  // (a total beyond int32 wraps tile_off negative, which the emit kernel reads as "does not fit": flagged as overflow)
  return bin_fill(start_xy, end_xy, n_gauss, width, height, tile_off, capacity, tile_start, tile_list, info, total64, ws, stream);
}

static int make_args(BlendArgs& a, const int32_t* start_xy, const int32_t* end_xy, const float* mean_xy,
                     const float* vinv, const float* opacity, const float* l_d, int32_t width, int32_t height,
                     const int32_t* tile_start, const int32_t* tile_list) {
  if (width < 0 || height < 0 || !tile_start) return GCP_ERR_INVALID_ARGUMENT;
  a.start = start_xy; a.end = end_xy; a.mean = mean_xy; a.vinv = vinv; a.opacity = opacity; a.l_d = l_d;
  a.tile_start = tile_start; a.tile_list = (const unsigned*)tile_list;
  a.W = width; a.H = height; a.tiles_x = tile_grid(width, height).tx;
  return GCP_OK;
}

size_t gcp_blend_checkpoint_floats(int64_t n_tile_pairs, int32_t width, int32_t height) {
  if (width < 0 || height < 0) return 0;
  const TileGrid tg = tile_grid(width, height);
  return ckpt_floats((i64)n_tile_pairs, tg.tx * tg.ty);
}

int gcp_blend_forward(const int32_t* start_xy, const int32_t* end_xy, const float* mean_xy, const float* vinv,
                      const float* opacity, const float* l_d, int64_t n_gauss, int32_t width, int32_t height,
                      const int32_t* tile_start, const int32_t* tile_list, float* image, float* t_ckpt, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlendArgs a;
  const int st = make_args(a, start_xy, end_xy, mean_xy, vinv, opacity, l_d, width, height, tile_start, tile_list);
  if (st != GCP_OK || !image || n_gauss < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss > 0 && (!start_xy || !end_xy || !mean_xy || !vinv || !opacity || !l_d)) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  if (t_ckpt) hipLaunchKernelGGL((k_blend_fwd<true>), dim3((unsigned)(tg.tx * tg.ty)), dim3(256), 0, stream, a, image, t_ckpt);
  else hipLaunchKernelGGL((k_blend_fwd<false>), dim3((unsigned)(tg.tx * tg.ty)), dim3(256), 0, stream, a, image, (float*)nullptr);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

size_t gcp_blend_backward_workspace_bytes(int64_t n_tile_pairs) {
  // per-entry partial sums
  return align256((size_t)(n_tile_pairs > 0 ? n_tile_pairs : 1) * kGradVals * sizeof(float));
}

int gcp_blend_backward(const int32_t* start_xy, const int32_t* end_xy, const float* mean_xy, const float* vinv,
                       const float* opacity, const float* l_d, int64_t n_gauss, int32_t width, int32_t height,
                       const int32_t* tile_off, int64_t n_tile_pairs, const int32_t* tile_start,
                       const int32_t* tile_list, const float* t_ckpt, const float* grad_image, float* grad_mean,
                       float* grad_vinv, float* grad_opacity, float* grad_l, void* ws, size_t ws_bytes,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlendArgs a;
  const int st = make_args(a, start_xy, end_xy, mean_xy, vinv, opacity, l_d, width, height, tile_start, tile_list);
  if (st != GCP_OK || n_gauss < 0 || n_tile_pairs < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss == 0) return GCP_OK;
  if (!start_xy || !end_xy || !mean_xy || !vinv || !opacity || !l_d || !tile_off || !t_ckpt || !grad_image ||
      !grad_mean || !grad_vinv || !grad_opacity || !grad_l || !ws)
    return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_blend_backward_workspace_bytes(n_tile_pairs)) return GCP_ERR_WORKSPACE;
  const TileGrid tg = tile_grid(width, height);
  float* partial = (float*)ws;
  if (n_tile_pairs > 0) {
    hipLaunchKernelGGL(k_blend_bwd, dim3((unsigned)(tg.tx * tg.ty)), dim3(256), 0, stream, a, tile_off, t_ckpt,
                       grad_image, partial);
    GCP_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_grad_reduce, dim3((unsigned)((n_gauss + 255) / 256)), dim3(256), 0, stream,
                     (const float*)partial, tile_off, tile_start, tg.tx * tg.ty, vinv, (i64)n_gauss, (i64)n_tile_pairs, grad_mean,
                     grad_vinv, grad_opacity, grad_l);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_pixel_lists_count(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                          int32_t height, const int32_t* tile_start, const int32_t* tile_list,
                          int32_t* pixel_count, int32_t* box_size, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlendArgs a;
  const int st = make_args(a, start_xy, end_xy, nullptr, nullptr, nullptr, nullptr, width, height, tile_start, tile_list);
  if (st != GCP_OK || !pixel_count || !box_size || n_gauss < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss > 0 && (!start_xy || !end_xy)) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  hipLaunchKernelGGL((k_pixel_lists<false>), dim3((unsigned)(tg.tx * tg.ty)), dim3(256), 0, stream, a, pixel_count,
                     (const int*)nullptr, (const int*)nullptr, (int*)nullptr, (int*)nullptr, (int*)nullptr);
  GCP_HIP(hipGetLastError());
  if (n_gauss > 0) {
    hipLaunchKernelGGL(k_box_sizes, dim3((unsigned)((n_gauss + 255) / 256)), dim3(256), 0, stream, start_xy, end_xy,
                       (i64)n_gauss, width, height, box_size);
    GCP_HIP(hipGetLastError());
  }
  return GCP_OK;
}

int gcp_gather_f32(const float* src, const int32_t* index, float* dst, int64_t n, void* stream_) {
  if (n < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!src || !index || !dst) return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_gather_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, src, index, dst, (i64)n);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_unsort_finish(const float* inclusive, const float* sorted_x, const int32_t* index, float* full, uint8_t* keep,
                      int64_t n, int32_t mode, void* stream_) {
  if (n < 0 || (mode != 0 && mode != 1)) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!inclusive || !sorted_x || !index || !full || !keep) return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_unsort_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, inclusive, sorted_x,
                     index, full, keep, (i64)n, mode);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

size_t gcp_sort_workspace_bytes(int64_t n) {
  const int64_t m = n > 0 ? n : 1;
  const int64_t nblk = (m + kSortChunk - 1) / kSortChunk;
  size_t b = 2 * align256((size_t)m * sizeof(unsigned));            // ping-pong key / payload
  b += 2 * align256((size_t)(256 * nblk + 1) * sizeof(int));        // hist, hist_excl
  b += gcp_scan_i32_workspace_bytes(256 * nblk);
  return b;
}

static int sort_impl(const unsigned* keys_in, bool rects, int64_t n, int32_t key_bits, int32_t id_width, uint32_t* keys_out,
                     int32_t* index_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 0 || n > 0x7fffffffLL || key_bits < 1 || key_bits > 32 || id_width < 0 || (id_width && (!rects || key_bits > 24)))
    return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!keys_in || !keys_out || !index_out || !ws) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_sort_workspace_bytes(n)) return GCP_ERR_WORKSPACE;
  static const int xcd_remap = [] { const char* e = getenv("GCP_SORT_XCD"); return (e && *e) ? atoi(e) : 1; }();
  const i64 nblk = (n + kBigChunk - 1) / kBigChunk;
  // chunks per block: as many as kSortSub, but keep a few thousand blocks for the 1024 block slots of the chip
  int sub = (int)(nblk / 2048);
  sub = sub < 1 ? 1 : (sub > kSortSub ? kSortSub : sub);
  const i64 nsuper = (nblk + sub - 1) / sub;
  const dim3 grid(sort_grid(nsuper, xcd_remap)), block(256);
  char* p = (char*)ws;
  unsigned* keyY = (unsigned*)p; p += align256((size_t)n * sizeof(unsigned));
  unsigned* valY = (unsigned*)p; p += align256((size_t)n * sizeof(unsigned));
  int* hist = (int*)p; p += align256((size_t)(256 * nblk + 1) * sizeof(int));
  int* hist_ex = (int*)p; p += align256((size_t)(256 * nblk + 1) * sizeof(int));
  int* sws = (int*)p;
  unsigned* keyX = keys_out;
  unsigned* valX = (unsigned*)index_out;
  // passes of at most 8 bits, all of the same width: 24 bits -> 3 x 8, 21 bits -> 3 x 7, 25 bits -> 4 x 7
  const int passes = (key_bits + 7) / 8;
  const int width = (key_bits + passes - 1) / passes;
  const unsigned* ks = keys_in;
  const unsigned* vs = nullptr;
  for (int pass = 0; pass < passes; ++pass) {
    const bool to_x = ((passes - 1 - pass) & 1) == 0;  // the last pass lands in the caller's buffers
    unsigned* kd = to_x ? keyX : keyY;
    unsigned* vd = to_x ? valX : valY;
    SortPass ps;
    ps.shift = width * pass;
    ps.dmask = (1u << width) - 1u;
    ps.idw = id_width;
    ps.restore = (id_width && pass == passes - 1) ? 1 : 0;
    ps.inv_idw = id_width ? 1.0f / (float)id_width : 0.0f;
    const bool src_rects = pass == 0 && rects;
    const int vec = (((uintptr_t)ks & 15u) == 0) ? 1 : 0;  // (the ping-pong buffers always are; a caller's view may not be)
    if (src_rects) hipLaunchKernelGGL((k_sort_hist2<true>), grid, block, 0, stream, ks, (i64)n, ps, hist, (int)nsuper, sub, xcd_remap, vec);
    else hipLaunchKernelGGL((k_sort_hist2<false>), grid, block, 0, stream, ks, (i64)n, ps, hist, (int)nsuper, sub, xcd_remap, vec);
    GCP_HIP(hipGetLastError());
    const int st = launch_excl_scan(hist, hist_ex, 256 * nsuper, sws, stream);
    if (st != GCP_OK) return st;
    const dim3 sblock(64 * kSortWaves);
    if (src_rects)
      hipLaunchKernelGGL((k_sort_scatter2<true, true>), grid, sblock, 0, stream, ks, vs, kd, vd, (i64)n, ps, (const int*)hist_ex,
                         (int)nsuper, sub, xcd_remap);
    else if (pass == 0)
      hipLaunchKernelGGL((k_sort_scatter2<true, false>), grid, sblock, 0, stream, ks, vs, kd, vd, (i64)n, ps, (const int*)hist_ex,
                         (int)nsuper, sub, xcd_remap);
    else
      hipLaunchKernelGGL((k_sort_scatter2<false, false>), grid, sblock, 0, stream, ks, vs, kd, vd, (i64)n, ps, (const int*)hist_ex,
                         (int)nsuper, sub, xcd_remap);
    GCP_HIP(hipGetLastError());
    ks = kd;
    vs = vd;
  }
  return GCP_OK;
}

int gcp_sort_pairs_u32(const uint32_t* keys_in, int64_t n, int32_t key_bits, uint32_t* keys_out, int32_t* index_out,
                       void* ws, size_t ws_bytes, void* stream_) {
  return sort_impl(keys_in, false, n, key_bits, 0, keys_out, index_out, ws, ws_bytes, (hipStream_t)stream_);
}

int gcp_sort_rects(const int32_t* rects_xy, int64_t n, int32_t key_bits, int32_t id_width, uint32_t* keys_out, int32_t* index_out,
                   void* ws, size_t ws_bytes, void* stream_) {
  return sort_impl((const unsigned*)rects_xy, true, n, key_bits, id_width, keys_out, index_out, ws, ws_bytes, (hipStream_t)stream_);
}

int gcp_rects_key_range(const int32_t* rects_xy, int64_t n, int32_t* out_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || !out_dev) return GCP_ERR_INVALID_ARGUMENT;
  // (device-side fills: an asynchronous copy from this function's stack could outlive it)
  GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)out_dev, 0, 1, stream));
  GCP_HIP(hipMemsetD32Async((hipDeviceptr_t)(out_dev + 1), 0x7fffffff, 1, stream));
  if (n == 0) return GCP_OK;
  if (!rects_xy) return GCP_ERR_INVALID_ARGUMENT;
  i64 blocks = (n + 4095) / 4096;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_rects_key_range, dim3((unsigned)blocks), dim3(256), 0, stream, rects_xy, (i64)n, out_dev);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

size_t gcp_compact_workspace_bytes(int64_t n) {
  const int64_t nb = ((n > 0 ? n : 1) + kCompactTile - 1) / kCompactTile + 1;
  return align256((size_t)(nb + 1) * sizeof(int)) * 2 + gcp_scan_i32_workspace_bytes(nb);
}

// per-tile kept counts from the counts of dropped elements gcp_pairs_scan_boxes took while writing the array
static __global__ void k_counts_from_dropped(const int* __restrict__ dropped, i64 n, i64 nb, int* __restrict__ cnt) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nb) return;
  const i64 len = n - t * kCompactTile;
  cnt[t] = (int)(len < kCompactTile ? len : kCompactTile) - dropped[t];
}

int gcp_compact_finish(const float* inclusive, const float* self, int64_t begin, int64_t end, int32_t mode, float* values,
                       uint8_t* keep, int32_t* count_dev, const int32_t* dropped_per_tile, void* ws, size_t ws_bytes,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (begin < 0 || end < begin || end - begin > 0x7fffffffLL || (mode != 0 && mode != 1) || !count_dev) return GCP_ERR_INVALID_ARGUMENT;
  if (dropped_per_tile && begin % kCompactTile != 0) return GCP_ERR_INVALID_ARGUMENT;
  const i64 n = end - begin;
  if (n == 0) {
    GCP_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), stream));
    return GCP_OK;
  }
  if (!inclusive || !self || !values || !keep || !ws) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_compact_workspace_bytes(n)) return GCP_ERR_WORKSPACE;
  const i64 nb = (n + kCompactTile - 1) / kCompactTile;
  char* p = (char*)ws;
  int* cnt = (int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  int* off = (int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  int* sws = (int*)p;
  const bool vec = (((uintptr_t)(inclusive + begin) | (uintptr_t)(self + begin) | (uintptr_t)keep) & 15u) == 0;
  // (a last tile cut short by `end` gets a count that covers elements beyond it: never used — ranks come from the
  // exclusive scan of the tiles before, the total from the write pass)
  if (dropped_per_tile)
    hipLaunchKernelGGL(k_counts_from_dropped, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream,
                       dropped_per_tile + begin / kCompactTile, n, nb, cnt);
  else if (vec) hipLaunchKernelGGL((k_compact<true, false>), dim3((unsigned)nb), dim3(256), 0, stream, inclusive + begin, self + begin, n, mode, cnt,
                                   (const int*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (int*)nullptr);
  else hipLaunchKernelGGL((k_compact<false, false>), dim3((unsigned)nb), dim3(256), 0, stream, inclusive + begin, self + begin, n, mode, cnt,
                          (const int*)nullptr, (float*)nullptr, (unsigned char*)nullptr, (int*)nullptr);
  GCP_HIP(hipGetLastError());
  const int st = launch_excl_scan(cnt, off, nb, sws, stream);
  if (st != GCP_OK) return st;
  if (vec) hipLaunchKernelGGL((k_compact<true, true>), dim3((unsigned)nb), dim3(256), 0, stream, inclusive + begin, self + begin, n, mode,
                              (int*)nullptr, (const int*)off, values, keep, count_dev);
  else hipLaunchKernelGGL((k_compact<false, true>), dim3((unsigned)nb), dim3(256), 0, stream, inclusive + begin, self + begin, n, mode,
                          (int*)nullptr, (const int*)off, values, keep, count_dev);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

// keep != NULL: the FINAL form (values + keep mask + zero counts); else the inclusive form, counting when `dropped_per_tile`
static int walk_impl(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                     const int32_t* tile_start, const int32_t* tile_list, const int32_t* box_off, const float* x, float* out,
                     int64_t n_pairs, int32_t mode, int32_t* dropped_per_tile, uint8_t* keep, void* stream_, bool prepared = false) {
  hipStream_t stream = (hipStream_t)stream_;
  BlendArgs a;
  const int st = make_args(a, start_xy, end_xy, nullptr, nullptr, nullptr, nullptr, width, height, tile_start, tile_list);
  if (st != GCP_OK || n_gauss < 0 || mode < 0 || mode > 2 || n_pairs < 0 || n_pairs > 0x7fffffffLL) return GCP_ERR_INVALID_ARGUMENT;
  if (keep && !dropped_per_tile && n_pairs > 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_pairs == 0) return GCP_OK;
  if (keep && !prepared) GCP_HIP(hipMemsetAsync(keep, 1, (size_t)n_pairs, stream));  // every pair kept until the walk finds its inclusive value 0
  if (dropped_per_tile && !prepared)
    GCP_HIP(hipMemsetAsync(dropped_per_tile, 0, (size_t)((n_pairs + kCompactTile - 1) / kCompactTile) * sizeof(int), stream));
  if (n_gauss == 0) return GCP_OK;
  if (!start_xy || !end_xy || !tile_list || !box_off || !x || !out || x == out) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  // stripes of one tile row dealt round-robin to the XCDs (walk_tile): as fast as contiguous bands on a scene that fills the
  // image evenly (0.65 against 0.64 ms at cfg3), and 18 % / 43 % faster where the Gaussians crowd the middle (sigma = extent / 4,
  // / 8: the bands of the crowded region worked while the others idled) — profiles/r04_walk_experiments.md
  static const int xcd_remap = [] { const char* e = getenv("GCP_WALK_XCD"); return (e && *e) ? atoi(e) : 2; }();
  // pair positions as 32-bit byte offsets while the list is no longer than 2^30 pairs (GCP_WALK_WIDE=1 forces the other form)
  const char* fw = getenv("GCP_WALK_WIDE");  // read per call: the tests switch it inside one process
  const bool wide = (fw && *fw && atoi(fw) != 0) || n_pairs > (1LL << 30);
  // a pair's position is formed with ONE 24-bit multiply (row in the tile) x (box width, in bytes unless WIDE): a box can be
  // as wide as the image, so the image has to fit — 2^22 columns in the byte-offset form, 2^24 in the element form
  if ((int64_t)width + 1 >= (wide ? (1LL << 24) : (1LL << 22))) return GCP_ERR_INVALID_ARGUMENT;
  const int out_mode = keep ? kWalkFinal : (dropped_per_tile ? kWalkCount : kWalkInclusive);
  const int n_tiles = tg.tx * tg.ty;
  const dim3 grid(walk_grid(n_tiles, tg.tx, xcd_remap)), block(kWalkThreads);
#if GCP_WALK_DENSE
  const int n_blocks = (n_tiles + 3) / 4;
  const dim3 dgrid(sort_grid(n_blocks, xcd_remap));
#define GCP_WALK(M, W_, O_) \
  hipLaunchKernelGGL((k_pairs_walk_dense<M, W_, O_>), dgrid, dim3(256), 0, stream, a, box_off, x, out, dropped_per_tile, keep, n_tiles, n_blocks, xcd_remap)
#else
#define GCP_WALK(M, W_, O_) \
  hipLaunchKernelGGL((k_pairs_scan_boxes<M, W_, O_>), grid, block, 0, stream, a, box_off, x, out, dropped_per_tile, keep, n_tiles, xcd_remap)
#endif
#define GCP_WALK_OUT(M, W_)                                  \
  do {                                                       \
    if (out_mode == kWalkFinal) GCP_WALK(M, W_, kWalkFinal); \
    else if (out_mode == kWalkCount) GCP_WALK(M, W_, kWalkCount); \
    else GCP_WALK(M, W_, kWalkInclusive);                    \
  } while (0)
#define GCP_WALK_MODE(M) do { if (wide) GCP_WALK_OUT(M, true); else GCP_WALK_OUT(M, false); } while (0)
  if (mode == 0) GCP_WALK_MODE(0);
  else if (mode == 1) GCP_WALK_MODE(1);
  else GCP_WALK_MODE(2);
#undef GCP_WALK_MODE
#undef GCP_WALK_OUT
#undef GCP_WALK
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_pairs_scan_boxes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                         const int32_t* tile_start, const int32_t* tile_list, const int32_t* box_off, const float* x,
                         float* inclusive, int64_t n_pairs, int32_t mode, int32_t* dropped_per_tile, void* stream) {
  if (n_gauss == 0) return (n_gauss < 0 || n_pairs < 0 || mode < 0 || mode > 2 || width < 0 || height < 0 || !tile_start) ? GCP_ERR_INVALID_ARGUMENT : GCP_OK;
  return walk_impl(start_xy, end_xy, n_gauss, width, height, tile_start, tile_list, box_off, x, inclusive, n_pairs, mode,
                   dropped_per_tile, nullptr, stream);
}

int gcp_pairs_finish_prepare(uint8_t* keep, int32_t* dropped_per_tile, int64_t n_pairs, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_pairs < 0 || n_pairs > 0x7fffffffLL) return GCP_ERR_INVALID_ARGUMENT;
  if (n_pairs == 0) return GCP_OK;
  if (!keep || !dropped_per_tile) return GCP_ERR_INVALID_ARGUMENT;
  GCP_HIP(hipMemsetAsync(keep, 1, (size_t)n_pairs, stream));
  GCP_HIP(hipMemsetAsync(dropped_per_tile, 0, (size_t)((n_pairs + kCompactTile - 1) / kCompactTile) * sizeof(int), stream));
  return GCP_OK;
}

int gcp_pairs_finish_boxes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                           const int32_t* tile_start, const int32_t* tile_list, const int32_t* box_off, const float* x,
                           float* values, uint8_t* keep, int64_t n_pairs, int32_t mode, int32_t* dropped_per_tile, int32_t prepared,
                           void* stream) {
  if (n_pairs > 0 && (!keep || !dropped_per_tile)) return GCP_ERR_INVALID_ARGUMENT;
  return walk_impl(start_xy, end_xy, n_gauss, width, height, tile_start, tile_list, box_off, x, values, n_pairs, mode,
                   dropped_per_tile, keep, stream, prepared != 0);
}

size_t gcp_compact_kept_workspace_bytes(int64_t n) { return gcp_compact_workspace_bytes(n); }

// One launch, one block: the per-tile kept counts from what the walk dropped (or, from_dropped = false, the counts as
// k_count_keep left them) and their TOTAL — all the read-back that sizes the result needs; the prefix sums are only made
// when something has to be moved (gcp_compact_kept_write).
static __global__ __launch_bounds__(1024) void k_kept_total(const int* __restrict__ dropped, i64 n, i64 nb, int* __restrict__ cnt,
                                                            int* __restrict__ count_dev, bool from_dropped) {
  __shared__ int s_w[16];
  int s = 0;
  // eight loads in flight per thread (one at a time, the 40 rounds of a 1.65e8-pair list took 27 us: a round trip each)
  for (i64 t0 = threadIdx.x; t0 < nb; t0 += 8 * 1024) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const i64 t = t0 + u * 1024;
      v[u] = t < nb ? (from_dropped ? dropped[t] : cnt[t]) : 0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const i64 t = t0 + u * 1024;
      if (t >= nb) continue;
      int c = v[u];
      if (from_dropped) {
        const i64 len = n - t * kCompactTile;
        c = (int)(len < kCompactTile ? len : kCompactTile) - c;
        cnt[t] = c;
      }
      s += c;
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int j = 0; j < 16; ++j) tot += s_w[j];
    count_dev[0] = tot;
  }
}

int gcp_compact_kept_count(const uint8_t* keep, const int32_t* dropped_per_tile, int64_t n_total, int64_t begin, int64_t end,
                           int32_t* count_dev, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (begin < 0 || end < begin || end > n_total || end - begin > 0x7fffffffLL || !count_dev) return GCP_ERR_INVALID_ARGUMENT;
  const i64 n = end - begin;
  if (n == 0) {
    GCP_HIP(hipMemsetAsync(count_dev, 0, sizeof(int), stream));
    return GCP_OK;
  }
  if (!keep || !ws) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_compact_kept_workspace_bytes(n)) return GCP_ERR_WORKSPACE;
  const i64 nb = (n + kCompactTile - 1) / kCompactTile;
  char* p = (char*)ws;
  int* cnt = (int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  int* off = (int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  int* sws = (int*)p;
  // the walk's own counts serve when the range's tiles are the array's tiles and its last tile is not cut short by `end`
  (void)off; (void)sws;  // (the prefix sums are gcp_compact_kept_write's)
  if (dropped_per_tile && begin % kCompactTile == 0 && (end == n_total || end % kCompactTile == 0)) {
    hipLaunchKernelGGL(k_kept_total, dim3(1), dim3(1024), 0, stream, dropped_per_tile + begin / kCompactTile, n, nb, cnt, count_dev, true);
  } else {
    hipLaunchKernelGGL(k_count_keep, dim3((unsigned)nb), dim3(256), 0, stream, keep + begin, n, cnt);
    hipLaunchKernelGGL(k_kept_total, dim3(1), dim3(1024), 0, stream, (const int*)nullptr, n, nb, cnt, count_dev, false);
  }
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_compact_kept_write(const float* values_in, const uint8_t* keep, int64_t begin, int64_t end, float* values_out, void* ws,
                           size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (begin < 0 || end < begin || end - begin > 0x7fffffffLL) return GCP_ERR_INVALID_ARGUMENT;
  const i64 n = end - begin;
  if (n == 0) return GCP_OK;
  if (!values_in || !keep || !values_out || !ws) return GCP_ERR_INVALID_ARGUMENT;
  if (ws_bytes < gcp_compact_kept_workspace_bytes(n)) return GCP_ERR_WORKSPACE;
  const i64 nb = (n + kCompactTile - 1) / kCompactTile;
  // ranks of the tiles from the counts gcp_compact_kept_count left in ws
  char* p = (char*)ws;
  const int* cnt = (const int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  int* off = (int*)p; p += align256((size_t)(nb + 1) * sizeof(int));
  const int st = launch_excl_scan(cnt, off, nb, (int*)p, stream);
  if (st != GCP_OK) return st;
  const bool vec = (((uintptr_t)(values_in + begin)) & 15u) == 0 && (((uintptr_t)(keep + begin)) & 3u) == 0;
  if (vec) hipLaunchKernelGGL((k_compact_kept<true>), dim3((unsigned)nb), dim3(256), 0, stream, values_in + begin, keep + begin, n, off, values_out);
  else hipLaunchKernelGGL((k_compact_kept<false>), dim3((unsigned)nb), dim3(256), 0, stream, values_in + begin, keep + begin, n, off, values_out);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_box_sizes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                  int32_t* box_size, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_gauss < 0 || width < 0 || height < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss == 0) return GCP_OK;
  if (!start_xy || !end_xy || !box_size) return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_box_sizes, dim3((unsigned)((n_gauss + 255) / 256)), dim3(256), 0, stream, start_xy, end_xy,
                     (i64)n_gauss, width, height, box_size);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_expand_rects(const int32_t* start_xy, const int32_t* end_xy, const int32_t* box_off, int64_t n_gauss,
                     int64_t n_pairs, int32_t width, int32_t height, int32_t* rects_xy, int32_t* pair_gauss,
                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_gauss < 0 || n_pairs < 0 || width < 0 || height < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_pairs == 0) return GCP_OK;
  if (!start_xy || !end_xy || !box_off || !rects_xy || n_gauss == 0) return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_expand_rects, dim3((unsigned)((n_gauss + kExpandBoxes - 1) / kExpandBoxes)), dim3(256), 0, stream, start_xy, end_xy,
                     box_off, (i64)n_gauss, (i64)n_pairs, width, height, (int2*)rects_xy, pair_gauss);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_pixel_lists_fill(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                         int32_t height, const int32_t* tile_start, const int32_t* tile_list,
                         const int32_t* pixel_off, const int32_t* box_off, int32_t* pair_gauss,
                         int32_t* pair_index, int32_t* pair_key, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BlendArgs a;
  const int st = make_args(a, start_xy, end_xy, nullptr, nullptr, nullptr, nullptr, width, height, tile_start, tile_list);
  if (st != GCP_OK || n_gauss < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss == 0) return GCP_OK;
  if (!start_xy || !end_xy || !pixel_off || !box_off || !pair_gauss || !pair_index) return GCP_ERR_INVALID_ARGUMENT;
  const TileGrid tg = tile_grid(width, height);
  hipLaunchKernelGGL((k_pixel_lists<true>), dim3((unsigned)(tg.tx * tg.ty)), dim3(256), 0, stream, a, (int*)nullptr,
                     pixel_off, box_off, pair_gauss, pair_index, pair_key);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

}  // extern "C"
