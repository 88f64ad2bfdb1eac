// gcp_scan.hip — segmented (grouped) inclusive scans for MI355X (gfx950, wave64).
//
// Implements the three ops the reference registers as module `grouped_cumprod`
// (reference: cuda_kernel/cuda_kernel.cpp:5-22) plus a suffix-sum variant:
//   a1 grouped_cumprod_forward   cuda_kernel/grouped_cumprod_forward.cu:6-24
//   a2 grouped_cumsum_forward    cuda_kernel/grouped_cumsum_forward.cu:6-24
//   a3 grouped_cumprod_backward  cuda_kernel/grouped_cumprod_backward.cu:9-65
// The reference delegates a1/a2 to thrust::inclusive_scan_by_key and runs a3 as
// an O(sum L^2) per-element loop.  This file is a from-scratch CDNA4 design:
//
//   * One 256-thread block (4 waves) owns one TILE of 1024*ROWS consecutive
//     elements.  Every lane loads 16 B vectors (4 consecutive elements), so a
//     wave instruction moves 1 KiB contiguous and a block row 4 KiB.
//   * Head flags come from comparing adjacent keys.  Each lane scans its 4
//     items serially, lane aggregates are scanned across the wave with a
//     flag-free segmented Kogge-Stone in DPP (row_shr 1/2/4/8, row_bcast15,
//     row_bcast31): the nearest head lane is derived from one ballot, so only
//     values travel between lanes.  Rows chain through a wave-uniform carry,
//     waves through 4 LDS words and ONE barrier.
//   * No inter-block communication on the common path: the carry entering a
//     tile is recomputed by wave 0 from the raw inputs with a bounded look-back
//     (LB_CHUNKS x 256 elements = one tile; those bytes are L2/MALL-resident
//     because the neighbouring tile is being streamed at the same time).
//   * Groups that reach further back than that window (pixel lists > 4096 deep)
//     continue IN THE SAME PASS on per-tile descriptors: every tile publishes
//     {aggregate of its trailing group, open/closed, first head offset} as one
//     64-bit word (agent-scope store) as soon as its local scan is done; a tile
//     whose raw window is exhausted walks the descriptors of the tiles before it,
//     64 per step, multiplying the aggregates of head-less ("open") tiles until it
//     meets a tile with a head or with an already resolved prefix.  Waiting for a
//     descriptor that is not published yet is BOUNDED (wall clock): a tile that
//     runs out of patience marks itself unresolved and ONE follow-up kernel
//     (<= 256 blocks, each owning a contiguous range of tiles; a no-op when nothing
//     is unresolved) folds the missing prefix into its leading elements.  So the
//     protocol needs no forward-progress or dispatch-order assumption, cannot
//     hang, and no value-carrying atomic decides a result: deterministic.
//     Descriptors live in two sets used alternately (parity of a device-side
//     launch counter); the follow-up kernel of one launch clears the set the next
//     launch will use, so a stale word is never taken for a published one.  The
//     workspace must be zeroed once (gcp_workspace_init).
//   * The backward (a3) is the same machinery run in reverse index order on
//     w[i] = grad_out[i] * cumprod[i] with the division by p'_j fused into the
//     store: one O(n) pass, 20 B / element.
//
// HBM-bound by construction (12 B or 20 B per element, ~10 VALU per element):
// no MFMA.  All index arithmetic on the array is 64-bit.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>

#include "gcp_device.hpp"
#include "grouped_cumprod_hip.h"

namespace gcp {
thread_local int t_last_hip_error = 0;
int hip_fail(hipError_t e) {
  t_last_hip_error = (int)e;
  return GCP_ERR_HIP;
}
}  // namespace gcp

namespace {
using namespace gcp;

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef int int4_t __attribute__((ext_vector_type(4)));

enum : int { M_CUMPROD_FWD = 0, M_CUMSUM_FWD = 1, M_CUMPROD_BWD = 2, M_CUMSUM_REV = 3 };

constexpr int kThreads = 256;      // 4 waves
constexpr int kWaves = 4;
#ifndef GCP_ROWS
#define GCP_ROWS 4
#endif
#ifndef GCP_XCD_REMAP_DEFAULT
#define GCP_XCD_REMAP_DEFAULT 1
#endif
constexpr int kRows = GCP_ROWS;    // 16-byte vectors per lane per array
constexpr int kTile = 1024 * kRows;
#ifndef GCP_LB_BATCH
#define GCP_LB_BATCH 5
#endif
constexpr int kLbBatch = GCP_LB_BATCH;  // look-back chunks fetched per dependent round trip after the first
constexpr int kLbChunks = ((kTile / 256 - 1) / kLbBatch) * kLbBatch + 1;  // window in 256-element chunks (16 = one tile at kRows 4)
constexpr int kFixBlocks = 256;     // upper bound of the fallback kernel's grid (two-pass mode)
constexpr int kFixBlocksQuiet = 32; // ... when the descriptor tree is on and the kernel normally finds nothing to do
constexpr int kWsHeaderBytes = 256;
// workspace header words
constexpr int kHdrEpoch = 0;        // launches completed on this workspace; its parity selects the descriptor set
constexpr int kHdrDone = 1;         // fallback blocks finished (the last one advances the epoch)
constexpr int kHdrUnresolved = 2;   // tiles the fallback kernel fixed up in the last launch (introspection)
constexpr int kHdrTiles = 3;        // [3], [4]: tiles written in descriptor set 0 / 1 by its last user
constexpr int kHdrDescResolved = 5; // tiles that took their carry from the block tree in the last launch (introspection;
                                    // counted by the follow-up kernel from a flag in the level-0 descriptors: one atomic per
                                    // tile inside the main kernel serialises at the L2 and cost 1 ms per 40 000 tiles)
constexpr int kHdrAnyUnresolved = 6; // set (plain store) by any tile that gave up waiting; read and cleared by the follow-up kernel
constexpr int kHdrBarrier = 7;      // arrival counter of the follow-up kernel's grid barriers (rare path only)
// descriptor = {aggregate bits (low word), flags (high word)}
constexpr unsigned kDOpen = 1u;        // the aggregate is relative to the tile's (still unknown) carry-in: no head in the tile
constexpr unsigned kDUnresolved = 2u;  // the tile's leading elements still lack their carry (the fallback kernel's work list)
constexpr unsigned kDValid = 1u << 15; // published in this launch (the set was cleared before)
constexpr unsigned kDTree = 1u << 16;  // the tile took its carry from the block tree (introspection; level 0 only)
#ifndef GCP_XCD_CHUNK
#define GCP_XCD_CHUNK 32
#endif
constexpr int kLevels = 4;          // radix-64 block tree over the tiles: 64^4 tiles = 6.9e10 elements
constexpr int kXcdChunk = GCP_XCD_CHUNK;  // consecutive tiles one XCD takes before the next XCD's run starts
#ifndef GCP_DESC_WAIT_US
#define GCP_DESC_WAIT_US 200
#endif
#ifndef GCP_LB_EARLY_EXIT
#define GCP_LB_EARLY_EXIT 1      // leave the raw look-back after its first chunk when wave 0's whole share is head-less (§3.1 of DESIGN.md)
#endif

static_assert(kLbChunks * 256 == kTile, "the raw look-back window is exactly the previous tile (the descriptor walk starts at the tile before it)");
static_assert((kLbChunks - 1) % kLbBatch == 0, "chunks after the first are fetched kLbBatch at a time");

template <int MODE>
struct Mode {
  static constexpr bool kMul = (MODE == M_CUMPROD_FWD);
  static constexpr bool kRev = (MODE == M_CUMPROD_BWD || MODE == M_CUMSUM_REV);
  static constexpr bool kBwd = (MODE == M_CUMPROD_BWD);
};

template <bool MUL>
struct Monoid {
  static __device__ __forceinline__ float identity() { return MUL ? 1.0f : 0.0f; }
  static __device__ __forceinline__ float op(float a, float b) { return MUL ? a * b : a + b; }
};

struct ScanArgs {
  const float* in0;  // x (a1/a2) or param (a3)
  const float* in1;  // param_cumprod (a3 only)
  const float* in2;  // grad_out (a3 only)
  const int* key;    // pixel key (a1/a2) or dense group id `inv` (a3)
  const float* carry;  // optional per-group prefix (indexed by `key`, which must then be the dense group id)
  const int* index;    // INDEXED scans: element i of the scan is in0[index[i]] and its result goes to out[index[i]]
  float* out;
  i64 n;
  i64 ntiles;
  unsigned long long* desc_sets;  // two interleaved sets (word 2 e + s) of descriptor entries {aggregate bits, flags |
                                  // first_head << 2}; entry e = tile t at level 0, then the upper levels of the block tree
  i64 lvl_off[kLevels + 1];       // first entry of every level of the radix-64 block tree; [kLevels] = entries in all
  unsigned* hdr;      // workspace header (kHdr*)
  int xcd_remap;
  long long patience; // longest wait for a missing descriptor, in 100 MHz ticks; < 0: no descriptor walk at all
};

// Inclusive segmented scan of one value per lane.  `h` = nearest lane <= this
// one that starts a segment, -1 if none: lane l may absorb lane s iff s >= h.
template <bool MUL>
__device__ __forceinline__ float wave_seg_scan(float v, int h, int lane) {
  typedef Monoid<MUL> M;
  const float id = M::identity();
  float t;
  t = dpp_f<0x111, 0xf>(id, v); v = M::op(v, (lane - 1 >= h) ? t : id);
  t = dpp_f<0x112, 0xf>(id, v); v = M::op(v, (lane - 2 >= h) ? t : id);
  t = dpp_f<0x114, 0xf>(id, v); v = M::op(v, (lane - 4 >= h) ? t : id);
  t = dpp_f<0x118, 0xf>(id, v); v = M::op(v, (lane - 8 >= h) ? t : id);
  // lane 15 of rows 0/2 -> rows 1/3 ; source lane = (lane & ~15) - 1
  t = dpp_f<0x142, 0xa>(id, v); v = M::op(v, ((lane & 48) - 1 >= h) ? t : id);
  // lane 31 -> rows 2,3
  t = dpp_f<0x143, 0xc>(id, v); v = M::op(v, (31 >= h) ? t : id);
  return v;
}

// Plain wave reduction (result valid in lane 63, returned broadcast).
template <bool MUL>
__device__ __forceinline__ float wave_reduce(float v) {
  typedef Monoid<MUL> M;
  const float id = M::identity();
  v = M::op(v, dpp_f<0x111, 0xf>(id, v));
  v = M::op(v, dpp_f<0x112, 0xf>(id, v));
  v = M::op(v, dpp_f<0x114, 0xf>(id, v));
  v = M::op(v, dpp_f<0x118, 0xf>(id, v));
  v = M::op(v, dpp_f<0x142, 0xa>(id, v));
  v = M::op(v, dpp_f<0x143, 0xc>(id, v));
  return readlane_f(v, 63);
}

// ----------------------------------------------------------------------------
// Loads / stores.  ALIGNED: all array bases are 16-byte aligned (torch
// allocations always are); otherwise dword accesses.
// ----------------------------------------------------------------------------
// Cache policy of the streaming accesses (A/B on cfg3, one process, interleaved): non-temporal STORES
// +1.8 % (forward) / +3.9 % (backward).  Non-temporal LOADS of everything: -7 % with the look-back loads
// included, +1.7 % forward / -3.5 % backward with only the tile loads (and -15 % on cache-resident cfg2):
// the look-back re-reads the END of the neighbouring tile and wants it cached.  Mode 2 — non-temporal for
// all waves but the last one in scan order, plus `param` in the backward, which no look-back touches —
// is neutral on the forward and cfg2 and +1.3 % on the backward.
#ifndef GCP_NT_LOAD
#define GCP_NT_LOAD 2
#endif
#ifndef GCP_NT_STORE
#define GCP_NT_STORE 1
#endif
template <bool ALIGNED, bool NT = false>
__device__ __forceinline__ float4_t ld4(const float* p) {
  if (ALIGNED) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const float4_t*>(p));
    return *reinterpret_cast<const float4_t*>(p);
  }
  float4_t v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
  return v;
}
template <bool ALIGNED, bool NT = false>
__device__ __forceinline__ int4_t ld4(const int* p) {
  if (ALIGNED) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const int4_t*>(p));
    return *reinterpret_cast<const int4_t*>(p);
  }
  int4_t v; v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3];
  return v;
}
template <bool ALIGNED>
__device__ __forceinline__ void st4(float* p, float4_t v) {
  if (ALIGNED) {
    if (GCP_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<float4_t*>(p));
    else *reinterpret_cast<float4_t*>(p) = v;
    return;
  }
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}
__device__ __forceinline__ float4_t ld4_guard(const float* base, i64 p0, i64 n, float fill) {
  float4_t v;
  v.x = (p0 + 0 < n) ? base[p0 + 0] : fill;
  v.y = (p0 + 1 < n) ? base[p0 + 1] : fill;
  v.z = (p0 + 2 < n) ? base[p0 + 2] : fill;
  v.w = (p0 + 3 < n) ? base[p0 + 3] : fill;
  return v;
}
__device__ __forceinline__ int4_t ld4_guard(const int* base, i64 p0, i64 n, int fill) {
  int4_t v;
  v.x = (p0 + 0 < n) ? base[p0 + 0] : fill;
  v.y = (p0 + 1 < n) ? base[p0 + 1] : fill;
  v.z = (p0 + 2 < n) ? base[p0 + 2] : fill;
  v.w = (p0 + 3 < n) ? base[p0 + 3] : fill;
  return v;
}
// INDEXED scans (the sort -> scan -> un-sort sandwich of the reference's _create_alpha_brend, gs_model.py:548-555, in one
// pass): values are gathered through the sort permutation on the way in and scattered through it on the way out, so
// neither the sorted copy of the values nor the sorted result ever exists in memory.
template <bool ALIGNED, bool NT>
__device__ __forceinline__ float4_t ld4_indexed(const float* src, const int* index, i64 p, int4_t& idx) {
  idx = ld4<ALIGNED, NT>(index + p);
  float4_t v; v.x = src[idx.x]; v.y = src[idx.y]; v.z = src[idx.z]; v.w = src[idx.w];
  return v;
}
__device__ __forceinline__ float4_t ld4_indexed_guard(const float* src, const int* index, i64 p0, i64 n, float fill, int4_t& idx) {
  idx = ld4_guard(index, p0, n, 0);
  float4_t v;
  v.x = (p0 + 0 < n) ? src[idx.x] : fill;
  v.y = (p0 + 1 < n) ? src[idx.y] : fill;
  v.z = (p0 + 2 < n) ? src[idx.z] : fill;
  v.w = (p0 + 3 < n) ? src[idx.w] : fill;
  return v;
}
template <bool REV> __device__ __forceinline__ float4_t to_scan_order(float4_t v) {
  if (!REV) return v;
  float4_t r; r.x = v.w; r.y = v.z; r.z = v.y; r.w = v.x;
  return r;
}
template <bool REV> __device__ __forceinline__ int4_t to_scan_order(int4_t v) {
  if (!REV) return v;
  int4_t r; r.x = v.w; r.y = v.z; r.z = v.y; r.w = v.x;
  return r;
}

// Logical (scan-order) tile index of this block.  Blocks are dealt round-robin over the 8 XCDs; block b runs on XCD
// b & 7.  Tiles are handed out in groups of 8 * kXcdChunk: inside a group XCD x takes kXcdChunk CONSECUTIVE tiles
// (blocks x, x+8, x+16, ...), so a tile and its look-back source share an XCD's L2 for all but one tile in
// kXcdChunk, while the tile before any tile is at most 8 * kXcdChunk blocks away in dispatch order — resident at the
// same time, which is what keeps the descriptor look-back's waits short.  Performance only: nothing depends on where
// or when a block runs.
__device__ __forceinline__ i64 logical_tile(i64 b, i64 ntiles, int xcd_remap) {
  // xcd_remap = log2 of the run length + 1 (0: dispatch order); kXcdChunk = 32 for the plain scans.  The INDEXED scans
  // take longer runs (128 tiles): their gathers and scatters go to the pairs' ORIGINAL positions, where the pairs of
  // vertically neighbouring pixels — a few dozen tiles apart in sorted order — share cache lines.
  if (!xcd_remap) return b;
  const int lg = xcd_remap - 1 + 3;  // log2 of the group of 8 runs
  const i64 full = (ntiles >> lg) << lg;
  if (b >= full) return b;  // the last, partial group keeps dispatch order
  const i64 g = b >> lg, r = b & (((i64)1 << lg) - 1);
  return (g << lg) + ((r & 7) << (lg - 3)) + (r >> 3);
}

__device__ __forceinline__ unsigned long long pack_desc(float agg, unsigned flags, int first_head) {
  return ((unsigned long long)(flags | kDValid | ((unsigned)first_head << 2)) << 32) | __builtin_bit_cast(unsigned, agg);
}

// ----------------------------------------------------------------------------
// Main kernel: one tile per block.
// ----------------------------------------------------------------------------
// FIXUP: the follow-up kernel re-runs a tile whose wait for the descriptor tree ran out, with the carry `fix_carry` it
// took from the (completed) tree: same code, same association, so the tile gets the bits it would have got in time.
template <int MODE, bool ALIGNED, bool FULL, bool CARRY, bool FIXUP = false, bool INDEXED = false, bool INPLACE = false>
__device__ __forceinline__ void scan_tile(const ScanArgs& a, const i64 lt, float* s_wv, int* s_wf,
                                          float* s_tc, int* s_fh, const float fix_carry = 0.0f) {
  typedef Mode<MODE> MD;
  constexpr bool REV = MD::kRev;
  constexpr bool BWD = MD::kBwd;
  typedef Monoid<MD::kMul> M;
  const float id = M::identity();
  constexpr int WT = 256 * kRows;  // elements per wave

  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;  // logical wave == hardware wave
  const i64 n = a.n;
  const i64 pt = REV ? (a.ntiles - 1 - lt) : lt;
  const i64 base = pt * (i64)kTile;
  // descriptor set of this launch (parity of the workspace's launch counter); a scalar load issued first
  unsigned long long* const desc = a.desc_sets + (a.hdr[kHdrEpoch] & 1u);  // tile t at desc[2 t]

  // ---- issue all loads of this lane ------------------------------------
  // Streaming (non-temporal) loads are faster for bytes nobody reads again, but the END of a tile (scan
  // order) is what the next tile's look-back re-reads and must stay cached: GCP_NT_LOAD 0 = never,
  // 1 = all tile loads, 2 = all waves but the last one in scan order.
  const bool nt_main = (GCP_NT_LOAD == 1) || (GCP_NT_LOAD == 2 && w < kWaves - 1);
  float4_t v[kRows];
  int4_t kk[kRows];
  float4_t xp[BWD ? kRows : 1];
  int4_t ix[INDEXED ? kRows : 1];  // the permutation entries of this lane's elements, memory order
  static_assert(!(INDEXED && BWD), "the backward has no indexed form");
  int4_t lb_ix;                    // (look-back gathers: only the values are used)
  i64 p0[kRows];
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const int q = w * WT + r * 256 + lane * 4;
    p0[r] = REV ? (base + kTile - 4 - q) : (base + q);
    if (FULL) {
      if (nt_main) {  // wave-uniform
        kk[r] = ld4<ALIGNED, true>(a.key + p0[r]);
        if constexpr (BWD) {
          v[r] = ld4<ALIGNED, true>(a.in2 + p0[r]) * ld4<ALIGNED, true>(a.in1 + p0[r]);
        } else if constexpr (INDEXED) {
          v[r] = ld4_indexed<ALIGNED, true>(a.in0, a.index, p0[r], ix[r]);
        } else {
          v[r] = ld4<ALIGNED, true>(a.in0 + p0[r]);
        }
      } else {
        kk[r] = ld4<ALIGNED>(a.key + p0[r]);
        if constexpr (BWD) {
          v[r] = ld4<ALIGNED>(a.in2 + p0[r]) * ld4<ALIGNED>(a.in1 + p0[r]);
        } else if constexpr (INDEXED) {
          v[r] = ld4_indexed<ALIGNED, false>(a.in0, a.index, p0[r], ix[r]);
        } else {
          v[r] = ld4<ALIGNED>(a.in0 + p0[r]);
        }
      }
      if constexpr (BWD) xp[r] = ld4<ALIGNED, (GCP_NT_LOAD != 0)>(a.in0 + p0[r]);  // param is never re-read by a look-back
    } else {
      kk[r] = ld4_guard(a.key, p0[r], n, 0);
      if constexpr (BWD) {
        const float4_t g = ld4_guard(a.in2, p0[r], n, 0.0f);
        const float4_t c = ld4_guard(a.in1, p0[r], n, 0.0f);
        xp[r] = ld4_guard(a.in0, p0[r], n, 1.0f);
        v[r] = g * c;
      } else if constexpr (INDEXED) {
        v[r] = ld4_indexed_guard(a.in0, a.index, p0[r], n, id, ix[r]);
      } else {
        v[r] = ld4_guard(a.in0, p0[r], n, id);
      }
    }
    v[r] = to_scan_order<REV>(v[r]);
    kk[r] = to_scan_order<REV>(kk[r]);
    if constexpr (BWD) xp[r] = to_scan_order<REV>(xp[r]);
  }

  // key of the element just before this wave's chunk in scan order
  const i64 pn = REV ? (base + kTile - (i64)w * WT) : (base + (i64)w * WT - 1);
  // (both bounds in both directions: in a partial tile the waves whose chunk lies wholly past the end of the array must not
  // touch key[pn] — up to 3 KB behind the allocation; rounds 1 and 2 read it, harmlessly wherever the allocator had mapped
  // more memory behind the keys, and with a memory fault where it had not: found with rocgdb in round 3)
  const bool nb_exists = (pn >= 0) && (pn < n);
  int nbk = 0;
  if (nb_exists) nbk = a.key[pn];

  // look-back chunk 0 (wave 0 only): issued now so its latency overlaps
  const bool do_lb = !FIXUP && !INPLACE && (w == 0) && (lt > 0);  // (in place: see below)
  float4_t lbv = {id, id, id, id};
  int4_t lbk = {0, 0, 0, 0};
  i64 lbp = 0;
  if (do_lb) {
    lbp = REV ? (base + kTile + lane * 4) : (base - (lane * 4 + 4));
    if (!REV) {
      lbk = ld4<ALIGNED>(a.key + lbp);
      if constexpr (BWD) lbv = ld4<ALIGNED>(a.in2 + lbp) * ld4<ALIGNED>(a.in1 + lbp);
      else if constexpr (INDEXED) lbv = ld4_indexed<ALIGNED, false>(a.in0, a.index, lbp, lb_ix);
      else lbv = ld4<ALIGNED>(a.in0 + lbp);
    } else {
      lbk = ld4_guard(a.key, lbp, n, 0);
      if constexpr (BWD) lbv = ld4_guard(a.in2, lbp, n, 0.0f) * ld4_guard(a.in1, lbp, n, 0.0f);
      else if constexpr (INDEXED) lbv = ld4_indexed_guard(a.in0, a.index, lbp, n, id, lb_ix);
      else lbv = ld4_guard(a.in0, lbp, n, id);
    }
  }

  // ---- per-row local scans ------------------------------------------------
  float s[kRows][4];
  float eloc[kRows];     // exclusive prefix of this lane inside the row (no row carry)
  float rowtot[kRows];   // row aggregate (wave-uniform)
  unsigned long long hmask[kRows];
  unsigned openbits = 0;  // bit 4r+k: no head among items 0..k of row r in this lane
  unsigned lanes_open = 0;  // bit r: no head in lanes [0, lane) of row r

#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    // previous key in scan order for item 0
    int lane0_prev;
    if (r == 0) lane0_prev = nbk;
    else lane0_prev = __builtin_amdgcn_readlane(kk[r - 1].w, 63);
    const int pk = dpp_i<0x138, 0xf>(lane0_prev, kk[r].w);  // wave_shr:1, lane 0 keeps old

    bool f0, f1, f2, f3;
    if (FULL) {
      const bool first_of_array = (lt == 0) && (w == 0) && (r == 0) && (lane == 0);
      f0 = first_of_array || (kk[r].x != pk);
      f1 = kk[r].y != kk[r].x;
      f2 = kk[r].z != kk[r].y;
      f3 = kk[r].w != kk[r].z;
    } else {
      // physical index of item k: fwd p0+k, rev p0+3-k; predecessor: fwd p-1, rev p+1
      bool val[4], pex[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const i64 p = REV ? (p0[r] + 3 - k) : (p0[r] + k);
        val[k] = p < n;
        pex[k] = REV ? (p + 1 < n) : (p > 0);
      }
      f0 = val[0] && (!pex[0] || kk[r].x != pk);
      f1 = val[1] && (!pex[1] || kk[r].y != kk[r].x);
      f2 = val[2] && (!pex[2] || kk[r].z != kk[r].y);
      f3 = val[3] && (!pex[3] || kk[r].w != kk[r].z);
    }
    if constexpr (CARRY) {
      // exact chunk carry: the value entering group g is carry[g]; fold it into the head element
      if (f0) v[r].x = M::op(a.carry[kk[r].x], v[r].x);
      if (f1) v[r].y = M::op(a.carry[kk[r].y], v[r].y);
      if (f2) v[r].z = M::op(a.carry[kk[r].z], v[r].z);
      if (f3) v[r].w = M::op(a.carry[kk[r].w], v[r].w);
    }
    const float s0 = v[r].x;
    const float s1 = f1 ? v[r].y : M::op(s0, v[r].y);
    const float s2 = f2 ? v[r].z : M::op(s1, v[r].z);
    const float s3 = f3 ? v[r].w : M::op(s2, v[r].w);
    s[r][0] = s0; s[r][1] = s1; s[r][2] = s2; s[r][3] = s3;
    const bool m0 = !f0, m1 = m0 && !f1, m2 = m1 && !f2, m3 = m2 && !f3;
    openbits |= ((unsigned)m0 | ((unsigned)m1 << 1) | ((unsigned)m2 << 2) | ((unsigned)m3 << 3)) << (4 * r);

    const unsigned long long mask = __ballot(!m3);
    hmask[r] = mask;
    const unsigned long long upto = mask & (~0ull >> (63 - lane));
    const int h = upto ? (63 - __clzll(upto)) : -1;
    if ((mask & ((1ull << lane) - 1ull)) == 0ull) lanes_open |= 1u << r;

    const float inc = wave_seg_scan<MD::kMul>(s3, h, lane);
    eloc[r] = dpp_f<0x138, 0xf>(id, inc);  // lane l <- lane l-1, lane 0 <- identity
    rowtot[r] = readlane_f(inc, 63);
  }

  // wave aggregate with identity carry-in
  float wagg = id;
  bool whead = false;
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    wagg = hmask[r] ? rowtot[r] : M::op(wagg, rowtot[r]);
    whead = whead || (hmask[r] != 0ull);
  }
  // offset (in this wave's chunk, scan order) of the first group head, WT if none
  int wfh = WT;
#pragma unroll
  for (int r = kRows - 1; r >= 0; --r) {
    if (hmask[r]) {
      const int hl = __builtin_ctzll(hmask[r]);
      const unsigned ob = (unsigned)__builtin_amdgcn_readlane((int)openbits, hl);
      wfh = r * 256 + hl * 4 + __builtin_popcount((ob >> (4 * r)) & 0xfu);
    }
  }
  if (lane == 0) { s_wv[w] = wagg; s_wf[w] = whead ? 1 : 0; s_fh[w] = wfh; }

  // ---- look-back (wave 0): carry entering the tile -------------------------
  // Chunk 0 (256 elements) was loaded speculatively with the tile.  If the group
  // reaches further back, chunks are fetched kLbBatch at a time (one dependent
  // round trip per batch) up to one full tile.
  if (w == 0) {
    float tc = FIXUP ? fix_carry : id;
    int unresolved = 0;
    if (do_lb) {
      const int k0 = __builtin_amdgcn_readfirstlane(kk[0].x);
      // returns true when the look-back is finished (group start found, or array end)
      auto process = [&](float4_t cv, int4_t ck, i64 cp, int j) -> bool {
        // scan-order distance of item k of this lane: j*256 + lane*4 + k + 1
        const float4_t lv = REV ? cv : to_scan_order<true>(cv);
        const int4_t lk = REV ? ck : to_scan_order<true>(ck);
        bool c0, c1, c2, c3;
        if (!REV) {
          c0 = lk.x == k0; c1 = lk.y == k0; c2 = lk.z == k0; c3 = lk.w == k0;
        } else {
          c0 = (cp + 0 < n) && lk.x == k0; c1 = (cp + 1 < n) && lk.y == k0;
          c2 = (cp + 2 < n) && lk.z == k0; c3 = (cp + 3 < n) && lk.w == k0;
        }
        float p = id;
        bool full = false;
        if (c0) { p = lv.x; if (c1) { p = M::op(p, lv.y); if (c2) { p = M::op(p, lv.z); if (c3) { p = M::op(p, lv.w); full = true; } } } }
        const unsigned long long fm = __ballot(full);
        const int L = (fm == ~0ull) ? 64 : __builtin_ctzll(~fm);
        const float contrib = (lane <= L) ? p : id;
        tc = M::op(wave_reduce<MD::kMul>(contrib), tc);
        if (L < 64) return true;  // group start found inside this chunk
        const bool more = REV ? (base + kTile + (i64)(j + 1) * 256 < n) : (base - (i64)(j + 1) * 256 > 0);
        return !more;             // reached the end of the array: resolved
      };
      bool done = process(lbv, lbk, lbp, 0);
      // A group that already spans the 256 elements behind the tile AND this wave's whole 1024-element share is long:
      // stop reading raw inputs (up to three more dependent round trips that would most likely end at the window's
      // edge) and take the carry from the descriptor tree.  A data-determined rule, so still deterministic.
      bool wave_headless = GCP_LB_EARLY_EXIT != 0;
#pragma unroll
      for (int r = 0; r < kRows; ++r) wave_headless = wave_headless && (hmask[r] == 0ull);
      for (int j = 1; !done && !wave_headless && j < kLbChunks; j += kLbBatch) {
        float4_t cv[kLbBatch];
        int4_t ck[kLbBatch];
        i64 cp[kLbBatch];
#pragma unroll
        for (int c = 0; c < kLbBatch; ++c) {
          cp[c] = REV ? (base + kTile + (i64)(j + c) * 256 + lane * 4) : (base - ((i64)(j + c) * 256 + lane * 4 + 4));
          if (!REV) {
            ck[c] = ld4<ALIGNED>(a.key + cp[c]);
            if constexpr (BWD) cv[c] = ld4<ALIGNED>(a.in2 + cp[c]) * ld4<ALIGNED>(a.in1 + cp[c]);
            else if constexpr (INDEXED) cv[c] = ld4_indexed<ALIGNED, false>(a.in0, a.index, cp[c], lb_ix);
            else cv[c] = ld4<ALIGNED>(a.in0 + cp[c]);
          } else {
            ck[c] = ld4_guard(a.key, cp[c], n, 0);
            if constexpr (BWD) cv[c] = ld4_guard(a.in2, cp[c], n, 0.0f) * ld4_guard(a.in1, cp[c], n, 0.0f);
            else if constexpr (INDEXED) cv[c] = ld4_indexed_guard(a.in0, a.index, cp[c], n, id, lb_ix);
            else cv[c] = ld4_guard(a.in0, cp[c], n, id);
          }
        }
#pragma unroll
        for (int c = 0; c < kLbBatch; ++c) {
          if (!done) done = process(cv[c], ck[c], cp[c], j + c);
        }
      }
      // window exhausted: the whole tile before this one belongs to the group; tc is its product and the search goes on
      // on the tile descriptors after the barrier
      if (!done) unresolved = 1;
      if constexpr (CARRY) {
        // group start found behind the tile (first element is not a head): its carry belongs to the prefix
        const bool first_is_head = nb_exists ? (nbk != k0) : true;
        if (done && !first_is_head) tc = M::op(a.carry[k0], tc);
      }
    }
    if (!FIXUP && INPLACE && lt > 0) {
      // in place the neighbouring tile's inputs may already be overwritten by its results: no raw look-back at all — a tile
      // that continues a group takes its carry from the descriptor tree (aggregates only, computed before any store)
      const int k0 = __builtin_amdgcn_readfirstlane(kk[0].x);
      if (nb_exists && nbk == k0) unresolved = 1;
    }
    // [kWaves]: the raw window was exhausted (never rewritten: every wave branches on it after the barrier);
    // [kWaves + 1]: the carry is still unknown (cleared by the descriptor walk when it succeeds)
    if (lane == 0) { s_tc[0] = tc; s_wf[kWaves] = unresolved; s_wf[kWaves + 1] = unresolved; }
  }
  __syncthreads();

  // ---- tile descriptors: the canonical radix-64 block tree ------------------------------
  // Level L, index i holds the segmented aggregate of tiles [i 64^L, (i+1) 64^L): {has a head, aggregate of the
  // trailing group}.  Tile t publishes level 0, and level L >= 1 when it is the LAST tile of a level-L block.  Every
  // value in the tree is a fixed-association function of the level-0 entries, so whoever needs the prefix entering
  // tile t — per level, the up to 63 blocks between the enclosing block's start and t, nearest first, until one holds a
  // head — gets the same bits in every run, whatever the timing: deterministic, and a chain of at most kLevels
  // publish-then-read hops behind the neighbours' local scans.
  //   * a tile that holds a head (the usual case) publishes {head, trailing aggregate} to all its levels at once:
  //     nothing read, and 1.016 stores per tile on average;
  //   * a tile without a head (a group longer than the tile) that ends a block reads the block's other 63 children first;
  //   * a tile whose raw window did not reach the start of its group takes its carry from the tree.
  // Both waits are bounded (a.patience); what times out is left to the follow-up kernel.
  // (first_head: elements [0, first_head) of the tile, scan order, take the carry-in; agg: aggregate of the tile's trailing
  // group relative to an identity carry-in — both only where a descriptor is written)
  auto tile_summary = [&](int& first_head, float& agg) {
    first_head = kTile;
    agg = id;
#pragma unroll
    for (int j = kWaves - 1; j >= 0; --j)
      if (s_wf[j]) first_head = j * WT + s_fh[j];
#pragma unroll
    for (int j = 0; j < kWaves; ++j) agg = s_wf[j] ? s_wv[j] : M::op(agg, s_wv[j]);
  };
  const bool has_head = (s_wf[0] | s_wf[1] | s_wf[2] | s_wf[3]) != 0;
  static_assert(kWaves == 4, "has_head reads four wave flags");
  // levels whose block ends with this tile: 1 .. zl (trailing base-64 digits equal to 63)
  int zl = 0;
  while (zl < kLevels - 1 && ((lt >> (6 * zl)) & 63) == 63) ++zl;
  auto entry = [&](int level, i64 idx) -> unsigned long long* { return desc + 2 * (a.lvl_off[level] + idx); };
  const int lvl = lane < kLevels ? lane : kLevels - 1;  // lane -> level for the lanes that publish upper levels (in bounds for all)
  const bool need_carry = s_wf[kWaves] != 0;
  if (!FIXUP && a.ntiles > 1 && (need_carry || (!has_head && zl > 0))) {  // block-uniform, rare
    if (w == 0) {
      int first_head;
      float agg;
      tile_summary(first_head, agg);
      // level 0 first: the tiles behind need it whatever this tile is still waiting for
      if (lane == 0)
        __hip_atomic_store(entry(0, lt), pack_desc(agg, has_head ? 0u : kDOpen, first_head), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (has_head && lane >= 1 && lane <= zl)
        __hip_atomic_store(entry(lvl, lt >> (6 * lvl)), pack_desc(agg, 0u, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // lane i at level L looks at block (t >> 6L) - 1 - i of that level, for i < digit L of t: the blocks between the
      // start of the enclosing level-(L+1) block and this tile, nearest first
      int pub_level = has_head ? zl : 0;  // levels 1 .. pub_level are published
      bool resolved = !need_carry;
      float tc = id;          // the carry (need_carry)
      float cur = agg;        // this tile's own block at the level being published (no head so far)
      bool cur_closed = false;
      if (a.patience >= 0) {
        const unsigned long long t0 = wall_clock64();
        float acc = id;       // head-less blocks gathered so far, levels below `lv`
        int lv = 0;           // level the carry search has reached
        while (true) {
          unsigned long long d[kLevels];
#pragma unroll
          for (int L = 0; L < kLevels; ++L) {
            const i64 self = lt >> (6 * L);
            d[L] = 0ull;
            if (lane < (int)(self & 63)) d[L] = __hip_atomic_load(entry(L, self - 1 - lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int L = 0; L < kLevels; ++L) {
            const int digit = (int)((lt >> (6 * L)) & 63);
            const unsigned fl = (unsigned)(d[L] >> 32);
            const float val = __builtin_bit_cast(float, (unsigned)d[L]);
            const bool act = lane < digit;
            const unsigned long long am = digit ? ((1ull << digit) - 1ull) : 0ull;
            const unsigned long long vm = __ballot(act && (fl & kDValid) != 0u);
            const unsigned long long cm = __ballot(act && (fl & kDValid) != 0u && (fl & kDOpen) == 0u);
            const int lf = cm ? __builtin_ctzll(cm) : 63;
            const unsigned long long want = am & ((lf >= 63) ? ~0ull : ((2ull << lf) - 1ull));  // up to the nearest head
            const bool ready = (vm & want) == want;
            // publish level L+1 (this tile ends that block and holds no head): its 63 other children are this level's lanes
            if (L == pub_level && L < zl && ready) {
              if (!cur_closed) cur = M::op(wave_reduce<MD::kMul>(((want >> lane) & 1ull) ? val : id), cur);
              cur_closed = cur_closed || cm != 0ull;
              if (lane == 0)
                __hip_atomic_store(entry(L + 1, lt >> (6 * (L + 1))), pack_desc(cur, cur_closed ? 0u : kDOpen, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              pub_level = L + 1;
            }
            // the carry: level by level, nearest blocks first
            if (!resolved && L == lv && ready) {
              acc = M::op(wave_reduce<MD::kMul>(((want >> lane) & 1ull) ? val : id), acc);  // older blocks on the left
              if (cm != 0ull || (lt >> (6 * (L + 1))) == 0) { tc = acc; resolved = true; }  // a head, or the start of the array
              else lv = L + 1;
            }
          }
          if (pub_level >= zl && resolved) break;
          if (lv >= kLevels) break;  // cannot happen (level kLevels-1 always reaches the start of the array)
          if (wall_clock64() - t0 > (unsigned long long)a.patience) break;
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (lane == 0 && need_carry) {
        s_tc[0] = resolved ? tc : id;  // unresolved: the follow-up kernel re-runs this tile with the carry from the completed tree
        s_wf[kWaves + 1] = resolved ? 0 : 1;
        if (!resolved) a.hdr[kHdrAnyUnresolved] = 1u;
      }
    }
    __syncthreads();
  }

  // ---- carry into this wave, final values, stores ----------------------------
  float R = s_tc[0];
  const int unresolved = s_wf[kWaves + 1];
#pragma unroll
  for (int j = 0; j < kWaves; ++j)
    if (j < w) R = s_wf[j] ? s_wv[j] : M::op(R, s_wv[j]);
  // ---- level-0 descriptor in its final form (and, for a tile with a head, all its levels), before the stores:
  //      the tiles behind may be waiting for it ----
  if (!FIXUP && w == kWaves - 1) {
    if (lt == 0 && lane == 0) {  // per-launch bookkeeping
      a.hdr[kHdrUnresolved] = 0;
      a.hdr[kHdrDescResolved] = 0;
      a.hdr[kHdrTiles + (a.hdr[kHdrEpoch] & 1u)] = (unsigned)a.lvl_off[kLevels];  // what the next launch's follow-up kernel clears
    }
    if (a.ntiles > 1) {
      int first_head;
      float agg;
      tile_summary(first_head, agg);
      const unsigned fl = (has_head ? 0u : kDOpen) | (unresolved ? kDUnresolved : 0u) | ((need_carry && !unresolved) ? kDTree : 0u);
      if (lane == 0)
        __hip_atomic_store(entry(0, lt), pack_desc(agg, fl, first_head), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (has_head && !need_carry && lane >= 1 && lane <= zl)
        __hip_atomic_store(entry(lvl, lt >> (6 * lvl)), pack_desc(agg, 0u, 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // in place a tile whose carry is still unknown must not store: the follow-up kernel re-runs it from its intact inputs
  // (out of place the re-run simply overwrites what is stored here)
  if constexpr (INPLACE) {
    if (__builtin_amdgcn_readfirstlane(unresolved)) return;
  }
#pragma unroll
  for (int r = 0; r < kRows; ++r) {
    const float e = ((lanes_open >> r) & 1u) ? M::op(R, eloc[r]) : eloc[r];
    float4_t y;
    y.x = ((openbits >> (4 * r + 0)) & 1u) ? M::op(e, s[r][0]) : s[r][0];
    y.y = ((openbits >> (4 * r + 1)) & 1u) ? M::op(e, s[r][1]) : s[r][1];
    y.z = ((openbits >> (4 * r + 2)) & 1u) ? M::op(e, s[r][2]) : s[r][2];
    y.w = ((openbits >> (4 * r + 3)) & 1u) ? M::op(e, s[r][3]) : s[r][3];
    if constexpr (BWD) {
      // reference: grouped_cumprod_backward.cu:25  param_idx = param != 0 ? param : 1e-8f
      y.x = y.x / (xp[r].x != 0.0f ? xp[r].x : 1e-8f);
      y.y = y.y / (xp[r].y != 0.0f ? xp[r].y : 1e-8f);
      y.z = y.z / (xp[r].z != 0.0f ? xp[r].z : 1e-8f);
      y.w = y.w / (xp[r].w != 0.0f ? xp[r].w : 1e-8f);
    }
    y = to_scan_order<REV>(y);  // involution: back to memory order
    if constexpr (INDEXED) {  // un-sort: through the permutation (gs_model.py:555 `output[torch.argsort(index)]`)
      if (FULL || p0[r] + 0 < n) a.out[ix[r].x] = y.x;
      if (FULL || p0[r] + 1 < n) a.out[ix[r].y] = y.y;
      if (FULL || p0[r] + 2 < n) a.out[ix[r].z] = y.z;
      if (FULL || p0[r] + 3 < n) a.out[ix[r].w] = y.w;
    } else if (FULL) {
      st4<ALIGNED>(a.out + p0[r], y);
    } else {
      if (p0[r] + 0 < n) a.out[p0[r] + 0] = y.x;
      if (p0[r] + 1 < n) a.out[p0[r] + 1] = y.y;
      if (p0[r] + 2 < n) a.out[p0[r] + 2] = y.z;
      if (p0[r] + 3 < n) a.out[p0[r] + 3] = y.w;
    }
    R = hmask[r] ? rowtot[r] : M::op(R, rowtot[r]);
  }

}

// Forward modes: six blocks per CU (<= 80 VGPRs), which the rare descriptor walk must not cost; the reverse modes are
// left to the register allocator (the backward holds three arrays per element and runs at three blocks per CU).
template <int MODE, bool ALIGNED, bool CARRY, bool INDEXED = false, bool INPLACE = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu((Mode<MODE>::kRev || INDEXED || INPLACE) ? 1 : 6, (Mode<MODE>::kRev || INDEXED || INPLACE) ? 8 : 6)))
void gcp_scan_main(const ScanArgs a) {
  __shared__ float s_wv[kWaves];
  __shared__ int s_wf[kWaves + 2];
  __shared__ float s_tc[1];
  __shared__ int s_fh[kWaves];
  const i64 lt = logical_tile((i64)blockIdx.x, a.ntiles, a.xcd_remap);
  const i64 pt = Mode<MODE>::kRev ? (a.ntiles - 1 - lt) : lt;
  if ((pt + 1) * (i64)kTile <= a.n) scan_tile<MODE, ALIGNED, true, CARRY, false, INDEXED, INPLACE>(a, lt, s_wv, s_wf, s_tc, s_fh);
  else scan_tile<MODE, ALIGNED, false, CARRY, false, INDEXED, INPLACE>(a, lt, s_wv, s_wf, s_tc, s_fh);
}

// ----------------------------------------------------------------------------
// Follow-up ("fallback") kernel, one small launch after every multi-tile scan.  It maintains the
// workspace (counts the introspection flags, clears the descriptor set the next launch will publish
// into, advances the launch counter) and — only when some tile of the main kernel gave up waiting for
// the descriptor tree (kHdrAnyUnresolved; every long-group tile in two-pass mode) — finishes those
// tiles with THE SAME BITS the tree would have given them in time:
//   1. the upper tree levels are completed bottom-up (one wave per missing entry, the association of the
//      main kernel's publisher; a grid barrier between levels — the grid is at most kFixBlocks blocks,
//      all resident),
//   2. wave 0 of the block that owns an unresolved tile takes the tile's carry from the tree with the
//      search of the main kernel (nothing to wait for any more),
//   3. the block re-runs scan_tile<FIXUP> on the tile with that carry.
// Every value involved is a fixed-association function of the level-0 descriptors and the raw inputs,
// so a launch with time-outs is bit-identical to one without (tests/test_scan_gpu.py::
// test_results_do_not_depend_on_the_descriptor_wait).
// ----------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ld_entry(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the up-to-`count` entries idx-1, idx-2, ... of one level, nearest first (lane i = entry idx-1-i), reduced up to and
// including the nearest one that holds a head: exactly the lanes, mask and DPP tree of the main kernel's walk
template <bool MUL>
__device__ __forceinline__ float reduce_nearest(const unsigned long long* level, i64 idx, int count, int lane, bool& closed) {
  typedef Monoid<MUL> M;
  unsigned long long d = 0ull;
  if (lane < count) d = ld_entry(level + 2 * (idx - 1 - lane));
  const unsigned fl = (unsigned)(d >> 32);
  const float val = __builtin_bit_cast(float, (unsigned)d);
  const unsigned long long am = count ? ((count >= 64) ? ~0ull : ((1ull << count) - 1ull)) : 0ull;
  const unsigned long long cm = __ballot(lane < count && (fl & kDOpen) == 0u);
  const int lf = cm ? __builtin_ctzll(cm) : 63;
  const unsigned long long want = am & ((lf >= 63) ? ~0ull : ((2ull << lf) - 1ull));
  closed = cm != 0ull;
  return wave_reduce<MUL>(((want >> lane) & 1ull) ? val : M::identity());
}

// entry `idx` of level L >= 1 from its 64 children at level L - 1 (what the last tile of the block publishes in the main kernel)
template <bool MUL>
__device__ __forceinline__ unsigned long long rebuild_entry(const unsigned long long* desc, const i64* lvl_off, int L, i64 idx, int lane) {
  typedef Monoid<MUL> M;
  const unsigned long long* level = desc + 2 * lvl_off[L - 1];
  const i64 lastc = (idx << 6) + 63;
  const unsigned long long last = ld_entry(level + 2 * lastc);
  const float lval = __builtin_bit_cast(float, (unsigned)last);
  if ((((unsigned)(last >> 32)) & kDOpen) == 0u) return pack_desc(lval, 0u, 0);  // the last child holds a head: its trailing aggregate
  bool closed;
  const float part = reduce_nearest<MUL>(level, lastc, 63, lane, closed);
  return pack_desc(M::op(part, lval), closed ? 0u : kDOpen, 0);
}

// the prefix entering tile lt, from the completed tree: level by level, nearest blocks first (main kernel, "the carry")
template <bool MUL>
__device__ __forceinline__ float tree_carry(const unsigned long long* desc, const i64* lvl_off, i64 lt, int lane) {
  typedef Monoid<MUL> M;
  float acc = M::identity();
  for (int L = 0; L < kLevels; ++L) {
    const i64 self = lt >> (6 * L);
    bool closed;
    const float part = reduce_nearest<MUL>(desc + 2 * lvl_off[L], self, (int)(self & 63), lane, closed);
    acc = M::op(part, acc);  // older blocks on the left
    if (closed || (lt >> (6 * (L + 1))) == 0) break;
  }
  return acc;
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
  }
  __syncthreads();
}

// Every block, whatever it finds, also (i) clears its share of the OTHER descriptor set over the range its last user
// wrote — the set the next launch on this workspace will publish into — and (ii) counts itself done; the last block
// to finish advances the launch counter, which flips the sets.  Nobody reads the other set or the counter's parity
// after that point in this launch, so neither needs a barrier.
template <int MODE, bool CARRY, bool INDEXED = false>
__global__ __launch_bounds__(kThreads) void gcp_fallback(const ScanArgs a) {
  typedef Mode<MODE> MD;
  __shared__ float s_wv[kWaves];
  __shared__ int s_wf[kWaves + 2];
  __shared__ float s_tc[1];
  __shared__ int s_fh[kWaves];
  __shared__ int s_any, s_tree, s_cnt;
  __shared__ int s_tile[64];
  __shared__ float s_carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned epoch = a.hdr[kHdrEpoch];
  const unsigned set = epoch & 1u;
  unsigned long long* const desc = a.desc_sets + set;
  const bool any = a.hdr[kHdrAnyUnresolved] != 0u;  // the same for every block: written by the main kernel only
  const i64 per = (a.ntiles + gridDim.x - 1) / gridDim.x;
  const i64 r0 = (i64)blockIdx.x * per;
  const i64 r1 = (r0 + per < a.ntiles) ? (r0 + per) : a.ntiles;

  // introspection counters of this launch, from the flags of the level-0 descriptors
  if (tid == 0) { s_any = 0; s_tree = 0; }
  __syncthreads();
  {
    int mine = 0, tree = 0;
    for (i64 t = r0 + tid; t < r1; t += kThreads) {
      const unsigned fl = (unsigned)(desc[2 * t] >> 32);
      mine += (int)((fl >> 1) & 1u);
      tree += (int)((fl >> 16) & 1u);
    }
    if (mine) atomicAdd(&s_any, mine);
    if (tree) atomicAdd(&s_tree, tree);
  }
  __syncthreads();
  if (tid == 0 && s_tree) atomicAdd(a.hdr + kHdrDescResolved, (unsigned)s_tree);
  if (tid == 0 && s_any) atomicAdd(a.hdr + kHdrUnresolved, (unsigned)s_any);

  if (any) {  // grid-uniform
    // 1. complete the upper levels, bottom-up
    unsigned phase = 0;
    for (int L = 1; L < kLevels; ++L) {
      const i64 n_complete = a.ntiles >> (6 * L);  // blocks of 64^L tiles that lie wholly inside the array
      if (n_complete == 0) break;
      unsigned long long* const level = desc + 2 * a.lvl_off[L];
      for (i64 idx = (i64)blockIdx.x * kWaves + w; idx < n_complete; idx += (i64)gridDim.x * kWaves) {
        if ((((unsigned)(ld_entry(level + 2 * idx) >> 32)) & kDValid) != 0u) continue;  // published in time (wave-uniform)
        const unsigned long long e = rebuild_entry<MD::kMul>(desc, a.lvl_off, L, idx, lane);
        if (lane == 0) __hip_atomic_store(level + 2 * idx, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      grid_barrier(a.hdr + kHdrBarrier, ++phase * gridDim.x);
    }
    // 2. + 3. this block's unresolved tiles
    if (s_any) {  // block-uniform
      for (i64 c0 = r0; c0 < r1; c0 += 64) {
        if (w == 0) {
          const i64 t = c0 + lane;
          const bool unres = t < r1 && ((((unsigned)(ld_entry(desc + 2 * t) >> 32)) & kDUnresolved) != 0u);
          const unsigned long long um = __ballot(unres);
          if (unres) s_tile[__builtin_popcountll(um & ((1ull << lane) - 1ull))] = lane;
          if (lane == 0) s_cnt = __builtin_popcountll(um);
        }
        __syncthreads();
        const int cnt = s_cnt;
        for (int i = 0; i < cnt; ++i) {
          const i64 lt = c0 + s_tile[i];
          if (w == 0) {
            const float c = tree_carry<MD::kMul>(desc, a.lvl_off, lt, lane);
            if (lane == 0) s_carry = c;
          }
          __syncthreads();
          const float c = s_carry;
          // guarded dword form of the tile routine: the same arithmetic as the vector form on any alignment and tile
          scan_tile<MODE, false, false, CARRY, true, INDEXED>(a, lt, s_wv, s_wf, s_tc, s_fh, c);
          __syncthreads();
        }
        __syncthreads();
      }
    }
  }

  unsigned long long* const other = a.desc_sets + (set ^ 1u);
  const i64 n_other = (i64)a.hdr[kHdrTiles + (set ^ 1u)];
  const i64 oper = (n_other + gridDim.x - 1) / gridDim.x;
  const i64 z0 = (i64)blockIdx.x * oper;
  const i64 z1 = (z0 + oper < n_other) ? (z0 + oper) : n_other;
  for (i64 t = z0 + tid; t < z1; t += kThreads) other[2 * t] = 0ull;
  __syncthreads();  // every thread's clears are issued before the block reports itself done
  if (tid == 0) {
    __threadfence();
    if (atomicAdd(a.hdr + kHdrDone, 1u) == gridDim.x - 1) {
      a.hdr[kHdrDone] = 0;
      a.hdr[kHdrAnyUnresolved] = 0;
      a.hdr[kHdrBarrier] = 0;
      a.hdr[kHdrEpoch] = epoch + 1u;
    }
  }
}

// ----------------------------------------------------------------------------
// gcp_check_groups kernel
// ----------------------------------------------------------------------------
__global__ void gcp_check_groups_kernel(const int* inv, const int* inv_len, i64 n, i64 G,
                                        unsigned long long* bad) {
  const i64 stride = (i64)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int g = inv[i];
    if (g < 0 || g >= G) { ++local; continue; }
    if (i == 0) { if (g != 0) ++local; }
    else {
      const int gp = inv[i - 1];
      if (g != gp && g != gp + 1) ++local;
      if (g == gp + 1 && gp >= 0 && gp < G && inv_len[gp] != (int)i) ++local;
    }
    if (i == n - 1) { if (g != G - 1 || inv_len[g] != (int)n) ++local; }
  }
  if (local) atomicAdd(bad, local);
}

// gcp_check_permutation: every index[i] in [0, n) and none twice (one bit per value; the atomic OR returns whether the
// bit was already set).  Integer counts only: the result does not depend on the order.
__global__ void gcp_check_perm_kernel(const int* index, i64 n, unsigned* seen, unsigned long long* bad) {
  const i64 stride = (i64)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int v = index[i];
    if (v < 0 || (i64)v >= n) { ++local; continue; }
    const unsigned bit = 1u << (v & 31);
    if (atomicOr(seen + (v >> 5), bit) & bit) ++local;
  }
  if (local) atomicAdd(bad, local);
}

// gcp_check_group_ids: every inv[i] in [0, G) — what the carry forms index `carry` with
__global__ void gcp_check_ids_kernel(const int* inv, i64 n, i64 G, unsigned long long* bad) {
  const i64 stride = (i64)gridDim.x * blockDim.x;
  unsigned long long local = 0;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int g = inv[i];
    if (g < 0 || (i64)g >= G) ++local;
  }
  if (local) atomicAdd(bad, local);
}

// ----------------------------------------------------------------------------
// Host side
// ----------------------------------------------------------------------------
// entries of the descriptor tree for n elements: level L has one entry per 64^L tiles
inline void ws_levels(i64 n, i64* off /*[kLevels + 1]*/) {
  const i64 t = (n + kTile - 1) / kTile;
  off[0] = 0;
  for (int L = 0; L < kLevels; ++L) off[L + 1] = off[L] + (t > 0 ? ((t - 1) >> (6 * L)) + 1 : 1);
}

size_t ws_bytes_for(i64 n) {
  i64 off[kLevels + 1];
  ws_levels(n > 0 ? n : 0, off);
  // header + two interleaved sets of descriptor entries
  return ((size_t)kWsHeaderBytes + (size_t)off[kLevels] * 2u * 8u + 255u) / 256u * 256u;
}

struct InternalWs {
  void* ptr = nullptr;
  size_t bytes = 0;
};
std::mutex g_ws_mutex;
InternalWs g_ws[64];

int get_internal_ws(size_t need, hipStream_t stream, void** out) {
  int dev = 0;
  GCP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return GCP_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  InternalWs& w = g_ws[dev];
  if (w.bytes < need) {
    if (w.ptr) {
      GCP_HIP(hipDeviceSynchronize());
      GCP_HIP(hipFree(w.ptr));
      w.ptr = nullptr; w.bytes = 0;
    }
    size_t cap = need + need / 2;
    cap = (cap + 255) / 256 * 256;
    GCP_HIP(hipMalloc(&w.ptr, cap));
    w.bytes = cap;
    GCP_HIP(hipMemsetAsync(w.ptr, 0, cap, stream));  // launch counter 0, both descriptor sets clear
  }
  *out = w.ptr;
  return GCP_OK;
}

// compute units of the current device (0 if the query fails: the caller then keeps its own bound)
int device_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v == 0) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) return 0;
    cached[dev].store(cu, std::memory_order_relaxed);
    v = cu;
  }
  return v;
}

std::atomic<long long> g_patience_us{-2};  // -2: not set yet (environment GCP_DESC_WAIT_US, else the default)
std::atomic<int> g_validate{-1};           // -1: not set yet (environment GCP_VALIDATE_OPERANDS, else off)

bool validate_operands() {
  int v = g_validate.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* s = getenv("GCP_VALIDATE_OPERANDS");
    v = (s && *s && atoi(s) != 0) ? 1 : 0;
    g_validate.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

// Count, on `stream`, what one of the two checking kernels finds; synchronises.  seen_words > 0: a zeroed bitmap of that
// many words is handed to the kernel (the permutation check).
template <typename Launch>
int count_bad(hipStream_t stream, size_t seen_words, unsigned long long* h_out, Launch launch) {
  char* d = nullptr;
  const size_t bytes = 8 + seen_words * 4;
  GCP_HIP(hipMalloc((void**)&d, bytes));
  hipError_t e = hipMemsetAsync(d, 0, bytes, stream);
  if (e == hipSuccess) {
    launch((unsigned long long*)d, (unsigned*)(d + 8));
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h_out, d, sizeof(*h_out), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d);
  return e == hipSuccess ? GCP_OK : hip_fail(e);
}

int check_permutation_impl(const int* index, i64 n, int64_t* n_bad, hipStream_t stream) {
  if (n_bad) *n_bad = 0;
  if (n < 0 || n > 0x7fffffffLL) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!index) return GCP_ERR_INVALID_ARGUMENT;
  unsigned long long h = 0;
  i64 blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const int st = count_bad(stream, (size_t)((n + 31) / 32), &h, [&](unsigned long long* bad, unsigned* seen) {
    hipLaunchKernelGGL(gcp_check_perm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, index, n, seen, bad);
  });
  if (st != GCP_OK) return st;
  if (n_bad) *n_bad = (int64_t)h;
  return h ? GCP_ERR_INVALID_ARGUMENT : GCP_OK;
}

int check_group_ids_impl(const int* inv, i64 n, i64 n_groups, int64_t* n_bad, hipStream_t stream) {
  if (n_bad) *n_bad = 0;
  if (n < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!inv || n_groups <= 0) return GCP_ERR_INVALID_ARGUMENT;
  unsigned long long h = 0;
  i64 blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const int st = count_bad(stream, 0, &h, [&](unsigned long long* bad, unsigned*) {
    hipLaunchKernelGGL(gcp_check_ids_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, inv, n, n_groups, bad);
  });
  if (st != GCP_OK) return st;
  if (n_bad) *n_bad = (int64_t)h;
  return h ? GCP_ERR_INVALID_ARGUMENT : GCP_OK;
}

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

template <int MODE>
int launch_scan(const float* in0, const float* in1, const float* in2, const int* key, float* out,
                i64 n, void* ws, size_t ws_bytes, void* stream_, const float* carry = nullptr, const int* index = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n == 0) return GCP_OK;
  if (!in0 || !key || !out) return GCP_ERR_INVALID_ARGUMENT;
  if (Mode<MODE>::kBwd && (!in1 || !in2)) return GCP_ERR_INVALID_ARGUMENT;
  // No aliasing: the look-back re-reads the neighbouring tile's RAW inputs while that tile's block may already be
  // storing its outputs, so an output range that shares bytes with an input range races between blocks.
  // Exactly in place (out == x) is served — the reference's thrust::inclusive_scan_by_key allows it
  // (grouped_cumprod_forward.cu:17-23) — by a mode that never re-reads another tile's inputs; any other overlap is refused.
  const bool inplace = !Mode<MODE>::kBwd && !index && !carry && (const void*)out == (const void*)in0;
  {
    const uintptr_t o0 = (uintptr_t)out, o1 = o0 + (uintptr_t)n * 4u;
    const void* ins[5] = {inplace ? nullptr : in0, in1, in2, key, index};
    for (const void* q : ins) {
      const uintptr_t q0 = (uintptr_t)q;
      if (q && o0 < q0 + (uintptr_t)n * 4u && q0 < o1) return GCP_ERR_INVALID_ARGUMENT;
    }
  }
  const i64 ntiles = (n + kTile - 1) / kTile;
  if (ntiles > 0x7fffffffLL) return GCP_ERR_INVALID_ARGUMENT;

  const size_t need = ws_bytes_for(n);
  if (ws == nullptr) {
    const int st = get_internal_ws(need, stream, &ws);
    if (st != GCP_OK) return st;
  } else {
    if (ws_bytes < need || ((uintptr_t)ws & 255u)) return GCP_ERR_WORKSPACE;
  }
  char* p = (char*)ws;
  ScanArgs a;
  a.in0 = in0; a.in1 = in1; a.in2 = in2; a.key = key; a.out = out; a.carry = carry; a.index = index;
  a.n = n; a.ntiles = ntiles;
  a.hdr = (unsigned*)p; p += kWsHeaderBytes;
  a.desc_sets = (unsigned long long*)p;
  ws_levels(n, a.lvl_off);
  static const int xcd_remap = env_int("GCP_XCD_REMAP", GCP_XCD_REMAP_DEFAULT);
  static const int chunk_lg = [] { int l = 0; while ((1 << l) < kXcdChunk) ++l; return l; }();
  static const int chunk_lg_indexed = env_int("GCP_XCD_CHUNK_INDEXED_LOG2", 7);
  a.xcd_remap = xcd_remap ? 1 + (index ? chunk_lg_indexed : chunk_lg) : 0;
  long long us = g_patience_us.load(std::memory_order_relaxed);
  if (us == -2) { us = env_int("GCP_DESC_WAIT_US", GCP_DESC_WAIT_US); g_patience_us.store(us, std::memory_order_relaxed); }
  a.patience = us < 0 ? -1 : us * 100;  // wall_clock64() counts at 100 MHz

  uintptr_t al = (uintptr_t)in0 | (uintptr_t)key | (uintptr_t)out;
  if (Mode<MODE>::kBwd) al |= (uintptr_t)in1 | (uintptr_t)in2;
  if (index) al |= (uintptr_t)index;
  const bool aligned = (al & 15u) == 0;

  const dim3 grid((unsigned)ntiles), block(kThreads);
  if constexpr (Mode<MODE>::kBwd) {
    if (carry || index) return GCP_ERR_INVALID_ARGUMENT;
    if (aligned) hipLaunchKernelGGL((gcp_scan_main<MODE, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gcp_scan_main<MODE, false, false>), grid, block, 0, stream, a);
  } else if (inplace) {  // out IS in0 (legal for the reference's Thrust scans): no tile may re-read another tile's inputs
    if (aligned) hipLaunchKernelGGL((gcp_scan_main<MODE, true, false, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gcp_scan_main<MODE, false, false, false, true>), grid, block, 0, stream, a);
  } else if (index) {
    if (carry) return GCP_ERR_INVALID_ARGUMENT;
    if (aligned) hipLaunchKernelGGL((gcp_scan_main<MODE, true, false, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gcp_scan_main<MODE, false, false, true>), grid, block, 0, stream, a);
  } else if (carry) {
    if (aligned) hipLaunchKernelGGL((gcp_scan_main<MODE, true, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gcp_scan_main<MODE, false, true>), grid, block, 0, stream, a);
  } else {
    if (aligned) hipLaunchKernelGGL((gcp_scan_main<MODE, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gcp_scan_main<MODE, false, false>), grid, block, 0, stream, a);
  }
  GCP_HIP(hipGetLastError());
  if (ntiles > 1) {
    // the follow-up kernel: a no-op walk over the tile descriptors unless a tile gave up waiting.  Every block ends
    // with one atomic on the same word (the last one advances the launch counter): 256 of them cost 7 us, so the
    // grid is small unless the descriptor walk is switched off and the kernel has real work on every long group
    // Its rare path spins in a hand-written grid barrier, so every block of the grid must be resident together: never
    // more blocks than the device has CUs (one 256-thread block of this kernel always fits a CU: 256 on a whole MI355X,
    // 32 on a CPX partition), queried once per device.
    i64 want = a.patience < 0 ? kFixBlocks : kFixBlocksQuiet;
    const int cus = device_cu_count();
    if (cus > 0 && want > cus) want = cus;
    const unsigned fb = (unsigned)(ntiles < want ? ntiles : want);
    if (index) hipLaunchKernelGGL((gcp_fallback<MODE, false, !Mode<MODE>::kBwd>), dim3(fb), dim3(kThreads), 0, stream, a);
    else if (carry) hipLaunchKernelGGL((gcp_fallback<MODE, !Mode<MODE>::kBwd>), dim3(fb), dim3(kThreads), 0, stream, a);
    else hipLaunchKernelGGL((gcp_fallback<MODE, false>), dim3(fb), dim3(kThreads), 0, stream, a);
    GCP_HIP(hipGetLastError());
  }
  return GCP_OK;
}

}  // namespace

extern "C" {

int gcp_abi_version(void) { return GCP_ABI_VERSION; }

#ifndef GCP_SOURCE_HASH
#define GCP_SOURCE_HASH "unknown"
#endif
const char* gcp_source_hash(void) {
  static const char tagged[] = "GCPSRCHASH:" GCP_SOURCE_HASH;  // the tag lets the build recipe find it in the file
  return tagged + 11;
}

int gcp_last_hip_error(void) { return gcp::t_last_hip_error; }

const char* gcp_status_string(int status) {
  switch (status) {
    case GCP_OK: return "ok";
    case GCP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GCP_ERR_WORKSPACE: return "workspace too small or misaligned";
    case GCP_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
  }
}

size_t gcp_workspace_bytes(int64_t n) { return ws_bytes_for((i64)n); }

int gcp_workspace_init(void* ws, size_t ws_bytes, void* stream) {
  if (!ws || ws_bytes < (size_t)kWsHeaderBytes || ((uintptr_t)ws & 255u)) return GCP_ERR_WORKSPACE;
  GCP_HIP(hipMemsetAsync(ws, 0, ws_bytes, (hipStream_t)stream));  // launch counter 0, both descriptor sets clear
  return GCP_OK;
}

int gcp_cumprod_forward(const float* x, const int32_t* key, float* y, int64_t n, void* ws,
                        size_t ws_bytes, void* stream) {
  return launch_scan<M_CUMPROD_FWD>(x, nullptr, nullptr, key, y, n, ws, ws_bytes, stream);
}

int gcp_cumsum_forward(const float* x, const int32_t* key, float* y, int64_t n, void* ws,
                       size_t ws_bytes, void* stream) {
  return launch_scan<M_CUMSUM_FWD>(x, nullptr, nullptr, key, y, n, ws, ws_bytes, stream);
}

int gcp_cumsum_reverse(const float* x, const int32_t* key, float* y, int64_t n, void* ws,
                       size_t ws_bytes, void* stream) {
  return launch_scan<M_CUMSUM_REV>(x, nullptr, nullptr, key, y, n, ws, ws_bytes, stream);
}

int gcp_cumprod_forward_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                                void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && !index) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_permutation_impl(index, n, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMPROD_FWD>(x, nullptr, nullptr, sorted_key, y, n, ws, ws_bytes, stream, nullptr, index);
}

int gcp_cumsum_forward_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                               void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && !index) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_permutation_impl(index, n, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMSUM_FWD>(x, nullptr, nullptr, sorted_key, y, n, ws, ws_bytes, stream, nullptr, index);
}

int gcp_cumsum_reverse_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                               void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && !index) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_permutation_impl(index, n, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMSUM_REV>(x, nullptr, nullptr, sorted_key, y, n, ws, ws_bytes, stream, nullptr, index);
}

int gcp_cumprod_forward_carry(const float* x, const int32_t* inv, const float* carry, float* y, int64_t n,
                              int64_t n_groups, void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && (!carry || n_groups <= 0)) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_group_ids_impl(inv, n, n_groups, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMPROD_FWD>(x, nullptr, nullptr, inv, y, n, ws, ws_bytes, stream, carry);
}

int gcp_cumsum_forward_carry(const float* x, const int32_t* inv, const float* carry, float* y, int64_t n,
                             int64_t n_groups, void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && (!carry || n_groups <= 0)) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_group_ids_impl(inv, n, n_groups, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMSUM_FWD>(x, nullptr, nullptr, inv, y, n, ws, ws_bytes, stream, carry);
}

int gcp_cumsum_reverse_carry(const float* x, const int32_t* inv, const float* carry, float* y, int64_t n,
                             int64_t n_groups, void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && (!carry || n_groups <= 0)) return GCP_ERR_INVALID_ARGUMENT;
  if (validate_operands()) {
    const int st = check_group_ids_impl(inv, n, n_groups, nullptr, (hipStream_t)stream);
    if (st != GCP_OK) return st;
  }
  return launch_scan<M_CUMSUM_REV>(x, nullptr, nullptr, inv, y, n, ws, ws_bytes, stream, carry);
}

int gcp_cumprod_backward(const float* param, const float* param_cumprod, const float* grad_out,
                         const int32_t* inv, float* grad_in, const int32_t* inv_len, int64_t n,
                         int64_t n_groups, void* ws, size_t ws_bytes, void* stream) {
  if (n > 0 && (!inv_len || n_groups <= 0)) return GCP_ERR_INVALID_ARGUMENT;
  return launch_scan<M_CUMPROD_BWD>(param, param_cumprod, grad_out, inv, grad_in, n, ws, ws_bytes,
                                    stream);
}

int gcp_check_groups(const int32_t* inv, const int32_t* inv_len, int64_t n, int64_t n_groups,
                     int64_t* n_bad, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!n_bad || n < 0) return GCP_ERR_INVALID_ARGUMENT;
  *n_bad = 0;
  if (n == 0) return GCP_OK;
  if (!inv || !inv_len || n_groups <= 0) return GCP_ERR_INVALID_ARGUMENT;
  unsigned long long* d = nullptr;
  GCP_HIP(hipMalloc((void**)&d, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), stream);
  if (e == hipSuccess) {
    i64 blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gcp_check_groups_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, inv,
                       inv_len, (i64)n, (i64)n_groups, d);
    e = hipGetLastError();
  }
  unsigned long long h = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d);
  if (e != hipSuccess) return hip_fail(e);
  *n_bad = (int64_t)h;
  return GCP_OK;
}

int gcp_check_permutation(const int32_t* index, int64_t n, int64_t* n_bad, void* stream) {
  return check_permutation_impl(index, (i64)n, n_bad, (hipStream_t)stream);
}

int gcp_check_group_ids(const int32_t* inv, int64_t n, int64_t n_groups, int64_t* n_bad, void* stream) {
  return check_group_ids_impl(inv, (i64)n, (i64)n_groups, n_bad, (hipStream_t)stream);
}

int gcp_set_validate_operands(int on) {
  g_validate.store(on ? 1 : 0, std::memory_order_relaxed);
  return GCP_OK;
}

int gcp_tile_elems(void) { return kTile; }

int gcp_set_lookback_wait_us(int64_t us) {
  g_patience_us.store(us < 0 ? -1 : (long long)us, std::memory_order_relaxed);
  return GCP_OK;
}

int gcp_last_fallback_tiles(void* ws, void* stream_, int64_t* n_tiles) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!n_tiles) return GCP_ERR_INVALID_ARGUMENT;
  *n_tiles = 0;
  if (!ws) {
    int dev = 0;
    GCP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (dev < 0 || dev >= 64 || !g_ws[dev].ptr) return GCP_OK;
    ws = g_ws[dev].ptr;
  }
  unsigned h = 0;
  GCP_HIP(hipMemcpyAsync(&h, (const char*)ws + 4 * kHdrUnresolved, sizeof(h), hipMemcpyDeviceToHost, stream));
  GCP_HIP(hipStreamSynchronize(stream));
  *n_tiles = (int64_t)h;
  return GCP_OK;
}

int gcp_last_lookback_tiles(void* ws, void* stream_, int64_t* n_tiles) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!n_tiles) return GCP_ERR_INVALID_ARGUMENT;
  *n_tiles = 0;
  if (!ws) {
    int dev = 0;
    GCP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (dev < 0 || dev >= 64 || !g_ws[dev].ptr) return GCP_OK;
    ws = g_ws[dev].ptr;
  }
  unsigned h = 0;
  GCP_HIP(hipMemcpyAsync(&h, (const char*)ws + 4 * kHdrDescResolved, sizeof(h), hipMemcpyDeviceToHost, stream));
  GCP_HIP(hipStreamSynchronize(stream));
  *n_tiles = (int64_t)h;
  return GCP_OK;
}

}  // extern "C"
