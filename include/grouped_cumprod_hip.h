/*
 * grouped_cumprod_hip.h — C ABI of libgrouped_cumprod_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-pixel alpha-compositing scan of
 * TaiseiNiman/SimpleGaussianSplat_tk71.  Every entry point below replaces one
 * function the reference registers in its pybind11 module `grouped_cumprod`
 * (reference: cuda_kernel/cuda_kernel.cpp:5-22).  Plain pointers and sizes only:
 * no torch types, no C++ types.  All pointers are DEVICE pointers on the
 * current HIP device; `stream` is a hipStream_t passed as void* (NULL = the
 * legacy default stream, which is what the reference launches on,
 * cuda_kernel/grouped_cumprod_backward.cu:56).
 *
 * Conventions shared by all entry points
 *   - arrays are 1-D, contiguous, fp32 values / int32 keys (reference contract:
 *     data_ptr<float>() / data_ptr<int>(), grouped_cumprod_forward.cu:8-10);
 *   - a "group" (one pixel's depth-sorted splat list) is a maximal run of equal
 *     ADJACENT keys — keys need not be globally sorted
 *     (thrust::equal_to<int>, grouped_cumprod_forward.cu:21);
 *   - outputs are caller-allocated and overwritten; nothing is retained by the
 *     library.  The forward scans (gcp_cumprod_forward, gcp_cumsum_forward,
 *     gcp_cumsum_reverse) may run EXACTLY in place, y == x, as the reference's
 *     thrust::inclusive_scan_by_key may (grouped_cumprod_forward.cu:17-23): the
 *     library then takes every carry from its tile descriptors instead of
 *     re-reading the neighbouring tile's inputs (0.5 instead of 0.7 of the HBM
 *     roof).  Any other overlap of an output with an input of the same call —
 *     shifted views, the backward, the carry and indexed forms — returns
 *     GCP_ERR_INVALID_ARGUMENT (blocks re-read raw inputs of the neighbouring
 *     tile while that tile's block is already storing).
 *   - launches are asynchronous on `stream`; no host synchronisation, no
 *     allocation when a workspace is supplied (graph-capturable);
 *   - every function returns GCP_OK (0) or a GCP_ERR_* code; n == 0 is a no-op;
 *   - results are deterministic run to run, bit for bit, whatever the timing: no value-carrying atomic, and a tile
 *     that is finished by the follow-up launch (below) gets the bits it would have got inside the main launch.
 *
 * Workspace: the scans are single-pass.  A group longer than one tile's raw look-back window (4096 elements)
 * continues on per-tile descriptors inside the same launch; a tile that would have to wait too long for another
 * tile's descriptor is finished by a small follow-up launch (a no-op otherwise), which completes the descriptor
 * tree in the main launch's fixed association and re-runs the tile with the carry taken from it.  Both need
 * gcp_workspace_bytes(n) bytes of scratch (0.4 % of one array).  Pass ws == NULL to let the library use an internal
 * per-device scratch buffer (grown with hipMalloc on demand — not graph-capturable, and not safe for concurrent
 * launches on two streams).  A caller-provided workspace must be zeroed ONCE before its first use
 * (gcp_workspace_init, or any memset of the whole buffer): it carries a launch counter and two descriptor sets used
 * alternately, each launch clearing the set the next one will publish into.  After that it may be reused by any
 * number of calls of any size that fits, as long as they are ORDERED on one stream (or otherwise never overlap).
 */
#ifndef GROUPED_CUMPROD_HIP_H
#define GROUPED_CUMPROD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCP_OK 0
#define GCP_ERR_INVALID_ARGUMENT 1 /* NULL pointer with n > 0, negative n, n too large, output overlapping an input */
#define GCP_ERR_WORKSPACE 2        /* workspace too small / misaligned */
#define GCP_ERR_HIP 3              /* a HIP call failed: see gcp_last_hip_error() */

#define GCP_ABI_VERSION 4

/* ABI version of the loaded library (== GCP_ABI_VERSION it was built with). */
int gcp_abi_version(void);

/* Hash of the kernel sources + headers + compile flags this binary was built from (the build recipe passes it as
 * -DGCP_SOURCE_HASH); the Python binding refuses a library whose hash differs from the checkout's. */
const char* gcp_source_hash(void);

/* hipError_t (as int) of the most recent failing HIP call on this host thread, 0 if none. */
int gcp_last_hip_error(void);

/* Static string for a GCP_* return code. */
const char* gcp_status_string(int status);

/* Bytes of scratch needed for arrays of n elements (multiple of 256). */
size_t gcp_workspace_bytes(int64_t n);

/* Zero a caller-provided workspace (async on `stream`); required once before its first use. */
int gcp_workspace_init(void* ws, size_t ws_bytes, void* stream);

/*
 * y[i] = prod of x[k] over k <= i in the same key-run as i   (inclusive).
 * Replaces grouped_cumprod_forward(x, key, y)
 *   reference: cuda_kernel/cuda_kernel.cpp:5,18;
 *              cuda_kernel/grouped_cumprod_forward.cu:6-24
 *              (thrust::inclusive_scan_by_key, equal_to<int>, multiplies<float>).
 * Traffic: 12 B / element (x, key in; y out).
 */
int gcp_cumprod_forward(const float* x, const int32_t* key, float* y, int64_t n,
                        void* ws, size_t ws_bytes, void* stream);

/*
 * y[i] = sum of x[k] over k <= i in the same key-run as i   (inclusive).
 * Replaces grouped_cumsum_forward(x, key, y)
 *   reference: cuda_kernel/cuda_kernel.cpp:14,21;
 *              cuda_kernel/grouped_cumsum_forward.cu:6-24 (thrust::plus<float>).
 * Traffic: 12 B / element.
 */
int gcp_cumsum_forward(const float* x, const int32_t* key, float* y, int64_t n,
                       void* ws, size_t ws_bytes, void* stream);

/*
 * VJP of gcp_cumprod_forward:
 *   grad_in[j] = sum_{i=j}^{end(j)-1} grad_out[i] * param_cumprod[i] / p'_j,
 *   p'_j = param[j] != 0 ? param[j] : 1e-8f,   end(j) = inv_len[inv[j]].
 * Replaces grouped_cumprod_backward(param, param_cumprod, grad_out, inv,
 *                                   grad_in, inv_len)
 *   reference: cuda_kernel/cuda_kernel.cpp:6-13,19-20;
 *              cuda_kernel/grouped_cumprod_backward.cu:9-41 (kernel), :43-65.
 * `inv` is the dense group id 0..G-1 of every element (non-decreasing runs),
 * `inv_len[g]` the exclusive end offset of group g (cuda_test.py:27).  The two
 * must describe the same partition (what gs_model.py / cuda_test.py pass); the
 * group ends are taken from the runs of `inv`, `inv_len` is only validated by
 * gcp_check_groups().  The reference's O(sum L^2) per-element loop is replaced
 * by one O(n) reverse segmented scan.  Traffic: 20 B / element.
 */
int gcp_cumprod_backward(const float* param, const float* param_cumprod,
                         const float* grad_out, const int32_t* inv, float* grad_in,
                         const int32_t* inv_len, int64_t n, int64_t n_groups,
                         void* ws, size_t ws_bytes, void* stream);

/*
 * Suffix form of gcp_cumsum_forward: y[i] = sum of x[k] over k >= i in the same
 * key-run.  The reference obtains this by flipping its arrays around
 * grouped_cumsum_forward (gs_model.py:716-722); this entry point does the same
 * scan in place without the two flips.  Traffic: 12 B / element.
 */
int gcp_cumsum_reverse(const float* x, const int32_t* key, float* y, int64_t n,
                       void* ws, size_t ws_bytes, void* stream);

/*
 * INDEXED forms: the sort -> scan -> un-sort sandwich of _create_alpha_brend (reference: gs_model.py:548
 * `anti_opacity[index]`, :551/:553 the scan, :555 `output[torch.argsort(index)]`) in ONE pass.  `x` and `y` are in the
 * caller's ORIGINAL pair order, `sorted_key` / `index` are what gcp_sort_pairs_u32 / gcp_sort_rects returned
 * (sorted_key[i] = key of original element index[i]; `index` a permutation of 0..n-1):
 *   y[index[i]] = scan over i, inside runs of equal adjacent sorted_key, of x[index[i]].
 * Neither the sorted copy of the values nor the sorted result is materialised.  Same association, determinism and
 * workspace rules as the plain scans.  Traffic: 16 B / element (key 4, index 4, gathered value 4, scattered result 4).
 */
int gcp_cumprod_forward_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                                void* ws, size_t ws_bytes, void* stream);
int gcp_cumsum_forward_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                               void* ws, size_t ws_bytes, void* stream);
int gcp_cumsum_reverse_indexed(const float* x, const int32_t* sorted_key, const int32_t* index, float* y, int64_t n,
                               void* ws, size_t ws_bytes, void* stream);

/*
 * Exact chunk carry.  Same scans, but group g starts from carry[g]
 * instead of the identity: y[i] = carry[inv[i]] (*|+) scan.  `inv` must be the DENSE group id
 * (it indexes `carry`, f32[n_groups]).  A caller that splits every pixel's depth-sorted list
 * into memory chunks (reference: gs_model.py:428, :675-686 forward, :634-643 backward) passes
 * the previous chunk's last inclusive value per pixel and gets the single-chunk result
 * exactly; the reference's own carry drops one factor per chunk boundary (its `amin` of the
 * EXCLUSIVE transmittance, gs_model.py:582-586).  For the reverse form the carry
 * is the suffix sum entering from the deeper chunk.  Traffic: 12 B / element + 4 B / group.
 */
int gcp_cumprod_forward_carry(const float* x, const int32_t* inv, const float* carry, float* y,
                              int64_t n, int64_t n_groups, void* ws, size_t ws_bytes, void* stream);
int gcp_cumsum_forward_carry(const float* x, const int32_t* inv, const float* carry, float* y,
                             int64_t n, int64_t n_groups, void* ws, size_t ws_bytes, void* stream);
int gcp_cumsum_reverse_carry(const float* x, const int32_t* inv, const float* carry, float* y,
                             int64_t n, int64_t n_groups, void* ws, size_t ws_bytes, void* stream);

/*
 * Debug aid (synchronises `stream`): checks that `inv` is a dense
 * non-decreasing group id starting at 0 and that inv_len[g] is the exclusive
 * end offset of run g.  *n_bad receives the number of violations found.
 */
int gcp_check_groups(const int32_t* inv, const int32_t* inv_len, int64_t n,
                     int64_t n_groups, int64_t* n_bad, void* stream);

/*
 * Operand checks for the two forms that index memory through an operand (debug aids like gcp_check_groups: they
 * allocate a scratch buffer and synchronise `stream`).  The scans themselves trust these operands — an `index` entry
 * outside [0, n) or an `inv` entry outside [0, n_groups) is an out-of-bounds device access.
 *   gcp_check_permutation: GCP_OK if index[0..n) is a permutation of 0..n-1 (what the indexed scans need: they read
 *     x[index[i]] and write y[index[i]]), GCP_ERR_INVALID_ARGUMENT otherwise; *n_bad (may be NULL) receives the
 *     number of entries that are out of range or repeat an earlier value.
 *   gcp_check_group_ids: GCP_OK if every inv[i] lies in [0, n_groups) (what the carry forms index `carry` with),
 *     GCP_ERR_INVALID_ARGUMENT otherwise; *n_bad (may be NULL) = entries out of range.
 *   gcp_set_validate_operands(1) (or environment GCP_VALIDATE_OPERANDS=1), process-wide, default off: every
 *     gcp_cum*_indexed call first runs gcp_check_permutation on its `index`, every gcp_cum*_carry call
 *     gcp_check_group_ids on its `inv`, and returns GCP_ERR_INVALID_ARGUMENT WITHOUT launching the scan when the
 *     operand is bad — at the price of one extra pass and a host synchronisation per call (not graph-capturable).
 */
int gcp_check_permutation(const int32_t* index, int64_t n, int64_t* n_bad, void* stream);
int gcp_check_group_ids(const int32_t* inv, int64_t n, int64_t n_groups, int64_t* n_bad, void* stream);
int gcp_set_validate_operands(int on);

/* Tuning / introspection (used by bench.py and the tests). */
/* Elements per scan tile of the loaded build. */
int gcp_tile_elems(void);
/* Number of tiles the most recent scan on workspace `ws` could not resolve inside the main launch (their wait for
 * another tile's descriptor ran out) and that the follow-up launch fixed up
 * (synchronises `stream`; ws == NULL selects the internal workspace). */
int gcp_last_fallback_tiles(void* ws, void* stream, int64_t* n_tiles);
/* Longest time (microseconds) a tile waits for another tile's descriptor before it leaves its carry to the
 * follow-up launch; default 200 (environment GCP_DESC_WAIT_US).  0 = look once, never wait; negative = skip the
 * descriptor walk altogether (every group that starts more than one tile back goes to the follow-up launch — the
 * two-pass behaviour, kept for tests and A/B timing).  Process-wide; results are bit-identical for every setting
 * (tests/test_scan_gpu.py::test_results_do_not_depend_on_the_descriptor_wait). */
int gcp_set_lookback_wait_us(int64_t us);
/* Number of tiles of that scan whose group started more than one tile back and that resolved their carry through
 * the descriptor look-back inside the main launch. */
int gcp_last_lookback_tiles(void* ws, void* stream, int64_t* n_tiles);


/* ------------------------------------------------------------------------------------------
 * Rows f1 / f2 of SURVEY.md §8: everything the reference's autograd Function does AROUND its
 * scan (reference: gs_model.py:598-663 _forward_batch/_backward_batch, :666-692 forward,
 * :786-820 backward), restated tile-based so that no splat-pixel pair array is ever
 * materialised.  Gaussians are given in depth order (front to back = array order) with integer
 * INCLUSIVE pixel boxes start_xy/end_xy int32[N,2] (x,y) (uitility.py:336-366), float means
 * [N,2], precision matrices vinv f32[N,2,2], opacity f32[N], colour l_d f32[N,3].  Images are
 * f32[(H+1),(W+1),3] (gs_model.py:505).  Tiles are 16x16 pixels.
 * ------------------------------------------------------------------------------------------ */

/* Tile grid of a (height+1) x (width+1) image. */
int gcp_tile_grid(int32_t width, int32_t height, int32_t* tiles_x, int32_t* tiles_y);

/* Exclusive prefix sum of int32; out has n+1 entries (out[n] = total). */
size_t gcp_scan_i32_workspace_bytes(int64_t n);
int gcp_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, void* ws, size_t ws_bytes,
                           void* stream);

/* f2, step 1: tile_off[g] = exclusive prefix of the number of tiles box g touches
 * (tile_off has n_gauss+1 entries).  Synchronises `stream` once to return the total K on the
 * host (the reference synchronises likewise: uitility.py:348 `.item()`).
 * ws: gcp_bin_workspace_bytes(n_gauss, 0) bytes. */
int gcp_bin_tiles_count(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss,
                        int32_t width, int32_t height, int32_t* tile_off,
                        int64_t* n_tile_pairs_host, void* ws, size_t ws_bytes, void* stream);
size_t gcp_bin_workspace_bytes(int64_t n_gauss, int64_t n_tile_pairs);
/* f2, step 2: tile_list[K] = Gaussian ids grouped by tile, depth order inside each tile (stable
 * radix sort of the Gaussian-major (tile, gaussian) entries); tile_start[n_tiles+1] = first entry
 * of every tile.  Replaces torch.sort / argsort / unique over M pixel keys
 * (gs_model.py:546-555, :582-586). */
int gcp_bin_tiles_fill(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss,
                       int32_t width, int32_t height, const int32_t* tile_off,
                       int64_t n_tile_pairs, int32_t* tile_start, int32_t* tile_list, void* ws,
                       size_t ws_bytes, void* stream);

/* The same binning in ONE call without a host read-back (graph-capturable): the caller bounds the entry count by
 * `capacity` (tile_list has `capacity` slots, ws = gcp_bin_workspace_bytes(n_gauss, capacity)); the real count stays on
 * the device.  info[0] = entries listed, info[1] = 1 if the capacity was too small — Gaussians whose entries did not
 * fit are then left out (front-most first kept) and get zero gradients; the caller reads info once per step, after the
 * fact, instead of synchronising inside the Function (the reference synchronises on `.item()` calls per chunk,
 * gs_model.py:677,793,801-802).  Buffers derived from the bins (checkpoints, backward workspace) are sized by
 * `capacity`, and `capacity` is what gcp_blend_backward takes as n_tile_pairs. */
int gcp_bin_tiles(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                  int32_t height, int64_t capacity, int32_t* tile_off, int32_t* tile_start,
                  int32_t* tile_list, int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* f1 forward: image = sum over pairs of T * l * o * g with T the exclusive grouped cumprod of
 * (1 - o g) per pixel in depth order; pairs whose inclusive product is exactly 0 are dropped.
 * Replaces _forward_batch + index_put_(accumulate=True) (gs_model.py:598-624, :510-514).
 * t_ckpt: NULL (inference), or gcp_blend_checkpoint_floats(K, width, height) floats that receive every pixel's
 * transmittance at every GCP-internal checkpoint interval of its tile's list — what gcp_blend_backward restarts
 * its per-chunk front-to-back pass from (the reference keeps the whole M-length T array for that, gs_model.py:691). */
size_t gcp_blend_checkpoint_floats(int64_t n_tile_pairs, int32_t width, int32_t height);
int gcp_blend_forward(const int32_t* start_xy, const int32_t* end_xy, const float* mean_xy,
                      const float* vinv, const float* opacity, const float* l_d, int64_t n_gauss,
                      int32_t width, int32_t height, const int32_t* tile_start,
                      const int32_t* tile_list, float* image, float* t_ckpt, void* stream);

/* f1 backward: gradients of <image, grad_image> w.r.t. mean [N,2], vinv [N,2,2], opacity [N],
 * l_d [N,3].  `t_ckpt` is what gcp_blend_forward wrote for the same inputs and bins.  Replaces _backward_batch +
 * grad_list_to_gause (gs_model.py:627-663, :733-783).  Every tile list is walked BACK TO FRONT: the per-pixel
 * suffix sums the reference takes from a flipped grouped cumsum (gs_model.py:716-722) are carried as
 * S_k / (1 - o_k g_k) = T_k R_k with R_{k-1} = R_k + o_k g_k (dL/dI . l_k - R_k) (no cancellation, no division), so
 * a gradient's round-off is relative to the transmittance of its own layer at any depth.  grad_l is the TRUE
 * gradient (the reference's is channel-collapsed, gs_model.py:710-712,:763-766), and dL/dopacity is the true one
 * also at opacity == 0 (the reference drops the direct term there, gs_model.py:737-738).
 * ws: gcp_blend_backward_workspace_bytes(K). */
size_t gcp_blend_backward_workspace_bytes(int64_t n_tile_pairs);
int gcp_blend_backward(const int32_t* start_xy, const int32_t* end_xy, const float* mean_xy,
                       const float* vinv, const float* opacity, const float* l_d, int64_t n_gauss,
                       int32_t width, int32_t height, const int32_t* tile_off,
                       int64_t n_tile_pairs, const int32_t* tile_start, const int32_t* tile_list,
                       const float* t_ckpt, const float* grad_image, float* grad_mean,
                       float* grad_vinv, float* grad_opacity, float* grad_l, void* ws,
                       size_t ws_bytes, void* stream);

/* Stable sort of n uint32 keys that also returns the permutation: keys_out[i] = keys_in[index_out[i]], equal
 * keys keep their input order — what the reference asks of torch.sort for its pixel keys (gs_model.py:546-547;
 * depth order inside a pixel rides on that stability).  LSD radix sort over the low `key_bits` bits (8 per
 * pass; the keys must be < 2^key_bits), deterministic (no atomic decides a slot).  n < 2^31.
 * ws: gcp_sort_workspace_bytes(n) bytes. */
size_t gcp_sort_workspace_bytes(int64_t n);
int gcp_sort_pairs_u32(const uint32_t* keys_in, int64_t n, int32_t key_bits, uint32_t* keys_out,
                       int32_t* index_out, void* ws, size_t ws_bytes, void* stream);
/* The same sort with the reference's pixel key computed on the fly from its rect list (gs_model.py:538-541, :546:
 * key = y * 10000 + x of rects_xy[i] = (x, y)): the int32 key array of `unique()` never exists.  keys_out receives
 * the sorted keys (what `sorted_inv` is at gs_model.py:547), index_out the permutation.
 *   id_width == 0: key_bits must cover max(y) * 10000 + max(x) (24 at 1920x1080, 25 at 3840x2160);
 *   id_width  > 0: the caller knows the image: every x < id_width (= width + 1).  The passes then run on the compact
 *     ids y * id_width + x (same order, fewer bits, narrower digits: 3 x 7 bits at 1920x1080) and the last pass writes
 *     the reference's keys; key_bits must cover max(y) * id_width + max(x) and be <= 24. */
int gcp_sort_rects(const int32_t* rects_xy, int64_t n, int32_t key_bits, int32_t id_width, uint32_t* keys_out,
                   int32_t* index_out, void* ws, size_t ws_bytes, void* stream);
/* max over i of (y_i * 10000 + x_i) and min over i of min(x_i, y_i), written to out_dev[0], out_dev[1] (device int32[2]):
 * what a caller needs to choose key_bits when it does not know the image size. */
int gcp_rects_key_range(const int32_t* rects_xy, int64_t n, int32_t* out_dev, void* stream);

/* Rows a5 / a6 for callers that still hold the boxes their rect list was expanded from (gs_model.py:601 -> :607): key,
 * stable sort, gather, grouped scan and un-sort of _create_alpha_brend (gs_model.py:546-555) as ONE walk of the tile lists of
 * gcp_bin_tiles*.  x / inclusive: f32[M] in the reference's Gaussian-major rect order (uitility.py:336-366); box_off:
 * int32[n_gauss + 1] exclusive prefix sums of the clamped box sizes (gcp_box_sizes + gcp_exclusive_scan_i32).
 *   inclusive[pair] = product (mode 0) / sum (mode 1) of x over the pairs of the same pixel up to and including this one
 *   in depth order; mode 2: sum from this one to the deepest (grad_cumsum's flipped scan, gs_model.py:716-722).
 * Every pixel is scanned sequentially in depth order (the association of the CPU statement).  8 B / pair.  Feed the result
 * to gcp_compact_finish.  n_pairs = M = box_off[n_gauss], the length of x and inclusive (< 2^31; up to 2^30 the kernel
 * addresses pairs by 32-bit byte offsets).  width + 1 must stay below 2^22 (2^24 beyond 2^30 pairs): a pair's position is
 * formed with one 24-bit multiply by the box width; GCP_ERR_INVALID_ARGUMENT otherwise.  dropped_per_tile: NULL, or int32[ceil(n_pairs / 4096)] that receives, for
 * every 4096 consecutive pairs, how many inclusive values are exactly 0 (integer adds, one per wave and list entry that
 * has any: deterministic) — the per-tile counts gcp_compact_finish would otherwise read the array once more for. */
int gcp_pairs_scan_boxes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                         const int32_t* tile_start, const int32_t* tile_list, const int32_t* box_off, const float* x,
                         float* inclusive, int64_t n_pairs, int32_t mode, int32_t* dropped_per_tile, void* stream);

/* The walk writing the FINAL values of _create_alpha_brend (gs_model.py:557-564), so that a list that drops nothing needs no
 * compaction pass at all.  Same walk as gcp_pairs_scan_boxes, same arguments, but every pair receives
 *   values[pair] = inclusive / x[pair] (mode 0, gs_model.py:562)  or  inclusive - x[pair] (modes 1, 2, :564)
 * — the fp32 operation gcp_compact_finish would apply to the stored inclusive value: the same bits — and
 *   keep[pair]   = (inclusive != 0)                                  (gs_model.py:560, :575-578)
 * as one byte per pair (the call sets every byte to 1 and the walk clears the bytes of the pairs it drops);
 * dropped_per_tile (required, int32[ceil(n_pairs / 4096)]) counts them per 4096 consecutive pairs.  prepared != 0: the
 * caller has already run gcp_pairs_finish_prepare(keep, dropped_per_tile, n_pairs, stream) on the same stream — the two
 * fills, which depend on nothing but n_pairs, can then be queued BEFORE the host waits for gcp_rects_cut's info8 and run
 * while it does.  When
 * gcp_compact_kept_count reports that everything was kept, `values` and `keep` ARE the result of _create_alpha_brend
 * (8 B per pair + the 1 B fill of the mask instead of 21); otherwise gcp_compact_kept_write moves the kept values together
 * (5 B read + 4 B written per pair). */
int gcp_pairs_finish_prepare(uint8_t* keep, int32_t* dropped_per_tile, int64_t n_pairs, void* stream);
int gcp_pairs_finish_boxes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width, int32_t height,
                           const int32_t* tile_start, const int32_t* tile_list, const int32_t* box_off, const float* x,
                           float* values, uint8_t* keep, int64_t n_pairs, int32_t mode, int32_t* dropped_per_tile, int32_t prepared,
                           void* stream);

/* The stream compaction that remains after gcp_pairs_finish_boxes, split at the one device->host read that sizes the result
 * (the reference's `output[mask]`, gs_model.py:575-578, synchronises likewise):
 *   gcp_compact_kept_count: count_dev[0] (device int32) = number of set bytes of keep[begin, end) — the rows a
 *     `cutting_number` slice leaves (gs_model.py:557-559) — and, in ws, the kept count of every 4096-element tile of the range
 *     (ONE launch when the walk's counts serve; the prefix sums are made by gcp_compact_kept_write, only when needed).
 *     dropped_per_tile (may be NULL; what the walk counted for the WHOLE array of n_total elements) is used instead of
 *     reading the keep bytes when the range starts at a multiple of 4096 and ends at one or at n_total.
 *   gcp_compact_kept_write: values_out[k] = values_in[i] for the k-th kept i of [begin, end), in order; `ws` as the count
 *     call left it, for the same range.  Not needed when the count equals end - begin.
 * Stable, no atomics, deterministic.  ws: gcp_compact_kept_workspace_bytes(end - begin). */
size_t gcp_compact_kept_workspace_bytes(int64_t n);
int gcp_compact_kept_count(const uint8_t* keep, const int32_t* dropped_per_tile, int64_t n_total, int64_t begin, int64_t end,
                           int32_t* count_dev, void* ws, size_t ws_bytes, void* stream);
int gcp_compact_kept_write(const float* values_in, const uint8_t* keep, int64_t begin, int64_t end, float* values_out, void* ws,
                           size_t ws_bytes, void* stream);

/* The rect list of the reference cut back into rectangles (gcp_pairs.hip), so that _create_alpha_brend / grad_cumsum can
 * take the tile-walk route (gcp_pairs_scan_boxes) from nothing but `rects` — which `_create_rects` always writes as a
 * concatenation of row-major boxes (gs_model.py:480-482, uitility.py:336-366).  Valid for ANY list: a list that is not made
 * of boxes yields about as many rectangles as elements, and the caller then sorts instead.
 *   gcp_rects_rows: rows = maximal runs (x, y), (x+1, y), ...  row_start[k] = index of the k-th row's first element,
 *     row_start[rows] = n, row_xy[k] = its (x, y); both have room for row_capacity entries.  info (device int32[5]) =
 *     {rows, max x, max y, min coordinate, not_boxes}; not_boxes != 0 — bit 1: the list has more than row_capacity - 1 rows;
 *     bit 0: an x >= 10000 (the reference's key y * 10000 + x then merges different pixels, gs_model.py:538-541: only the
 *     key-based sort route reproduces its groups) or a y >= 2^17 — nothing was written, sort instead.  gcp_rects_rows_capacity(n) = n / 2 + 2
 *     is what a list of boxes is allowed; a caller whose list starts or ends with c single-pixel carry rows (the
 *     `cutting_number` rows of gs_model.py:611, :636 — lexicographically sorted unique pixels: they come out as one-pixel-wide
 *     rectangles) passes c + gcp_rects_rows_capacity(n - c).
 *   gcp_rows_rectangles: rectangles = maximal runs of rows with equal first x and length and y growing by one.
 *     rect_row[b] = first row of rectangle b (room for n_rows + 1 entries; rect_row[rectangles] = n_rows).  info (device
 *     int32[2]) = {rectangles, 0}.
 *   gcp_rectangle_boxes: start_xy / end_xy (inclusive, int32[n_rects][2]) and box_off (int32[n_rects + 1], box_off[b] =
 *     index of the rectangle's first pair, box_off[n_rects] = n): the arguments of gcp_bin_tiles* and gcp_pairs_scan_boxes.
 * Each cut is a stable stream compaction without atomics on the data path (per-tile counts, one exclusive scan, ranked
 * writes); the first reads the M-sized list once.  ws: the matching *_workspace_bytes. */
/* The three cuts AND the binning's counting pass in ONE call with nothing read back in between (gcp_rects_cut): what
 * _create_alpha_brend / grad_cumsum run by default.  Every stage takes its predecessor's count from device memory; the caller
 * reads info8 once:
 *   info8 (device int32[8]) = {rows, max x, max y, min coordinate, flags, rectangles, K, 0}
 *   flags: 0 = start_xy / end_xy / box_off (as gcp_rectangle_boxes) and tile_off (as gcp_bin_tiles_count: exclusive prefix
 *     sums of the tiles every rectangle touches, tile_off[rectangles] = K — the image being [0, max x] x [0, max y], the
 *     binning's clamp is the identity) are valid: hand them to gcp_bin_tiles_fill(…, max x, max y, tile_off, K, …).
 *     1 = a coordinate the walk cannot take (x >= 10000, y >= 2^17: see gcp_rects_rows) — sort instead;
 *     2 = more rows than there is room for (slot_rows per 4096-pair tile on average; tiles that need more draw on a shared
 *       pool of n / 32 records) — take the step-by-step cut (gcp_rects_rows: room for a row per two pairs) or sort;
 *     4 = more than rect_capacity rectangles; 8 = K does not fit int32.  A negative min coordinate: refuse the list.
 *   slot_rows: row records a tile of boxes may park (<= 4096; 512 serves boxes of 8 columns and more).  carry_front /
 *     carry_back: the list starts / ends with that many single-pixel carry rows (the `cutting_number` rows of gs_model.py:611,
 *     :636): their tiles get one slot per element.  rect_capacity: rectangles start_xy / end_xy (int32[cap][2]), box_off and
 *     tile_off (int32[cap + 1]) have room for.
 * Scratch (ws, 256-byte aligned): 8 B x slot_rows per tile for the parked records, as much again for the rows (kept
 * packed), the pool and a few words per tile: 2.3 B per pair at slot_rows = 512 (+ 28 B per rectangle of capacity in the
 * caller's arrays: 0.45 B per pair at 64 pairs per rectangle), against gcp_rects_rows' 8 B + the caller's 6 B per pair. */
size_t gcp_rects_cut_workspace_bytes(int64_t n, int64_t carry_front, int64_t carry_back, int32_t slot_rows, int64_t rect_capacity);
int gcp_rects_cut(const void* rects_xy, int32_t rects_are_int64, int64_t n, int64_t carry_front, int64_t carry_back, int32_t slot_rows,
                  int64_t rect_capacity, int32_t* start_xy, int32_t* end_xy, int32_t* box_off, int32_t* tile_off, int32_t* info8,
                  void* ws, size_t ws_bytes, void* stream);

/* ---- the per-pixel carry of the reference's chunked calls ----------------------------------
 * `_create_alpha_brend_min(rects, T)` (gs_model.py:582-586; every forward chunk, :609 / :615) =
 * torch.unique(rects, dim=0) + scatter_reduce(amin) over its inverse, and `create_grad_alphabrend_min(rects, grad)`
 * (:724-730; every backward chunk, :639 / :643) = the same with the pair's own index as the value.
 * gcp_pixels_min: for every distinct pixel of the list — in (x, y) ascending order, the row order of
 * torch.unique(dim=0) — its coordinates (out_xy, [capacity][2] of the element width of rects_xy) and the minimum of
 * `values` over its pairs (out_val; a NaN among them gives NaN, as amin does); with values = NULL, the index of its
 * FIRST pair, as the float the reference carries it in (gs_model.py:728: exact below 2^24, rounded to nearest-even
 * above).  One pass over the list (integer minima into an image-sized table: the result does not depend on the order
 * the pairs arrive in), then the table is read out; nothing M-sized is written or sorted.
 *   width, height: every coordinate must lie in [0, width] x [0, height] (the reference's (H+1) x (W+1) image,
 *   gs_model.py:505); info (device int32[4]) = {distinct pixels, 1 if a coordinate lay outside (those pairs are
 *   skipped), 0, 0}.  Rows beyond `capacity` are counted but not written ((width + 1) * (height + 1) always suffices).
 *   ws: gcp_pixels_min_workspace_bytes(width, height) (0: image too large), 256-byte aligned.
 * gcp_pixels_range: out3 (device int32[3]) = {max x, max y, min coordinate} of a list, for callers that do not hold
 * the image size. */
int gcp_pixels_range(const void* rects_xy, int32_t rects_are_int64, int64_t n, int32_t* out3, void* stream);
size_t gcp_pixels_min_workspace_bytes(int32_t width, int32_t height);
int gcp_pixels_min(const void* rects_xy, int32_t rects_are_int64, const float* values /* NULL: the pair's index */, int64_t n,
                   int32_t width, int32_t height, void* out_xy, float* out_val, int64_t capacity, int32_t* info, void* ws,
                   size_t ws_bytes, void* stream);

size_t gcp_rects_rows_workspace_bytes(int64_t n);
int64_t gcp_rects_rows_capacity(int64_t n);
int gcp_rects_rows(const int32_t* rects_xy, int64_t n, int64_t row_capacity, int32_t* row_start, int32_t* row_xy, int32_t* info,
                   void* ws, size_t ws_bytes, void* stream);
/* The same on a list of int64 coordinates — the dtype the reference's own make_rect_points_parallel returns
 * (uitility.py:336-366: `start + ix`, ix from torch.arange): read where it lies instead of being narrowed by a pass of its
 * own.  A coordinate outside [0, 2^31) is reported as a negative minimum (info[3] < 0). */
int gcp_rects_rows_i64(const int64_t* rects_xy, int64_t n, int64_t row_capacity, int32_t* row_start, int32_t* row_xy, int32_t* info,
                       void* ws, size_t ws_bytes, void* stream);
size_t gcp_rows_rectangles_workspace_bytes(int64_t n_rows);
int gcp_rows_rectangles(const int32_t* row_start, const int32_t* row_xy, int64_t n_rows, int32_t* rect_row, int32_t* info, void* ws,
                        size_t ws_bytes, void* stream);
int gcp_rectangle_boxes(const int32_t* rect_row, const int32_t* row_start, const int32_t* row_xy, int64_t n_rects, int64_t n,
                        int32_t* start_xy, int32_t* end_xy, int32_t* box_off, void* stream);

/* The tail of _create_alpha_brend (gs_model.py:557-564) on the un-sorted inclusive scan values, elements [begin, end)
 * of the original pair order (the reference's `cutting_number` slices, :557-559):
 *   keep[j]   = inclusive[begin + j] != 0                                   (:560, :575-578)
 *   values[k] = inclusive[i] / self[i] (mode 0, :562) or inclusive[i] - self[i] (mode 1, :564)
 *               for the k-th kept i, in order                               (the boolean-mask compaction)
 * count_dev[0] (device int32) receives the number of kept elements.  Stable stream compaction in two launches (per-tile
 * counts, then ranks from one exclusive scan of them): no atomics, deterministic.  values needs room for end - begin
 * floats.  ws: gcp_compact_workspace_bytes(end - begin).  Traffic: 17 B / element.
 * dropped_per_tile: NULL, or what gcp_pairs_scan_boxes counted for THIS inclusive array (int32 per 4096 elements of
 * the whole array, how many are exactly 0): the counting launch is then skipped (13 B / element); begin must be a
 * multiple of 4096. */
size_t gcp_compact_workspace_bytes(int64_t n);
int gcp_compact_finish(const float* inclusive, const float* self, int64_t begin, int64_t end, int32_t mode, float* values,
                       uint8_t* keep, int32_t* count_dev, const int32_t* dropped_per_tile, void* ws, size_t ws_bytes, void* stream);

/* The index plumbing of _create_alpha_brend around the scan, with the int32 permutation of gcp_sort_pairs_u32:
 * gcp_gather_f32: dst[i] = src[index[i]] (gs_model.py:548); gcp_unsort_finish: full[index[i]] = inclusive[i] / x_i
 * (mode 0, gs_model.py:562) or inclusive[i] - x_i (mode 1, :564) with x_i = sorted_x[i], keep[index[i]] =
 * (inclusive[i] != 0) (gs_model.py:560) — the un-sort (:555) fused with the steps that follow the compaction. */
int gcp_gather_f32(const float* src, const int32_t* index, float* dst, int64_t n, void* stream);
int gcp_unsort_finish(const float* inclusive, const float* sorted_x, const int32_t* index, float* full,
                      uint8_t* keep, int64_t n, int32_t mode, void* stream);

/* The reference's Gaussian-major rect list (Utilities.make_rect_points_parallel, uitility.py:336-366;
 * _create_rects, gs_model.py:480-482) for callers that still want it: box sizes (clamped to the image), then —
 * given their exclusive prefix sum box_off[N+1] — rects_xy int32[M,2] and optionally the owning Gaussian of
 * every pair (gause_points_inv, gs_model.py:768-773). */
int gcp_box_sizes(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss, int32_t width,
                  int32_t height, int32_t* box_size, void* stream);
int gcp_expand_rects(const int32_t* start_xy, const int32_t* end_xy, const int32_t* box_off,
                     int64_t n_gauss, int64_t n_pairs, int32_t width, int32_t height,
                     int32_t* rects_xy, int32_t* pair_gauss /* may be NULL */, void* stream);

/* f2, CSR export for callers of the scan API: per-pixel pair counts (row-major over the
 * (H+1)x(W+1) image = ascending pixel key y*10000+x) and box sizes; then, given their exclusive
 * prefix sums, pair_gauss[M] (Gaussian of every pair, pixel-major, depth order) and
 * pair_index[M] (the pair's position in the reference's Gaussian-major rect list) — i.e. the
 * `index` torch.sort(stable) returns at gs_model.py:547, bit for bit. */
int gcp_pixel_lists_count(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss,
                          int32_t width, int32_t height, const int32_t* tile_start,
                          const int32_t* tile_list, int32_t* pixel_count, int32_t* box_size,
                          void* stream);
int gcp_pixel_lists_fill(const int32_t* start_xy, const int32_t* end_xy, int64_t n_gauss,
                         int32_t width, int32_t height, const int32_t* tile_start,
                         const int32_t* tile_list, const int32_t* pixel_off,
                         const int32_t* box_off, int32_t* pair_gauss, int32_t* pair_index,
                         int32_t* pair_key /* optional: y*10000+x of every pair, may be NULL */,
                         void* stream);

/* ---- the caller's camera projection, fused ----------------------------------------------
 * One camera of GS_model_with_param.forward up to the Function call (reference: gs_model.py:289-365, :404-425;
 * helpers uitility.py:231-287, :431-462): world->camera, pinhole projection, pixel covariance J W S W^T J^T + 1e-6 I,
 * 3-sigma box from its eigen-decomposition, its inverse, SH colour (degree <= 2, coefficients [n_basis][3] per
 * Gaussian), sigmoid opacity, the cull test and the clamped integer box.
 *   cam_P float[12] = [R|t] row major, cam_K float[9] row major, both in device memory;
 *   box_clamp = the float the 3-sigma half extents are clamped to before truncation (gs_model.py:364-365).
 * gcp_project_forward writes, in the Gaussians' own order: record float[n_gauss][16] (16-byte aligned; opaque, read
 * back by gcp_project_gather), sort_key int32 (bit pattern of the positive camera depth, 0x7fffffff for culled
 * Gaussians: a stable ascending sort of the keys IS the depth order of gs_model.py:356 with the kept ones first),
 * keep uint8 (the cull mask of gs_model.py:405-407) and row_of = -1.
 * gcp_project_gather: for the first n_kept entries of the sorted permutation `perm`, the Function's arguments in
 * depth order (boxes, pixel means, boxsize = gs_model.py:425, Sigma'^-1 [4], opacity, l_d [3], index = Gaussian id)
 * and row_of[index[r]] = r.  With `keep` (the mask gcp_project_forward wrote) and n_kept = n_gauss, the list holds ALL
 * Gaussians without the kept count ever being read back: the culled ones follow the kept ones with an empty box (binned
 * into no tile, blended nowhere, zero gradients) — the capture-safe form; keep = NULL: the first n_kept entries only.
 * gcp_project_backward: gradients of (mean, quaternion, log scale, opacity logit, SH coefficients), n_gauss rows
 * each, all rows written (zeros where row_of < 0), from those of (Sigma'^-1, opacity, l_d) in list order. */
int gcp_project_forward(const float* mean, const float* quat_xyzw, const float* log_scale, const float* opacity_logit,
                        const float* sh_coeff, const float* cam_P, const float* cam_K, int64_t n_gauss, int32_t sh_degree,
                        int32_t n_basis, int32_t width, int32_t height, float box_clamp, float* record, int32_t* sort_key,
                        uint8_t* keep, int32_t* row_of, void* stream);
int gcp_project_gather(const float* record, const int32_t* perm, int64_t n_kept, int32_t* start_xy, int32_t* end_xy,
                       int32_t* mean_xy, int64_t* boxsize, float* vinv, float* alpha, float* l_d, int64_t* index,
                       int32_t* row_of, const uint8_t* keep /* may be NULL */, void* stream);
int gcp_project_backward(const float* mean, const float* quat_xyzw, const float* log_scale, const float* opacity_logit,
                         const float* sh_coeff, const float* cam_P, const float* cam_K, int64_t n_gauss, int32_t sh_degree,
                         int32_t n_basis, const int32_t* row_of, const float* grad_vinv, const float* grad_alpha,
                         const float* grad_l_d, float* grad_mean, float* grad_quat, float* grad_log_scale,
                         float* grad_opacity_logit, float* grad_sh_coeff, void* stream);

/* ---- the caller's training loss, fused ---------------------------------------------------
 * (1 - lambda) * mean|a - b| + lambda * (1 - mean SSIM(a, b)) of gs_control.py:180-182 (kornia.metrics.ssim with an
 * 11-tap Gaussian window and reflect padding), over `planes` = batch * channels planes of height x width floats.
 *   window11_host: the 11 normalised window weights, HOST memory (passed to the kernel by value); c1, c2 = (0.01 L)^2,
 *   (0.03 L)^2.
 * gcp_ssim_l1_forward writes partial[2 * b] = sum of the SSIM map and partial[2 * b + 1] = sum of |a - b| over block
 * b of gcp_ssim_blocks() (no atomics: the caller adds them, in any fixed order, for a reproducible loss) and, when
 * the three map pointers are non-NULL, dSSIM/dmu1, dSSIM/dE[a^2], dSSIM/dE[ab] per pixel for the backward.
 * gcp_ssim_l1_backward: grad_img1 = scales[0] * dSum(SSIM)/da + scales[1] * sign(a - b), scales in DEVICE memory
 * (so that an upstream gradient needs no host read); the adjoint of the reflect-padded blur is exact. */
int64_t gcp_ssim_blocks(int64_t planes, int32_t height, int32_t width);
int gcp_ssim_l1_forward(const float* img1, const float* img2, int64_t planes, int32_t height, int32_t width,
                        const float* window11_host, float c1, float c2, float* dm_dmu1, float* dm_de11, float* dm_de12,
                        float* partial, void* stream);
int gcp_ssim_l1_backward(const float* img1, const float* img2, const float* dm_dmu1, const float* dm_de11, const float* dm_de12,
                         int64_t planes, int32_t height, int32_t width, const float* window11_host, const float* scales,
                         float* grad_img1, void* stream);

/* ---- the caller's optimiser step ---------------------------------------------------------
 * torch.optim.Adam's update (no weight decay, no amsgrad; gs_model.py:43-47, :64-67) of one parameter tensor of n
 * floats, in place: exp_avg = beta1 exp_avg + (1 - beta1) grad; exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) grad^2;
 * param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^step) + eps).  step counts from 1.
 * Hyper-parameters are doubles (1 - beta and the bias corrections are formed in double, as torch does, then rounded
 * once).  All four arrays 16-byte aligned. */
int gcp_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, int64_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GROUPED_CUMPROD_HIP_H */
