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
 *   - outputs are caller-allocated and overwritten in place; nothing is
 *     retained by the library;
 *   - launches are asynchronous on `stream`; no host synchronisation, no
 *     allocation when a workspace is supplied (graph-capturable);
 *   - every function returns GCP_OK (0) or a GCP_ERR_* code; n == 0 is a no-op;
 *   - results are deterministic run to run (no value-carrying atomics).
 *
 * Workspace: the scans are single-pass for realistic inputs; groups longer than
 * one tile's look-back window use a per-tile descriptor fallback that needs
 * gcp_workspace_bytes(n) bytes of scratch.  Pass ws == NULL to let the library
 * use an internal per-device scratch buffer (grown with hipMalloc on demand —
 * not graph-capturable, and not safe for concurrent launches on two streams).
 * A caller-provided workspace must be prepared ONCE with gcp_workspace_init()
 * before its first use (and again after any failed call); it may then be
 * reused by any number of calls that are ordered on one stream.
 */
#ifndef GROUPED_CUMPROD_HIP_H
#define GROUPED_CUMPROD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCP_OK 0
#define GCP_ERR_INVALID_ARGUMENT 1 /* NULL pointer with n > 0, negative n, n too large */
#define GCP_ERR_WORKSPACE 2        /* workspace too small / misaligned */
#define GCP_ERR_HIP 3              /* a HIP call failed: see gcp_last_hip_error() */

#define GCP_ABI_VERSION 1

/* ABI version of the loaded library (== GCP_ABI_VERSION it was built with). */
int gcp_abi_version(void);

/* hipError_t (as int) of the most recent failing HIP call on this host thread, 0 if none. */
int gcp_last_hip_error(void);

/* Static string for a GCP_* return code. */
const char* gcp_status_string(int status);

/* Bytes of scratch needed for arrays of n elements (multiple of 256). */
size_t gcp_workspace_bytes(int64_t n);

/* Zero the control words of a caller-provided workspace (async on `stream`). */
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
 * Debug aid (synchronises `stream`): checks that `inv` is a dense
 * non-decreasing group id starting at 0 and that inv_len[g] is the exclusive
 * end offset of run g.  *n_bad receives the number of violations found.
 */
int gcp_check_groups(const int32_t* inv, const int32_t* inv_len, int64_t n,
                     int64_t n_groups, int64_t* n_bad, void* stream);

/* Tuning / introspection (used by bench.py and the tests). */
/* Elements per scan tile of the loaded build. */
int gcp_tile_elems(void);
/* Number of tiles the most recent scan on this host thread's workspace `ws`
 * could not resolve by look-back and fixed up through the descriptor path
 * (synchronises `stream`; ws == NULL selects the internal workspace). */
int gcp_last_fallback_tiles(void* ws, void* stream, int64_t* n_tiles);

#ifdef __cplusplus
}
#endif
#endif /* GROUPED_CUMPROD_HIP_H */
