/*
 * gcp_oracle.c — CPU restatement of the reference's grouped scan kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * simplegaussiansplat_tk71_amd, grouped_cumprod.py, cuda_kernel.py) may import,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * Parity pin: checked against
 *   - the reference's own known-answer vectors (cuda_test.py:19-34 and the
 *     worked example in uitility.py:383-393), tests/test_oracle.py;
 *   - the reference's cuda_kernel/grouped_cum{prod,sum}_forward.cu compiled
 *     unmodified for the host (rocThrust CPP backend) into oracle/_ref/, both
 *     live (tests/test_oracle.py::test_oracle_vs_live_reference_host_build,
 *     only where oracle/_ref/ exists)
 *     and through committed golden vectors (tests/golden/).
 *
 * Every function is strictly sequential, left to right, in fp32 — the
 * strictest CPU statement of the algorithm; compile with -ffp-contract=off.
 */
#include <stddef.h>
#include <stdint.h>

/*
 * reference: cuda_kernel/grouped_cumprod_forward.cu:17-23
 *   thrust::inclusive_scan_by_key(key, key+n, x, y, equal_to<int>, multiplies<float>)
 * A run is a maximal sequence of equal ADJACENT keys (predicate applied to
 * neighbours, :21); the running value restarts at every run.
 */
void oracle_cumprod_forward(const float* x, const int32_t* key, float* y, int64_t n) {
  if (n <= 0) return;
  float acc = x[0];
  y[0] = acc;
  for (int64_t i = 1; i < n; ++i) {
    acc = (key[i] == key[i - 1]) ? acc * x[i] : x[i];
    y[i] = acc;
  }
}

/* reference: cuda_kernel/grouped_cumsum_forward.cu:17-23 (thrust::plus<float>, :22) */
void oracle_cumsum_forward(const float* x, const int32_t* key, float* y, int64_t n) {
  if (n <= 0) return;
  float acc = x[0];
  y[0] = acc;
  for (int64_t i = 1; i < n; ++i) {
    acc = (key[i] == key[i - 1]) ? acc + x[i] : x[i];
    y[i] = acc;
  }
}

/*
 * reference: cuda_kernel/grouped_cumprod_backward.cu:18-29, one "thread" per idx:
 *   gid = inv[idx]; i_max = inv_len[gid];
 *   param_idx = param[idx] != 0 ? param[idx] : 1e-8f;
 *   val = 0; for (i = idx; i < i_max; i++) val += grad_out[i] * (param_cumprod[i] / param_idx);
 *   grad_in[idx] = val;
 * Same association (divide, multiply, then add in ascending i), same fp32.
 * O(sum L^2): use on small/medium inputs only.
 */
void oracle_cumprod_backward(const float* param, const float* param_cumprod, const float* grad_out,
                             const int32_t* inv, float* grad_in, const int32_t* inv_len,
                             int64_t n) {
  for (int64_t idx = 0; idx < n; ++idx) {
    const int32_t gid = inv[idx];
    const int64_t i_max = inv_len[gid];
    float val = 0.0f;
    const float param_idx = (param[idx] != 0.0f) ? param[idx] : 1e-8f;
    for (int64_t i = idx; i < i_max; ++i) {
      val += grad_out[i] * (param_cumprod[i] / param_idx);
    }
    grad_in[idx] = val;
  }
}

/*
 * Same quantity with double accumulation and O(n) cost (suffix sum of
 * grad_out*cumprod inside each group, then one division): the "true value"
 * used to bound the fp32 error of both the reference order and the GPU order.
 */
void oracle_cumprod_backward_f64(const float* param, const float* param_cumprod,
                                 const float* grad_out, const int32_t* inv, double* grad_in,
                                 int64_t n) {
  double acc = 0.0;
  for (int64_t i = n - 1; i >= 0; --i) {
    if (i == n - 1 || inv[i] != inv[i + 1]) acc = 0.0;
    acc += (double)grad_out[i] * (double)param_cumprod[i];
    const double p = (param[i] != 0.0f) ? (double)param[i] : (double)1e-8f;
    grad_in[i] = acc / p;
  }
}

/* fp64 forward scans (error-bound references). */
void oracle_cumprod_forward_f64(const float* x, const int32_t* key, double* y, int64_t n) {
  if (n <= 0) return;
  double acc = x[0];
  y[0] = acc;
  for (int64_t i = 1; i < n; ++i) {
    acc = (key[i] == key[i - 1]) ? acc * (double)x[i] : (double)x[i];
    y[i] = acc;
  }
}

void oracle_cumsum_forward_f64(const float* x, const int32_t* key, double* y, int64_t n) {
  if (n <= 0) return;
  double acc = x[0];
  y[0] = acc;
  for (int64_t i = 1; i < n; ++i) {
    acc = (key[i] == key[i - 1]) ? acc + (double)x[i] : (double)x[i];
    y[i] = acc;
  }
}

/*
 * Suffix form of the grouped cumsum: what the reference obtains with
 * flip -> grouped_cumsum_forward -> flip (gs_model.py:716-722).  Sequential
 * from the right, fp32.
 */
void oracle_cumsum_reverse(const float* x, const int32_t* key, float* y, int64_t n) {
  if (n <= 0) return;
  float acc = x[n - 1];
  y[n - 1] = acc;
  for (int64_t i = n - 2; i >= 0; --i) {
    acc = (key[i] == key[i + 1]) ? acc + x[i] : x[i];
    y[i] = acc;
  }
}

/*
 * Multi-threaded forms for the CPU baseline of bench.py (OpenMP over groups / elements; every group is still
 * scanned sequentially left to right in fp32, so the results are bit-identical to the functions above).
 * `group_end` = exclusive end offset per group (the reference's inv_len, cuda_test.py:27).
 */
#ifdef _OPENMP
#include <omp.h>
int oracle_max_threads(void) { return omp_get_max_threads(); }
#else
int oracle_max_threads(void) { return 1; }
#endif

void oracle_cumprod_forward_mt(const float* x, const int32_t* group_end, float* y, int64_t n_groups, int threads) {
  (void)threads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(threads)
  for (int64_t g = 0; g < n_groups; ++g) {
    const int64_t b = g ? group_end[g - 1] : 0, e = group_end[g];
    float acc = 1.0f;
    for (int64_t i = b; i < e; ++i) {
      acc = (i == b) ? x[i] : acc * x[i];
      y[i] = acc;
    }
  }
}

void oracle_cumprod_backward_mt(const float* param, const float* param_cumprod, const float* grad_out,
                                const int32_t* inv, float* grad_in, const int32_t* inv_len, int64_t n, int threads) {
  (void)threads;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(threads)
  for (int64_t idx = 0; idx < n; ++idx) {  /* reference: grouped_cumprod_backward.cu:18-29, one "thread" per idx */
    const int32_t gid = inv[idx];
    const int64_t i_max = inv_len[gid];
    float val = 0.0f;
    const float param_idx = (param[idx] != 0.0f) ? param[idx] : 1e-8f;
    for (int64_t i = idx; i < i_max; ++i) val += grad_out[i] * (param_cumprod[i] / param_idx);
    grad_in[idx] = val;
  }
}
