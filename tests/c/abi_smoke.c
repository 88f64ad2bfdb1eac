/* abi_smoke.c — a plain C program on the C ABI (no Python, no torch, no C++): what a host written in
 * another language would do through its FFI.  Builds with gcc against include/grouped_cumprod_hip.h and
 * libgrouped_cumprod_hip.so; device memory comes from the HIP runtime's C API.
 * Runs the reference's known-answer test (reference: cuda_test.py:19-34) and a 1M-element scan.
 * Exit code 0 = pass.  With argument "link-only" it only checks that the symbols resolve (no GPU). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "grouped_cumprod_hip.h"

/* the four HIP runtime entry points used, declared by hand to stay plain C */
extern int hipMalloc(void** p, size_t n);
extern int hipFree(void* p);
extern int hipMemcpy(void* dst, const void* src, size_t n, int kind); /* 1 = H2D, 2 = D2H */
extern int hipDeviceSynchronize(void);

#define CHECK(x) do { int rc_ = (x); if (rc_) { printf("FAIL %s -> %d (%s)\n", #x, rc_, gcp_status_string(rc_)); return 1; } } while (0)

static int run_kat(void) {
  const float param[5] = {0.4f, 0.2f, 0.1f, 0.8f, 0.2f};
  const int32_t index[5] = {0, 0, 1, 1, 2}, index_len[3] = {2, 4, 5};
  const float want_cp[5] = {0.4f, 0.08f, 0.1f, 0.08f, 0.2f}, want_g[5] = {0.44f, 0.08f, 0.74f, 0.08f, 0.2f};
  float *d_p, *d_cp, *d_g;
  int32_t *d_i, *d_l;
  float cp[5], g[5];
  if (hipMalloc((void**)&d_p, 20) || hipMalloc((void**)&d_cp, 20) || hipMalloc((void**)&d_g, 20) ||
      hipMalloc((void**)&d_i, 20) || hipMalloc((void**)&d_l, 12)) return 1;
  hipMemcpy(d_p, param, 20, 1); hipMemcpy(d_i, index, 20, 1); hipMemcpy(d_l, index_len, 12, 1);
  CHECK(gcp_cumprod_forward(d_p, d_i, d_cp, 5, NULL, 0, NULL));
  CHECK(gcp_cumprod_backward(d_p, d_cp, d_p, d_i, d_g, d_l, 5, 3, NULL, 0, NULL));
  hipDeviceSynchronize();
  hipMemcpy(cp, d_cp, 20, 2); hipMemcpy(g, d_g, 20, 2);
  for (int i = 0; i < 5; ++i)
    if (fabsf(cp[i] - want_cp[i]) > 1e-6f || fabsf(g[i] - want_g[i]) > 1e-6f) { printf("KAT mismatch at %d: %g %g\n", i, cp[i], g[i]); return 1; }
  hipFree(d_p); hipFree(d_cp); hipFree(d_g); hipFree(d_i); hipFree(d_l);
  return 0;
}

static int run_big(void) {
  const int64_t n = 1 << 20;
  float* x = (float*)malloc(n * 4); float* y = (float*)malloc(n * 4);
  int32_t* k = (int32_t*)malloc(n * 4);
  for (int64_t i = 0; i < n; ++i) { x[i] = 1.0f; k[i] = (int32_t)(i / 37); }
  float *d_x, *d_y; int32_t* d_k; void* ws;
  const size_t wsb = gcp_workspace_bytes(n);
  if (hipMalloc((void**)&d_x, n * 4) || hipMalloc((void**)&d_y, n * 4) || hipMalloc((void**)&d_k, n * 4) || hipMalloc(&ws, wsb)) return 1;
  hipMemcpy(d_x, x, n * 4, 1); hipMemcpy(d_k, k, n * 4, 1);
  CHECK(gcp_workspace_init(ws, wsb, NULL));
  CHECK(gcp_cumsum_forward(d_x, d_k, d_y, n, ws, wsb, NULL));  /* caller-provided workspace */
  hipDeviceSynchronize();
  hipMemcpy(y, d_y, n * 4, 2);
  for (int64_t i = 0; i < n; ++i)
    if (y[i] != (float)(i % 37 + 1)) { printf("cumsum mismatch at %lld: %g\n", (long long)i, y[i]); return 1; }
  CHECK(gcp_cumsum_forward(d_x, d_k, d_y, n, ws, 16, NULL) == GCP_ERR_WORKSPACE ? 0 : 1);  /* too-small workspace is refused */
  hipFree(d_x); hipFree(d_y); hipFree(d_k); hipFree(ws); free(x); free(y); free(k);
  return 0;
}

int main(int argc, char** argv) {
  if (gcp_abi_version() != GCP_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
  if (argc > 1 && strcmp(argv[1], "link-only") == 0) { printf("link ok, tile = %d elements\n", gcp_tile_elems()); return 0; }
  if (run_kat()) return 1;
  if (run_big()) return 1;
  printf("abi_smoke ok\n");
  return 0;
}
