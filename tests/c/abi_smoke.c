/* abi_smoke.c — a plain C program on the C ABI (no Python, no torch, no C++): what a host written in
 * another language would do through its FFI.  Builds with gcc against include/grouped_cumprod_hip.h and
 * libgrouped_cumprod_hip.so; device memory comes from the HIP runtime's C API.
 * Runs the reference's known-answer test (reference: cuda_test.py:19-34), a 1M-element scan, the three-stage general route
 * of _create_alpha_brend (gs_model.py:544-566) on the reference's worked example for it (uitility.py:383-393), the rect cut
 * step by step, and the default route of _create_alpha_brend (one-call cut, binning, walk, kept count) on two boxes.
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

/* _create_alpha_brend through the C ABI alone: gcp_sort_rects -> gcp_cumprod_forward_indexed -> gcp_compact_finish.
 * The reference's worked example of a grouped cumprod returned in ORIGINAL order (uitility.py:383-393):
 * A = [1..7], groups (1,1),(1,2),(1,1),(1,2),(1,3),(1,1),(1,3) -> inclusive [1,2,3,8,5,18,35];
 * _create_alpha_brend then divides by self -> exclusive [1,1,1,2,1,3,5], nothing dropped. */
static int run_alpha_brend(void) {
  const int64_t n = 7;
  const int32_t rects[14] = {1, 1, 2, 1, 1, 1, 2, 1, 3, 1, 1, 1, 3, 1};  /* (x, y) with y = 1: key = 10000 + x */
  const float a[7] = {1, 2, 3, 4, 5, 6, 7};
  const float want_incl[7] = {1, 2, 3, 8, 5, 18, 35}, want_excl[7] = {1, 1, 1, 2, 1, 3, 5};
  int32_t *d_r, *d_idx, *d_cnt; uint32_t* d_key; float *d_a, *d_incl, *d_val; uint8_t* d_keep; void *ws_sort, *ws_scan, *ws_cmp;
  const size_t b_sort = gcp_sort_workspace_bytes(n), b_scan = gcp_workspace_bytes(n), b_cmp = gcp_compact_workspace_bytes(n);
  if (hipMalloc((void**)&d_r, 56) || hipMalloc((void**)&d_idx, 28) || hipMalloc((void**)&d_key, 28) || hipMalloc((void**)&d_a, 28) ||
      hipMalloc((void**)&d_incl, 28) || hipMalloc((void**)&d_val, 28) || hipMalloc((void**)&d_keep, 8) || hipMalloc((void**)&d_cnt, 4) ||
      hipMalloc(&ws_sort, b_sort) || hipMalloc(&ws_scan, b_scan) || hipMalloc(&ws_cmp, b_cmp)) return 1;
  hipMemcpy(d_r, rects, 56, 1); hipMemcpy(d_a, a, 28, 1);
  CHECK(gcp_sort_rects(d_r, n, 14 /* bits of 10003 */, 0, d_key, d_idx, ws_sort, b_sort, NULL));
  CHECK(gcp_workspace_init(ws_scan, b_scan, NULL));
  CHECK(gcp_cumprod_forward_indexed(d_a, (const int32_t*)d_key, d_idx, d_incl, n, ws_scan, b_scan, NULL));
  CHECK(gcp_compact_finish(d_incl, d_a, 0, n, 0, d_val, d_keep, d_cnt, NULL, ws_cmp, b_cmp, NULL));
  hipDeviceSynchronize();
  float incl[7], val[7]; uint8_t keep[7]; int32_t cnt = -1, idx[7]; uint32_t key[7];
  hipMemcpy(incl, d_incl, 28, 2); hipMemcpy(val, d_val, 28, 2); hipMemcpy(keep, d_keep, 7, 2); hipMemcpy(&cnt, d_cnt, 4, 2);
  hipMemcpy(idx, d_idx, 28, 2); hipMemcpy(key, d_key, 28, 2);
  const int32_t want_idx[7] = {0, 2, 5, 1, 3, 4, 6};  /* stable: input order inside a key */
  if (cnt != 7) { printf("alpha_brend: kept %d of 7\n", cnt); return 1; }
  for (int i = 0; i < 7; ++i)
    if (incl[i] != want_incl[i] || val[i] != want_excl[i] || keep[i] != 1 || idx[i] != want_idx[i] || key[i] != 10000u + (uint32_t)rects[2 * want_idx[i]]) {
      printf("alpha_brend mismatch at %d: incl %g excl %g keep %d idx %d key %u\n", i, incl[i], val[i], keep[i], idx[i], key[i]);
      return 1;
    }
  hipFree(d_r); hipFree(d_idx); hipFree(d_key); hipFree(d_a); hipFree(d_incl); hipFree(d_val); hipFree(d_keep); hipFree(d_cnt);
  hipFree(ws_sort); hipFree(ws_scan); hipFree(ws_cmp);
  return 0;
}

/* The rect list cut back into rectangles through the C ABI (gcp_rects_rows -> gcp_rows_rectangles -> gcp_rectangle_boxes)
 * on the same seven pairs: rows [(1,1),(2,1)], [(1,1),(2,1),(3,1)], [(1,1)], [(3,1)] — four one-row rectangles. */
static int run_rect_cut(void) {
  const int64_t n = 7;
  const int32_t rects[14] = {1, 1, 2, 1, 1, 1, 2, 1, 3, 1, 1, 1, 3, 1};
  const int64_t cap = gcp_rects_rows_capacity(n) + n;  /* seven pairs in four rows: more rows than a list of boxes is allowed */
  int32_t *d_r, *d_rs, *d_rxy, *d_info, *d_rr, *d_info2, *d_s, *d_e, *d_off; void *ws1, *ws2;
  const size_t b1 = gcp_rects_rows_workspace_bytes(n);
  if (hipMalloc((void**)&d_r, 56) || hipMalloc((void**)&d_rs, cap * 4) || hipMalloc((void**)&d_rxy, cap * 8) || hipMalloc((void**)&d_info, 20) ||
      hipMalloc(&ws1, b1)) return 1;
  hipMemcpy(d_r, rects, 56, 1);
  CHECK(gcp_rects_rows(d_r, n, cap, d_rs, d_rxy, d_info, ws1, b1, NULL));
  hipDeviceSynchronize();
  int32_t info[5]; hipMemcpy(info, d_info, 20, 2);
  if (info[0] != 4 || info[1] != 3 || info[2] != 1 || info[3] != 1 || info[4] != 0) { printf("rect cut: rows info %d %d %d %d %d\n", info[0], info[1], info[2], info[3], info[4]); return 1; }
  const int64_t n_rows = info[0];
  const size_t b2 = gcp_rows_rectangles_workspace_bytes(n_rows);
  if (hipMalloc((void**)&d_rr, (n_rows + 1) * 4) || hipMalloc((void**)&d_info2, 8) || hipMalloc(&ws2, b2)) return 1;
  CHECK(gcp_rows_rectangles(d_rs, d_rxy, n_rows, d_rr, d_info2, ws2, b2, NULL));
  hipDeviceSynchronize();
  int32_t n_rects = -1; hipMemcpy(&n_rects, d_info2, 4, 2);
  if (n_rects != 4) { printf("rect cut: %d rectangles\n", n_rects); return 1; }
  if (hipMalloc((void**)&d_s, n_rects * 8) || hipMalloc((void**)&d_e, n_rects * 8) || hipMalloc((void**)&d_off, (n_rects + 1) * 4)) return 1;
  CHECK(gcp_rectangle_boxes(d_rr, d_rs, d_rxy, n_rects, n, d_s, d_e, d_off, NULL));
  hipDeviceSynchronize();
  int32_t st[8], en[8], off[5];
  hipMemcpy(st, d_s, 32, 2); hipMemcpy(en, d_e, 32, 2); hipMemcpy(off, d_off, 20, 2);
  const int32_t want_s[8] = {1, 1, 1, 1, 1, 1, 3, 1}, want_e[8] = {2, 1, 3, 1, 1, 1, 3, 1}, want_off[5] = {0, 2, 5, 6, 7};
  if (memcmp(st, want_s, 32) || memcmp(en, want_e, 32) || memcmp(off, want_off, 20)) { printf("rect cut: boxes differ\n"); return 1; }
  hipFree(d_r); hipFree(d_rs); hipFree(d_rxy); hipFree(d_info); hipFree(d_rr); hipFree(d_info2); hipFree(d_s); hipFree(d_e); hipFree(d_off);
  hipFree(ws1); hipFree(ws2);
  return 0;
}

/* The DEFAULT route of _create_alpha_brend (gs_model.py:544-566) through the C ABI alone, two device->host reads:
 * gcp_rects_cut (rows, rectangles, boxes, tiles per box; read info8) -> gcp_bin_tiles_fill -> gcp_pairs_finish_boxes (the walk
 * writes final values + mask) -> gcp_compact_kept_count (read the kept count) -> gcp_compact_kept_write.
 * Two boxes in depth order, A = [0,9] x [0,7] with factor 0.5 (0 at pixel (6,4): opaque) and B = [5,14] x [3,10] with factor
 * 0.25: A's pairs see transmittance 1, B's 0.5 where A lies in front and 1 elsewhere; the opaque pair and B's pair behind it
 * have an inclusive product of exactly 0 and are dropped (gs_model.py:560). */
static int run_default_route(void) {
  enum { NA = 80, NB = 80, N = 160, CAP = 8 };
  int32_t rects[2 * N]; float a[N], want[N]; uint8_t want_keep[N];
  int64_t p = 0;
  for (int y = 0; y <= 7; ++y) for (int x = 0; x <= 9; ++x, ++p) {
    rects[2 * p] = x; rects[2 * p + 1] = y;
    const int opaque = (x == 6 && y == 4);
    a[p] = opaque ? 0.0f : 0.5f; want[p] = 1.0f; want_keep[p] = !opaque;
  }
  for (int y = 3; y <= 10; ++y) for (int x = 5; x <= 14; ++x, ++p) {
    rects[2 * p] = x; rects[2 * p + 1] = y;
    const int behind_a = (x <= 9 && y <= 7);
    a[p] = 0.25f; want[p] = behind_a ? 0.5f : 1.0f; want_keep[p] = !(x == 6 && y == 4);
  }
  int32_t *d_r, *d_s, *d_e, *d_boff, *d_toff, *d_info, *d_tstart, *d_tlist, *d_drop, *d_cnt; float *d_a, *d_fin, *d_val; uint8_t* d_keep;
  void *ws_cut, *ws_bin, *ws_cmp;
  const size_t b_cut = gcp_rects_cut_workspace_bytes(N, 0, 0, 512, CAP), b_cmp = gcp_compact_kept_workspace_bytes(N);
  if (hipMalloc((void**)&d_r, sizeof rects) || hipMalloc((void**)&d_a, sizeof a) || hipMalloc((void**)&d_s, CAP * 8) || hipMalloc((void**)&d_e, CAP * 8) ||
      hipMalloc((void**)&d_boff, (CAP + 1) * 4) || hipMalloc((void**)&d_toff, (CAP + 1) * 4) || hipMalloc((void**)&d_info, 32) ||
      hipMalloc(&ws_cut, b_cut) || hipMalloc(&ws_cmp, b_cmp)) return 1;
  hipMemcpy(d_r, rects, sizeof rects, 1); hipMemcpy(d_a, a, sizeof a, 1);
  CHECK(gcp_rects_cut(d_r, 0, N, 0, 0, 512, CAP, d_s, d_e, d_boff, d_toff, d_info, ws_cut, b_cut, NULL));
  hipDeviceSynchronize();
  int32_t info[8]; hipMemcpy(info, d_info, 32, 2);  /* read 1: {rows, max x, max y, min, flags, rectangles, K, 0} */
  const int32_t want_info[8] = {16, 14, 10, 0, 0, 2, 2, 0};
  if (memcmp(info, want_info, 32)) { printf("default route: info %d %d %d %d %d %d %d\n", info[0], info[1], info[2], info[3], info[4], info[5], info[6]); return 1; }
  const int32_t w = info[1], h = info[2], n_rects = info[5], K = info[6];
  int32_t tx = 0, ty = 0;
  CHECK(gcp_tile_grid(w, h, &tx, &ty));
  const size_t b_bin = gcp_bin_workspace_bytes(n_rects, K);
  if (hipMalloc((void**)&d_tstart, (size_t)(tx * ty + 1) * 4) || hipMalloc((void**)&d_tlist, (size_t)K * 4) || hipMalloc(&ws_bin, b_bin) ||
      hipMalloc((void**)&d_fin, N * 4) || hipMalloc((void**)&d_val, N * 4) || hipMalloc((void**)&d_keep, N) || hipMalloc((void**)&d_drop, 4) ||
      hipMalloc((void**)&d_cnt, 4)) return 1;
  CHECK(gcp_bin_tiles_fill(d_s, d_e, n_rects, w, h, d_toff, K, d_tstart, d_tlist, ws_bin, b_bin, NULL));
  CHECK(gcp_pairs_finish_boxes(d_s, d_e, n_rects, w, h, d_tstart, d_tlist, d_boff, d_a, d_fin, d_keep, N, 0, d_drop, 0, NULL));
  CHECK(gcp_compact_kept_count(d_keep, d_drop, N, 0, N, d_cnt, ws_cmp, b_cmp, NULL));
  hipDeviceSynchronize();
  int32_t kept = -1; hipMemcpy(&kept, d_cnt, 4, 2);  /* read 2: sizes the result, as the reference's output[mask] does */
  if (kept != N - 2) { printf("default route: kept %d of %d\n", kept, N); return 1; }
  CHECK(gcp_compact_kept_write(d_fin, d_keep, 0, N, d_val, ws_cmp, b_cmp, NULL));
  hipDeviceSynchronize();
  float val[N]; uint8_t keep[N];
  hipMemcpy(val, d_val, (size_t)kept * 4, 2); hipMemcpy(keep, d_keep, N, 2);
  int k = 0;
  for (int i = 0; i < N; ++i) {
    if (keep[i] != want_keep[i]) { printf("default route: mask differs at %d\n", i); return 1; }
    if (keep[i] && val[k++] != want[i]) { printf("default route: value %d is %g, expected %g\n", i, val[k - 1], want[i]); return 1; }
  }
  hipFree(d_r); hipFree(d_a); hipFree(d_s); hipFree(d_e); hipFree(d_boff); hipFree(d_toff); hipFree(d_info); hipFree(d_tstart); hipFree(d_tlist);
  hipFree(d_fin); hipFree(d_val); hipFree(d_keep); hipFree(d_drop); hipFree(d_cnt); hipFree(ws_cut); hipFree(ws_bin); hipFree(ws_cmp);
  return 0;
}

/* The per-pixel carry of a chunk (gs_model.py:582-586, :724-730) through the C ABI: the two boxes of run_default_route with
 * values 0.5 (A) and 0.25 (B): 80 + 80 - 25 distinct pixels in (x, y) order, the minimum 0.25 wherever B lies; with values =
 * NULL the index of every pixel's first pair. */
static int run_pixels_min(void) {
  enum { N = 160, W = 14, H = 10, CELLS = (W + 1) * (H + 1) };
  int32_t rects[2 * N]; float a[N]; int first[W + 1][H + 1]; float mn[W + 1][H + 1];
  for (int x = 0; x <= W; ++x) for (int y = 0; y <= H; ++y) { first[x][y] = -1; mn[x][y] = 2.0f; }
  int64_t p = 0;
  for (int y = 0; y <= 7; ++y) for (int x = 0; x <= 9; ++x, ++p) { rects[2 * p] = x; rects[2 * p + 1] = y; a[p] = 0.5f; }
  for (int y = 3; y <= 10; ++y) for (int x = 5; x <= 14; ++x, ++p) { rects[2 * p] = x; rects[2 * p + 1] = y; a[p] = 0.25f; }
  for (int i = 0; i < N; ++i) {
    const int x = rects[2 * i], y = rects[2 * i + 1];
    if (first[x][y] < 0) first[x][y] = i;
    if (a[i] < mn[x][y]) mn[x][y] = a[i];
  }
  int32_t *d_r, *d_xy, *d_info; float *d_a, *d_val; void* ws;
  const size_t b = gcp_pixels_min_workspace_bytes(W, H);
  if (!b || hipMalloc((void**)&d_r, sizeof rects) || hipMalloc((void**)&d_a, sizeof a) || hipMalloc((void**)&d_xy, CELLS * 8) ||
      hipMalloc((void**)&d_val, CELLS * 4) || hipMalloc((void**)&d_info, 16) || hipMalloc(&ws, b)) return 1;
  hipMemcpy(d_r, rects, sizeof rects, 1); hipMemcpy(d_a, a, sizeof a, 1);
  for (int index = 0; index < 2; ++index) {
    CHECK(gcp_pixels_min(d_r, 0, index ? NULL : d_a, N, W, H, d_xy, d_val, CELLS, d_info, ws, b, NULL));
    hipDeviceSynchronize();
    int32_t info[4]; hipMemcpy(info, d_info, 16, 2);
    if (info[0] != 135 || info[1] != 0) { printf("pixels min: info %d %d\n", info[0], info[1]); return 1; }
    int32_t xy[2 * CELLS]; float val[CELLS];
    hipMemcpy(xy, d_xy, (size_t)info[0] * 8, 2); hipMemcpy(val, d_val, (size_t)info[0] * 4, 2);
    int k = 0;
    for (int x = 0; x <= W; ++x) for (int y = 0; y <= H; ++y) {
      if (first[x][y] < 0) continue;
      const float want = index ? (float)first[x][y] : mn[x][y];
      if (xy[2 * k] != x || xy[2 * k + 1] != y || val[k] != want) { printf("pixels min (%d): row %d is (%d, %d) %g, expected (%d, %d) %g\n", index, k, xy[2 * k], xy[2 * k + 1], val[k], x, y, want); return 1; }
      ++k;
    }
    if (k != info[0]) { printf("pixels min: %d rows checked, %d returned\n", k, info[0]); return 1; }
  }
  /* a coordinate outside the image the caller names: flagged, the pair skipped */
  CHECK(gcp_pixels_min(d_r, 0, d_a, N, W - 1, H, d_xy, d_val, CELLS, d_info, ws, b, NULL));
  hipDeviceSynchronize();
  int32_t info[4]; hipMemcpy(info, d_info, 16, 2);
  if (info[1] != 1) { printf("pixels min: a coordinate outside the image was not flagged\n"); return 1; }
  int32_t ext[3]; int32_t* d_ext;
  if (hipMalloc((void**)&d_ext, 12)) return 1;
  CHECK(gcp_pixels_range(d_r, 0, N, d_ext, NULL));
  hipDeviceSynchronize();
  hipMemcpy(ext, d_ext, 12, 2);
  if (ext[0] != W || ext[1] != H || ext[2] != 0) { printf("pixels range: %d %d %d\n", ext[0], ext[1], ext[2]); return 1; }
  hipFree(d_r); hipFree(d_a); hipFree(d_xy); hipFree(d_val); hipFree(d_info); hipFree(ws); hipFree(d_ext);
  return 0;
}

int main(int argc, char** argv) {
  if (gcp_abi_version() != GCP_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
  if (argc > 1 && strcmp(argv[1], "link-only") == 0) { printf("link ok, tile = %d elements\n", gcp_tile_elems()); return 0; }
  if (run_kat()) return 1;
  if (run_big()) return 1;
  if (run_alpha_brend()) return 1;
  if (run_rect_cut()) return 1;
  if (run_default_route()) return 1;
  if (run_pixels_min()) return 1;
  printf("abi_smoke ok\n");
  return 0;
}
