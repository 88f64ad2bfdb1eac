// gcp_project.hip — the caller's per-Gaussian camera projection, fused.
//
// reference: gs_model.py:277-425 (GS_model_with_param.forward up to the Function call): ~150 small PyTorch
// kernels, batched 3x3 matmuls and a CPU round trip for torch.linalg.eigh per step.  At N = 1e6 Gaussians that
// is 64 ms forward + 110 ms backward around a 1.8 ms rasterise-and-blend.  Here one thread owns one Gaussian:
// k_project_fwd writes, in the Gaussians' own order, everything the Function needs of it for one camera;
// k_project_bwd recomputes the chain in registers and turns (dL/dSigma'^-1, dL/dopacity, dL/dl_d) into the
// gradients of (mean, quaternion, log-scale, opacity logit, SH coefficients).  HBM bound: 152 B in, 77 B out per
// Gaussian forward; no LDS, no cross-lane traffic.  Depth order: the forward also emits a 31-bit sort key per
// Gaussian (bits of the positive depth; culled last) for gcp_sort_pairs_u32, and k_project_gather unpacks the
// kept records in that order.
//
// The arithmetic follows the reference's order of operations (matrix products accumulated left to right, k
// ascending, no FMA contraction: the library is built with -ffp-contract=off) so that the integer boxes that come
// out of float -> int32 truncation agree with the reference's.
#include "gcp_device.hpp"
#include "grouped_cumprod_hip.h"

namespace {

using gcp::i64;

constexpr int kThreads = 256;
constexpr float kShC0 = 0.28209479177387814f;
constexpr float kShC1 = 0.4886025119029199f;
constexpr float kShC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                            0.5462742152960396f};

struct Camera {
  float P[12];  // world -> camera [R|t], row major 3x4
  float K[9];   // intrinsics, row major 3x3
};

__device__ __forceinline__ Camera load_camera(const float* __restrict__ P, const float* __restrict__ K) {
  Camera c;
#pragma unroll
  for (int i = 0; i < 12; ++i) c.P[i] = P[i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.K[i] = K[i];
  return c;
}

// Everything between the parameters of one Gaussian and the arguments of the Function.
struct Projected {
  float t[3];        // mean in camera coordinates (gs_model.py:289-290)
  float px, py;      // pixel mean before truncation (:293-294)
  float qn[4], qlen; // unit quaternion (x, y, z, w) and the clamped norm (:297)
  float R[9];        // rotation (:299)
  float s[3];        // exp(log scale) (:302)
  float S[9];        // covariance, world (:307)
  float Sc[9];       // covariance, camera (:309)
  float J[6];        // 2x3 Jacobian (:311)
  float cov[4];      // pixel covariance before the clamp (:321)
  float a, b, c, d;  // pixel covariance after clamp + 1e-6 I
  float det;         // a d - b c + 1e-6 (uitility.py:447-451)
  float view[3], tlen;  // direction towards the camera (:337)
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__device__ __forceinline__ void project_one(const Camera& cam, const float* __restrict__ mean, const float* __restrict__ q,
                                            const float* __restrict__ log_scale, i64 i, Projected& o) {
  const float m0 = mean[3 * i], m1 = mean[3 * i + 1], m2 = mean[3 * i + 2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
    o.t[j] = ((m0 * cam.P[4 * j] + m1 * cam.P[4 * j + 1]) + m2 * cam.P[4 * j + 2]) + cam.P[4 * j + 3];
  float ph[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) ph[j] = (o.t[0] * cam.K[3 * j] + o.t[1] * cam.K[3 * j + 1]) + o.t[2] * cam.K[3 * j + 2];
  const float pz = fmaxf(ph[2], 1e-2f);
  o.px = ph[0] / pz;
  o.py = ph[1] / pz;

  const float qx = q[4 * i], qy = q[4 * i + 1], qz = q[4 * i + 2], qw = q[4 * i + 3];
  o.qlen = fmaxf(sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw), 1e-8f);
  const float x = qx / o.qlen, y = qy / o.qlen, z = qz / o.qlen, w = qw / o.qlen;
  o.qn[0] = x, o.qn[1] = y, o.qn[2] = z, o.qn[3] = w;
  float* R = o.R;
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - w * z), R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y), R[7] = 2 * (y * z + w * x), R[8] = 1 - 2 * (x * x + y * y);
#pragma unroll
  for (int k = 0; k < 3; ++k) o.s[k] = expf(log_scale[3 * i + k]);
  // R diag(s) diag(s)^T R^T, left to right
  float B[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) B[3 * r + k] = (R[3 * r + k] * o.s[k]) * o.s[k];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) o.S[3 * r + c] = (B[3 * r] * R[3 * c] + B[3 * r + 1] * R[3 * c + 1]) + B[3 * r + 2] * R[3 * c + 2];
  // W S W^T with W = P[:, :3]
  float WS[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      WS[3 * r + c] = (cam.P[4 * r] * o.S[c] + cam.P[4 * r + 1] * o.S[3 + c]) + cam.P[4 * r + 2] * o.S[6 + c];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      o.Sc[3 * r + c] = (WS[3 * r] * cam.P[4 * c] + WS[3 * r + 1] * cam.P[4 * c + 1]) + WS[3 * r + 2] * cam.P[4 * c + 2];
  // Jacobian of the pinhole projection (uitility.py:257-287)
  const float fx = cam.K[0], fy = cam.K[4];
  const float zc = fmaxf(o.t[2], 1e-2f);
  float* J = o.J;
  J[0] = fx / zc, J[1] = 0.f, J[2] = -fx * o.t[0] / (zc * zc);
  J[3] = 0.f, J[4] = fy / zc, J[5] = -fy * o.t[1] / (zc * zc);
  float JS[6];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) JS[3 * r + c] = (J[3 * r] * o.Sc[c] + J[3 * r + 1] * o.Sc[3 + c]) + J[3 * r + 2] * o.Sc[6 + c];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) o.cov[2 * r + c] = (JS[3 * r] * J[3 * c] + JS[3 * r + 1] * J[3 * c + 1]) + JS[3 * r + 2] * J[3 * c + 2];
  const float lim = 3.4028234663852886e+38f / 1000.f;
  o.a = clampf(o.cov[0], -lim, lim) + 1e-6f;
  o.b = clampf(o.cov[1], -lim, lim);
  o.c = clampf(o.cov[2], -lim, lim);
  o.d = clampf(o.cov[3], -lim, lim) + 1e-6f;
  o.det = (o.a * o.d - o.b * o.c) + 1e-6f;
  o.tlen = fmaxf(sqrtf((o.t[0] * o.t[0] + o.t[1] * o.t[1]) + o.t[2] * o.t[2]), 1e-8f);
#pragma unroll
  for (int k = 0; k < 3; ++k) o.view[k] = -o.t[k] / o.tlen;
}

// The parameter rows of a block's 256 consecutive Gaussians, copied into LDS with contiguous 16-byte loads: read
// straight from global memory they are 37 four-byte loads per thread, 12-108 bytes apart between neighbouring lanes.
// Layout: mean [256][3] | quaternion [256][4] | log scale [256][3] | SH [256][3 n_basis].
struct ParamTile {
  float *mean, *q, *ls, *sh;
};

__device__ __forceinline__ ParamTile param_tile(float* base, int n_basis) {
  ParamTile t;
  t.mean = base;
  t.q = t.mean + 3 * kThreads;
  t.ls = t.q + 4 * kThreads;
  t.sh = t.ls + 3 * kThreads;
  (void)n_basis;
  return t;
}

__device__ __forceinline__ void copy_rows(float* __restrict__ dst, const float* __restrict__ src, i64 first_word, int words) {
  // first_word is a multiple of 4 (256 rows per block), so the run starts 16-byte aligned
  const float4* s4 = reinterpret_cast<const float4*>(src + first_word);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int j = threadIdx.x; j < (words >> 2); j += kThreads) d4[j] = s4[j];
  for (int j = (words & ~3) + threadIdx.x; j < words; j += kThreads) dst[j] = src[first_word + j];
}

__device__ __forceinline__ void load_param_tile(const ParamTile& t, const float* mean, const float* q, const float* log_scale,
                                                const float* color, i64 base, int cnt, int n_basis) {
  copy_rows(t.mean, mean, 3 * base, 3 * cnt);
  copy_rows(t.q, q, 4 * base, 4 * cnt);
  copy_rows(t.ls, log_scale, 3 * base, 3 * cnt);
  copy_rows(t.sh, color, 3 * (i64)n_basis * base, 3 * n_basis * cnt);
}

// 3 sqrt(V^2 |lambda|) of the symmetric matrix read from the lower triangle (gs_model.py:327-332)
__device__ __forceinline__ void box_halfsize(float a, float b, float c, float& hx, float& hy) {
  const float m = 0.5f * (a + c), d = 0.5f * (a - c);
  const float r = sqrtf(d * d + b * b);
  const float lo = m - r, hi = m + r;
  float ex = a, ey = c;
  if (!(lo >= 0.f)) {
    const float ratio = r > 0.f ? d / r : 0.f;
    const float w_hi = 0.5f * (1.f + ratio), w_lo = 0.5f * (1.f - ratio);
    ex = w_lo * fabsf(lo) + w_hi * fabsf(hi);
    ey = w_hi * fabsf(lo) + w_lo * fabsf(hi);
  }
  hx = 3.f * sqrtf(fabsf(ex));
  hy = 3.f * sqrtf(fabsf(ey));
}

__device__ __forceinline__ int trunc_i32(float v) { return (int)v; }

// Per Gaussian, one 64-byte record (what the gather reads back in one piece), the sort key of its depth and the cull flag.
//   record words: 0-3 box x0 y0 x1 y1 | 4-5 pixel mean | 6-9 Sigma'^-1 | 10 opacity | 11-13 colour | 14-15 unused

__global__ __launch_bounds__(kThreads) void k_project_fwd(
    const float* __restrict__ mean, const float* __restrict__ q, const float* __restrict__ log_scale,
    const float* __restrict__ opacity, const float* __restrict__ color, const float* __restrict__ cam_P,
    const float* __restrict__ cam_K, i64 n, int sh_degree, int n_basis, int width, int height, float box_clamp,
    float4* __restrict__ record, int* __restrict__ sort_key, uint8_t* __restrict__ keep, int* __restrict__ row_of) {
  extern __shared__ float s_stage[];
  const ParamTile tile = param_tile(s_stage, n_basis);
  const Camera cam = load_camera(cam_P, cam_K);
  for (i64 base = (i64)blockIdx.x * kThreads; base < n; base += (i64)gridDim.x * kThreads) {
    const int cnt = (int)min((i64)kThreads, n - base);
    __syncthreads();  // the previous chunk's rows are no longer read
    load_param_tile(tile, mean, q, log_scale, color, base, cnt, n_basis);
    __syncthreads();
    const i64 i = base + threadIdx.x;
    if (i >= n) continue;
    Projected p;
    project_one(cam, tile.mean, tile.q, tile.ls, threadIdx.x, p);
    float hx, hy;
    box_halfsize(p.a, p.c, p.d, hx, hy);
    const float ilim = 2147483647.f / 1000.f;
    const int mx = trunc_i32(clampf(p.px, -ilim, ilim)), my = trunc_i32(clampf(p.py, -ilim, ilim));
    const int bw = trunc_i32(fminf(hx, box_clamp)), bh = trunc_i32(fminf(hy, box_clamp));
    const bool k = p.t[2] > 0.f && bw != 0 && mx - bw < width && mx + bw > 0 && my - bh < height && my + bh > 0;
    const int x0 = min(max(mx - bw, 0), width), y0 = min(max(my - bh, 0), height);
    const int x1 = min(max(mx + bw, 0), width), y1 = min(max(my + bh, 0), height);
    keep[i] = k ? 1 : 0;
    row_of[i] = -1;
    // kept depths are positive floats: their bit patterns sort like the values; culled Gaussians sort last
    sort_key[i] = k ? __float_as_int(p.t[2]) : 0x7fffffff;
    // real spherical harmonics, degree <= 2 (the build's eval_sh; the reference's sh_utility is absent)
    const float* sh = tile.sh + threadIdx.x * n_basis * 3;
    const float x = p.view[0], y = p.view[1], z = p.view[2];
    float l[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float v = kShC0 * sh[ch];
      if (sh_degree > 0) {
        v = ((v - kShC1 * y * sh[3 + ch]) + kShC1 * z * sh[6 + ch]) - kShC1 * x * sh[9 + ch];
        if (sh_degree > 1) {
          const float xx = x * x, yy = y * y, zz = z * z;
          v = ((((v + kShC2[0] * (x * y) * sh[12 + ch]) + kShC2[1] * (y * z) * sh[15 + ch]) +
                kShC2[2] * (2.f * zz - xx - yy) * sh[18 + ch]) + kShC2[3] * (x * z) * sh[21 + ch]) +
              kShC2[4] * (xx - yy) * sh[24 + ch];
        }
      }
      l[ch] = v;
    }
    const float alpha = 1.f / (1.f + expf(-opacity[i]));
    float4* rec = record + 4 * i;
    rec[0] = make_float4(__int_as_float(x0), __int_as_float(y0), __int_as_float(x1), __int_as_float(y1));
    rec[1] = make_float4(__int_as_float(mx), __int_as_float(my), p.d / p.det, -p.b / p.det);
    rec[2] = make_float4(-p.c / p.det, p.a / p.det, alpha, l[0]);
    rec[3] = make_float4(l[1], l[2], 0.f, 0.f);
  }
}

// Row r of the depth-ordered list is Gaussian perm[r]: unpack its record into the Function's argument arrays.
__global__ __launch_bounds__(kThreads) void k_project_gather(
    const float4* __restrict__ record, const int* __restrict__ perm, i64 m, int* __restrict__ start_xy,
    int* __restrict__ end_xy, int* __restrict__ mean_xy, i64* __restrict__ boxsize, float* __restrict__ vinv,
    float* __restrict__ alpha, float* __restrict__ l_d, i64* __restrict__ index, int* __restrict__ row_of,
    const unsigned char* __restrict__ keep) {
  for (i64 r = (i64)blockIdx.x * kThreads + threadIdx.x; r < m; r += (i64)gridDim.x * kThreads) {
    const int i = perm[r];
    const float4* rec = record + 4 * (i64)i;
    const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
    int x0 = __float_as_int(a.x), y0 = __float_as_int(a.y), x1 = __float_as_int(a.z), y1 = __float_as_int(a.w);
    // `keep` given (the list holds ALL Gaussians, no kept count was read back): a culled one stays in the list behind
    // the kept ones with an EMPTY box — binned into no tile, blended nowhere, zero gradients (its row_of stays -1)
    const bool culled = keep != nullptr && keep[i] == 0;
    if (culled) { x0 = 1; y0 = 1; x1 = 0; y1 = 0; }
    reinterpret_cast<int2*>(start_xy)[r] = make_int2(x0, y0);
    reinterpret_cast<int2*>(end_xy)[r] = make_int2(x1, y1);
    reinterpret_cast<int2*>(mean_xy)[r] = make_int2(__float_as_int(b.x), __float_as_int(b.y));
    boxsize[r] = (i64)(x1 - x0 + 1) * (i64)(y1 - y0 + 1);
    reinterpret_cast<float4*>(vinv)[r] = make_float4(b.z, b.w, c.x, c.y);
    alpha[r] = c.z;
    l_d[3 * r] = c.w, l_d[3 * r + 1] = d.x, l_d[3 * r + 2] = d.y;
    index[r] = i;
    if (!culled) row_of[i] = (int)r;
  }
}

// One thread per Gaussian, in the Gaussians' own order (coalesced parameter reads and gradient writes); the only
// scattered reads are the 8 upstream gradient words of its row `row_of[i]` in the depth-ordered list.  Culled
// Gaussians (row -1) get zeros: every gradient row is written, nothing needs clearing first.
__global__ __launch_bounds__(kThreads) void k_project_bwd(
    const float* __restrict__ mean, const float* __restrict__ q, const float* __restrict__ log_scale,
    const float* __restrict__ opacity, const float* __restrict__ color, const float* __restrict__ cam_P,
    const float* __restrict__ cam_K, i64 n, int sh_degree, int n_basis, const int* __restrict__ row_of,
    const float* __restrict__ g_vinv, const float* __restrict__ g_alpha, const float* __restrict__ g_ld,
    float* __restrict__ grad_mean, float* __restrict__ grad_q, float* __restrict__ grad_log_scale,
    float* __restrict__ grad_opacity, float* __restrict__ grad_color) {
  // parameter rows come in and gradient rows go out through LDS as contiguous runs: straight from / to registers they are
  // 37 + 38 four-byte accesses per thread, 12-108 bytes apart (gradient rows direct: 365 us per 10^6 Gaussians; staged: 140)
  extern __shared__ float s_stage[];
  float* s_mean = s_stage;
  float* s_q = s_mean + 3 * kThreads;
  float* s_ls = s_q + 4 * kThreads;
  float* s_sh = s_ls + 3 * kThreads;
  const int sh_words = 3 * n_basis;
  float* lm = s_mean + 3 * threadIdx.x;
  float* lq = s_q + 4 * threadIdx.x;
  float* lls = s_ls + 3 * threadIdx.x;
  float* gsh = s_sh + sh_words * threadIdx.x;
  const Camera cam = load_camera(cam_P, cam_K);
  const ParamTile tile = param_tile(s_stage, n_basis);  // same layout as the gradient rows: a thread's parameter row is
                                                         // replaced, in place and by that thread alone, with its gradient row
  for (i64 base = (i64)blockIdx.x * kThreads; base < n; base += (i64)gridDim.x * kThreads) {
    load_param_tile(tile, mean, q, log_scale, color, base, (int)min((i64)kThreads, n - base), n_basis);
    __syncthreads();
    const i64 i = base + threadIdx.x;
    const i64 r = i < n ? row_of[i] : -1;
    if (r < 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) lm[k] = 0.f, lls[k] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) lq[k] = 0.f;
      for (int k = 0; k < sh_words; ++k) gsh[k] = 0.f;
      if (i < n) grad_opacity[i] = 0.f;
    } else {
    Projected p;
    project_one(cam, tile.mean, tile.q, tile.ls, threadIdx.x, p);  // reads this thread's row before anything overwrites it

    // opacity = sigmoid(o)
    const float al = 1.f / (1.f + expf(-opacity[i]));
    grad_opacity[i] = g_alpha[r] * al * (1.f - al);

    // colour: l_d[ch] = sum_k B_k(view) sh[k][ch]
    const float x = p.view[0], y = p.view[1], z = p.view[2];
    float Bk[9] = {kShC0, -kShC1 * y, kShC1 * z, -kShC1 * x, kShC2[0] * x * y, kShC2[1] * y * z,
                   kShC2[2] * (2.f * z * z - x * x - y * y), kShC2[3] * x * z, kShC2[4] * (x * x - y * y)};
    const int nb = (sh_degree + 1) * (sh_degree + 1);
    float gv[3] = {0.f, 0.f, 0.f};  // dL/dview
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float g = g_ld[3 * r + ch];
      // this channel's coefficients first: the gradient row goes into the very words they are read from
      float c[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) c[k] = k < nb ? gsh[3 * k + ch] : 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (k < n_basis) gsh[3 * k + ch] = k < nb ? g * Bk[k] : 0.f;
      for (int k = 9; k < n_basis; ++k) gsh[3 * k + ch] = 0.f;
      if (sh_degree > 0) {
        gv[0] += g * (-kShC1 * c[3]);
        gv[1] += g * (-kShC1 * c[1]);
        gv[2] += g * (kShC1 * c[2]);
        if (sh_degree > 1) {
          gv[0] += g * (kShC2[0] * y * c[4] - 2.f * kShC2[2] * x * c[6] + kShC2[3] * z * c[7] + 2.f * kShC2[4] * x * c[8]);
          gv[1] += g * (kShC2[0] * x * c[4] + kShC2[1] * z * c[5] - 2.f * kShC2[2] * y * c[6] - 2.f * kShC2[4] * y * c[8]);
          gv[2] += g * (kShC2[1] * y * c[5] + 4.f * kShC2[2] * z * c[6] + kShC2[3] * x * c[7]);
        }
      }
    }
    float gt[3];  // dL/dt (camera-space mean)
    {
      const float dot = gv[0] * x + gv[1] * y + gv[2] * z;
      const float vv[3] = {x, y, z};
#pragma unroll
      for (int k = 0; k < 3; ++k) gt[k] = -(gv[k] - vv[k] * dot) / p.tlen;
    }

    // Sigma'^-1 = adj(A) / det  ->  dL/dA
    const float g00 = g_vinv[4 * r], g01 = g_vinv[4 * r + 1], g10 = g_vinv[4 * r + 2], g11 = g_vinv[4 * r + 3];
    const float sdot = ((g00 * p.d - g01 * p.b) - g10 * p.c) + g11 * p.a;
    const float gdet = -sdot / (p.det * p.det);
    const float lim = 3.4028234663852886e+38f / 1000.f;
    float D[4] = {g11 / p.det + gdet * p.d, -g01 / p.det - gdet * p.c, -g10 / p.det - gdet * p.b, g00 / p.det + gdet * p.a};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (!(fabsf(p.cov[k]) <= lim)) D[k] = 0.f;  // clamped (or NaN): no gradient

    // A = J Sc J^T:  dL/dSc = J^T D J,  dL/dJ = D J Sc^T + D^T J Sc
    const float* J = p.J;
    float gSc[9], gJ[6];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        gSc[3 * a + b] = (J[a] * D[0] + J[3 + a] * D[2]) * J[b] + (J[a] * D[1] + J[3 + a] * D[3]) * J[3 + b];
    {
      float JS[6], JSt[6];  // J Sc and J Sc^T
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          JS[3 * r2 + c] = J[3 * r2] * p.Sc[c] + J[3 * r2 + 1] * p.Sc[3 + c] + J[3 * r2 + 2] * p.Sc[6 + c];
          JSt[3 * r2 + c] = J[3 * r2] * p.Sc[3 * c] + J[3 * r2 + 1] * p.Sc[3 * c + 1] + J[3 * r2 + 2] * p.Sc[3 * c + 2];
        }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        gJ[c] = (D[0] * JSt[c] + D[1] * JSt[3 + c]) + (D[0] * JS[c] + D[2] * JS[3 + c]);
        gJ[3 + c] = (D[2] * JSt[c] + D[3] * JSt[3 + c]) + (D[1] * JS[c] + D[3] * JS[3 + c]);
      }
    }
    // J(t): only entries (0,0), (0,2), (1,1), (1,2) depend on t
    {
      const float fx = cam.K[0], fy = cam.K[4];
      const float zc = fmaxf(p.t[2], 1e-2f), iz2 = 1.f / (zc * zc), iz3 = iz2 / zc;
      gt[0] += gJ[2] * (-fx * iz2);
      gt[1] += gJ[5] * (-fy * iz2);
      if (p.t[2] > 1e-2f)
        gt[2] += gJ[0] * (-fx * iz2) + gJ[2] * (2.f * fx * p.t[0] * iz3) + gJ[4] * (-fy * iz2) + gJ[5] * (2.f * fy * p.t[1] * iz3);
    }
    // t = W m + t0  ->  dL/dm = W^T dL/dt
#pragma unroll
    for (int k = 0; k < 3; ++k) lm[k] = (cam.P[k] * gt[0] + cam.P[4 + k] * gt[1]) + cam.P[8 + k] * gt[2];

    // Sc = W S W^T  ->  E = dL/dS = W^T gSc W
    float E[9];
    {
      float T[9];  // W^T gSc
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) T[3 * a + b] = cam.P[a] * gSc[b] + cam.P[4 + a] * gSc[3 + b] + cam.P[8 + a] * gSc[6 + b];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) E[3 * a + b] = T[3 * a] * cam.P[b] + T[3 * a + 1] * cam.P[4 + b] + T[3 * a + 2] * cam.P[8 + b];
    }
    // S = R diag(s^2) R^T:  dL/dR = (E + E^T) R diag(s^2),  dL/d(log s_k) = 2 s_k^2 (R^T E R)_kk
    const float* R = p.R;
    float gR[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float er = (E[3 * a] + E[a]) * R[k] + (E[3 * a + 1] + E[3 + a]) * R[3 + k] + (E[3 * a + 2] + E[6 + a]) * R[6 + k];
        gR[3 * a + k] = er * (p.s[k] * p.s[k]);
      }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc += R[3 * a + k] * E[3 * a + b] * R[3 * b + k];
      lls[k] = 2.f * (p.s[k] * p.s[k]) * acc;
    }
    // R(qn), qn = q / |q|
    {
      const float qx = p.qn[0], qy = p.qn[1], qz = p.qn[2], qw = p.qn[3];
      float g[4];
      g[0] = 2.f * (qy * gR[1] + qz * gR[2] + qy * gR[3] - 2.f * qx * gR[4] - qw * gR[5] + qz * gR[6] + qw * gR[7] - 2.f * qx * gR[8]);
      g[1] = 2.f * (-2.f * qy * gR[0] + qx * gR[1] + qw * gR[2] + qx * gR[3] + qz * gR[5] - qw * gR[6] + qz * gR[7] - 2.f * qy * gR[8]);
      g[2] = 2.f * (-2.f * qz * gR[0] - qw * gR[1] + qx * gR[2] + qw * gR[3] - 2.f * qz * gR[4] + qy * gR[5] + qx * gR[6] + qy * gR[7]);
      g[3] = 2.f * (-qz * gR[1] + qy * gR[2] + qz * gR[3] - qx * gR[5] - qy * gR[6] + qx * gR[7]);
      const float dot = g[0] * qx + g[1] * qy + g[2] * qz + g[3] * qw;
      const bool clamped = !(p.qlen > 1e-8f);
#pragma unroll
      for (int k = 0; k < 4; ++k) lq[k] = clamped ? g[k] / p.qlen : (g[k] - p.qn[k] * dot) / p.qlen;
    }
    }  // kept Gaussian
    __syncthreads();
    const int cnt = (int)min((i64)kThreads, n - base);
    for (int j = threadIdx.x; j < 3 * cnt; j += kThreads) grad_mean[3 * base + j] = s_mean[j], grad_log_scale[3 * base + j] = s_ls[j];
    for (int j = threadIdx.x; j < 4 * cnt; j += kThreads) grad_q[4 * base + j] = s_q[j];
    float* out_sh = grad_color + base * sh_words;
    for (int j = threadIdx.x; j < sh_words * cnt; j += kThreads) out_sh[j] = s_sh[j];
    __syncthreads();
  }
}

int grid_for(i64 n) { return (int)((n + kThreads - 1) / kThreads < 65536 ? (n + kThreads - 1) / kThreads : 65536); }

}  // namespace

extern "C" {

int gcp_project_forward(const float* mean, const float* quat_xyzw, const float* log_scale, const float* opacity_logit,
                        const float* sh_coeff, const float* cam_P, const float* cam_K, int64_t n_gauss, int32_t sh_degree,
                        int32_t n_basis, int32_t width, int32_t height, float box_clamp, float* record, int32_t* sort_key,
                        uint8_t* keep, int32_t* row_of, void* stream) {
  if (n_gauss < 0 || n_gauss > 0x7fffffff || sh_degree < 0 || sh_degree > 2 || n_basis < (sh_degree + 1) * (sh_degree + 1) ||
      width < 0 || height < 0)
    return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss == 0) return GCP_OK;
  if (!mean || !quat_xyzw || !log_scale || !opacity_logit || !sh_coeff || !cam_P || !cam_K || !record || !sort_key || !keep ||
      !row_of || ((uintptr_t)record & 15))
    return GCP_ERR_INVALID_ARGUMENT;
  const size_t lds_fwd = (size_t)kThreads * (10 + 3 * (size_t)n_basis) * sizeof(float);
  if (lds_fwd > 64 * 1024) return GCP_ERR_INVALID_ARGUMENT;  // n_basis <= 18
  hipLaunchKernelGGL(k_project_fwd, dim3(grid_for(n_gauss)), dim3(kThreads), lds_fwd, (hipStream_t)stream, mean, quat_xyzw, log_scale,
                     opacity_logit, sh_coeff, cam_P, cam_K, (i64)n_gauss, (int)sh_degree, (int)n_basis, (int)width, (int)height,
                     box_clamp, (float4*)record, sort_key, keep, row_of);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_project_gather(const float* record, const int32_t* perm, int64_t n_kept, int32_t* start_xy, int32_t* end_xy,
                       int32_t* mean_xy, int64_t* boxsize, float* vinv, float* alpha, float* l_d, int64_t* index,
                       int32_t* row_of, const uint8_t* keep, void* stream) {
  if (n_kept < 0) return GCP_ERR_INVALID_ARGUMENT;
  if (n_kept == 0) return GCP_OK;
  if (!record || !perm || !start_xy || !end_xy || !mean_xy || !boxsize || !vinv || !alpha || !l_d || !index || !row_of ||
      (((uintptr_t)record | (uintptr_t)vinv) & 15) || (((uintptr_t)start_xy | (uintptr_t)end_xy | (uintptr_t)mean_xy) & 7))
    return GCP_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(k_project_gather, dim3(grid_for(n_kept)), dim3(kThreads), 0, (hipStream_t)stream, (const float4*)record, perm,
                     (i64)n_kept, start_xy, end_xy, mean_xy, (i64*)boxsize, vinv, alpha, l_d, (i64*)index, row_of,
                     (const unsigned char*)keep);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

int gcp_project_backward(const float* mean, const float* quat_xyzw, const float* log_scale, const float* opacity_logit,
                         const float* sh_coeff, const float* cam_P, const float* cam_K, int64_t n_gauss, int32_t sh_degree,
                         int32_t n_basis, const int32_t* row_of, const float* grad_vinv, const float* grad_alpha,
                         const float* grad_l_d, float* grad_mean, float* grad_quat, float* grad_log_scale,
                         float* grad_opacity_logit, float* grad_sh_coeff, void* stream) {
  if (n_gauss < 0 || sh_degree < 0 || sh_degree > 2 || n_basis < (sh_degree + 1) * (sh_degree + 1)) return GCP_ERR_INVALID_ARGUMENT;
  if (n_gauss == 0) return GCP_OK;
  if (!mean || !quat_xyzw || !log_scale || !opacity_logit || !sh_coeff || !cam_P || !cam_K || !row_of || !grad_mean ||
      !grad_quat || !grad_log_scale || !grad_opacity_logit || !grad_sh_coeff)
    return GCP_ERR_INVALID_ARGUMENT;  // the three upstream arrays may be NULL when no Gaussian was kept
  const size_t lds = (size_t)kThreads * (10 + 3 * (size_t)n_basis) * sizeof(float);
  if (lds > 64 * 1024) return GCP_ERR_INVALID_ARGUMENT;  // n_basis <= 18
  hipLaunchKernelGGL(k_project_bwd, dim3(grid_for(n_gauss)), dim3(kThreads), lds, (hipStream_t)stream, mean, quat_xyzw, log_scale,
                     opacity_logit, sh_coeff, cam_P, cam_K, (i64)n_gauss, (int)sh_degree, (int)n_basis, row_of, grad_vinv, grad_alpha,
                     grad_l_d, grad_mean, grad_quat, grad_log_scale, grad_opacity_logit, grad_sh_coeff);
  GCP_HIP(hipGetLastError());
  return GCP_OK;
}

}  // extern "C"
