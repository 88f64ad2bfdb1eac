// What does a global_load_dword / global_store_dword cost per wave instruction when only some lanes are active?
// (not part of the product: a measurement for DESIGN.md §3.4's reading of the tile-list walk)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/vmem_issue.hip -o /tmp/vmem_issue && /tmp/vmem_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// every wave does `iters` dependent-free loads (+ stores) at 64 B x 4 rows pattern; `mask` selects the active lanes
template <bool STORE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, float* __restrict__ y, int iters, unsigned long long mask,
                                         int row_stride /*floats*/, size_t span) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool on = (mask >> lane) & 1ull;
  // lane -> (row = lane / 16, col = lane % 16): four row pieces of 16 floats, rows `row_stride` floats apart
  size_t off = (wave * 4099u * 64u) % span + (size_t)(lane >> 4) * row_stride + (lane & 15);
  float acc = 0.0f;
  for (int i = 0; i < iters; i += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (off + (size_t)(i + u) * 977u * 64u) % span;
      v[u] = on ? x[o] : 0.0f;
      if (STORE && on) y[o] = (float)(i + u);  // independent of the loads: nothing waits inside the loop
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u];
  }
  if (acc == 12345.678f) y[0] = acc;
}

int main() {
  const size_t span = (size_t)160 << 20;  // floats: 640 MB
  float *x, *y;
  hipMalloc(&x, (span + 4096) * 4);
  hipMalloc(&y, (span + 4096) * 4);
  hipMemset(x, 0, (span + 4096) * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 8160, iters = 448;  // ~ the walk's 14.6e6 wave instructions = 8160 x 4 waves x 448
  struct Case { const char* name; unsigned long long mask; int stride; };
  unsigned long long m24 = 0;
  for (int r = 0; r < 3; ++r) m24 |= 0xffull << (16 * r);  // 3 rows x 8 columns = 24 lanes
  unsigned long long m13x4 = 0;
  for (int r = 0; r < 4; ++r) m13x4 |= 0x1fffull << (16 * r);  // 4 rows x 13 columns
  const Case cases[] = {{"64 lanes, rows of 16 contiguous (stride 16)", ~0ull, 16},
                        {"64 lanes, rows 13 apart (a box row)", ~0ull, 13},
                        {"52 lanes = 4 rows x 13, rows 13 apart", m13x4, 13},
                        {"24 lanes = 3 rows x 8, rows 13 apart", m24, 13},
                        {"4 lanes (one per row)", 0x0001000100010001ull, 13},
                        {"1 lane", 1ull, 13}};
  for (int st = 0; st < 2; ++st)
    for (const Case& c : cases) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        if (st) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, x, y, iters, c.mask, c.stride, span);
        else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, x, y, iters, c.mask, c.stride, span);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
      }
      const double instr = (double)blocks * 4 * iters * (st ? 2 : 1);
      printf("%-6s %-48s %8.3f ms  %6.2f ns per wave instruction per CU-slot (x256 CUs: %5.1f cycles @2.4GHz)\n", st ? "ld+st" : "load", c.name, best,
             best * 1e6 / instr, best * 1e-3 / instr * 256 * 2.4e9);
    }
  return 0;
}
