// Standalone timing / phase stamps of the convolutional classifier's 50 -> 50 @ 14x14 convolution (conv50_hip.inc)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "sgmcmc_hip.h"
namespace sgmcmc_timing { static hipEvent_t e0 = nullptr, e1 = nullptr; }
#define SGMCMC_LAUNCH(kernel, grid, block, lds, stream, ...)                                           \
  do {                                                                                                 \
    if (sgmcmc_timing::e0) {                                                                           \
      hipEvent_t a_ = sgmcmc_timing::e0, b_ = sgmcmc_timing::e1;                                       \
      sgmcmc_timing::e0 = sgmcmc_timing::e1 = nullptr;                                                 \
      hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, a_, b_, 0, __VA_ARGS__);                 \
    } else {                                                                                           \
      hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                               \
    }                                                                                                  \
  } while (0)
#define SGMCMC_FRESH_ERROR_STATE() (void)hipGetLastError()
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__device__ long long g_trace[2048 * 8];
#ifndef NO_STAMPS
#define CONV50_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#endif
#include "conv_hip.inc"
#include "conv50_hip.inc"

static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); return p; }
static void fill(float* d, size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float u = ((s >> 33) & 0xFFFFFF) / 16777216.0f;
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float v = ((s >> 33) & 0xFFFFFF) / 16777216.0f;
    h[i] = sqrtf(-2.f * logf(u + 1e-7f)) * cosf(6.2831853f * v) * scale;
  }
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}
template <typename F>
static void timeit(const char* name, F fn, int iters = 40) {
  for (int i = 0; i < 5; ++i) fn();
  CK(hipDeviceSynchronize());
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < iters; ++i) { sgmcmc_timing::e0 = ev[2 * i]; sgmcmc_timing::e1 = ev[2 * i + 1]; fn(); }
  CK(hipDeviceSynchronize());
  double sum = 0; float mn = 1e9, ms;
  for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); sum += ms; mn = std::min(mn, ms); }
  for (auto& e : ev) CK(hipEventDestroy(e));
  printf("  %-28s avg %7.2f us  min %7.2f us\n", name, 1e3 * sum / iters, 1e3 * mn);
}
static void stamps(const char* what, int b0, int b1, int nk) {
#ifdef NO_STAMPS
  return;
#endif
  std::vector<long long> t(2048 * 8);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8));
  long long t0 = t[b0 * 8];
  for (int b = b0; b < b1; ++b) t0 = std::min(t0, t[b * 8]);
  printf("  %s stamps (10 ns ticks, mean over workgroups %d..%d):", what, b0, b1);
  for (int k = 0; k < nk; ++k) { double m = 0, mx = 0; for (int b = b0; b < b1; ++b) { m += (double)(t[b * 8 + k] - t0) / (b1 - b0); mx = std::max(mx, (double)(t[b * 8 + k] - t0)); } printf("  [%d] %.0f (max %.0f)", k, m, mx); }
  printf("\n");
}
// double-precision host reference of a few outputs
static void check(const char* what, const std::vector<float>& x, const std::vector<float>& w, const float* ydev, int n_img, bool transpose) {
  const int C = 50, HW = 14;
  std::vector<float> y((size_t)n_img * C * HW * HW);
  CK(hipMemcpy(y.data(), ydev, y.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, scale = 0;
  for (int s = 0; s < 4000; ++s) {
    const int img = (s * 7919) % n_img, co = (s * 31) % C, py = (s * 17) % HW, px = (s * 5) % HW;
    double a = 0;
    for (int ci = 0; ci < C; ++ci) for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
      const int yy = py + r - 1, xx = px + q - 1;
      if (yy < 0 || yy >= HW || xx < 0 || xx >= HW) continue;
      const double wv = transpose ? w[((size_t)ci * C + co) * 9 + (8 - (r * 3 + q))] : w[((size_t)co * C + ci) * 9 + r * 3 + q];
      a += wv * x[((size_t)img * C + ci) * HW * HW + yy * HW + xx];
    }
    worst = std::max(worst, fabs(a - y[((size_t)img * C + co) * HW * HW + py * HW + px])); scale = std::max(scale, fabs(a));
  }
  printf("  %s max |err| vs float64 = %.3g (scale %.3g)\n", what, worst, scale);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 128;
  const size_t act = (size_t)n * 50 * 196, wn = 50 * 50 * 9;
  float *x = dalloc(act), *dy = dalloc(act), *w = dalloc(wn), *y = dalloc(act), *dx = dalloc(act), *dw = dalloc(wn);
  fill(x, act, 1, 1.f); fill(dy, act, 2, 1.f); fill(w, wn, 3, 0.05f);
  std::vector<float> hx(act), hdy(act), hw(wn);
  CK(hipMemcpy(hx.data(), x, act * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hdy.data(), dy, act * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hw.data(), w, wn * 4, hipMemcpyDeviceToHost));
  float* scratch = dalloc((size_t)sgmcmc_conv50_scratch_floats(n));
  int slabs = 0;
  float* wT = dalloc(wn);
  auto fwd = [&] { int e = sgmcmc_conv50_fwd(x, w, y, wT, n, nullptr); if (e) { printf("fwd err %d\n", e); exit(1); } };
  auto dg = [&] { int e = sgmcmc_conv50(dy, w, dx, n, 1, nullptr); if (e) { printf("dgrad err %d\n", e); exit(1); } };
  auto wrw = [&] { int e = sgmcmc_conv50_bwd(x, w, dy, nullptr, nullptr, scratch, n, &slabs, nullptr); if (e) { printf("wrw err %d\n", e); exit(1); } };
  auto bwd = [&] { int e = sgmcmc_conv50_bwd_t(x, wT, dy, dx, nullptr, scratch, n, &slabs, nullptr); if (e) { printf("bwd err %d\n", e); exit(1); } };
  fwd(); CK(hipDeviceSynchronize()); check("fwd  ", hx, hw, y, n, false); stamps("fwd", 0, 2 * n, 4);
  dg(); CK(hipDeviceSynchronize()); check("dgrad", hdy, hw, dx, n, true); stamps("dgrad", 0, 2 * n, 4);
  {  // weight gradient: reduce and check a few entries; the data gradient of the merged launch (prepared wT)
    bwd(); CK(hipDeviceSynchronize());
    check("dgrad (merged, wT)", hdy, hw, dx, n, true);
    SGMCMC_LAUNCH(conv::wrw_reduce_kernel, dim3(conv::reduce_blocks(slabs, (int)wn)), dim3(256), 0, nullptr, scratch, slabs, (int)wn, dw, 9);
    std::vector<float> hdw(wn); CK(hipMemcpy(hdw.data(), dw, wn * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int s = 0; s < 300; ++s) {
      const int co = (s * 7) % 50, ci = (s * 13) % 50, rs = s % 9, r = rs / 3, q = rs % 3;
      double a = 0;
      for (int img = 0; img < n; ++img) for (int py = 0; py < 14; ++py) for (int px = 0; px < 14; ++px) {
        const int yy = py + r - 1, xx = px + q - 1;
        if (yy < 0 || yy >= 14 || xx < 0 || xx >= 14) continue;
        a += (double)hdy[((size_t)img * 50 + co) * 196 + py * 14 + px] * hx[((size_t)img * 50 + ci) * 196 + yy * 14 + xx];
      }
      worst = std::max(worst, fabs(a - hdw[((size_t)co * 50 + ci) * 9 + rs])); scale = std::max(scale, fabs(a));
    }
    printf("  wrw   max |err| vs float64 = %.3g (scale %.3g)\n", worst, scale);
  }
  timeit("conv50 fwd", fwd); timeit("conv50 dgrad", dg); timeit("conv50 wrw only", wrw); timeit("conv50 bwd", bwd);
  return 0;
}
