// Standalone A/B of the trunk convolution kernels (no torch, no Python): conv:: (round 2) against conv2:: (persistent),
// numerics compared, every launch timed through its dispatch packet.   hipcc --offload-arch=gfx950 -O3 ... lab.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "sgmcmc_hip.h"
#include "sgmcmc_hip_alternatives.h"   // (the lab compares the default kernels with the measured alternatives)

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

__device__ long long g_trace[1024 * 16];
#ifndef NO_STAMPS
#define CONV2_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_trace[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#endif
// sustained rate of v_mfma_f32_16x16x4_f32 from registers: N MFMAs per wave on 4 independent accumulators
template <int N>
__global__ __launch_bounds__(256, 2) void mfma_rate_kernel(float* out, float a0, float b0) {
  typedef __attribute__((ext_vector_type(4))) float v4;
  v4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = v4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0;
#pragma unroll 8
  for (int i = 0; i < N / 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  v4 s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s[0] == 1234.5f) out[threadIdx.x] = s[1];
}
__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0.f; }

#include "conv_hip.inc"
#include "conv2_hip.inc"

static float* dalloc(size_t n) { float* p; CK(hipMalloc(&p, n * sizeof(float))); return p; }
static void fill(float* d, size_t n, unsigned seed, float scale, bool relu = false) {
  std::vector<float> h(n);
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
  for (size_t i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    float u = ((s >> 33) & 0xFFFFFF) / 16777216.0f, v;
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v = ((s >> 33) & 0xFFFFFF) / 16777216.0f;
    float z = sqrtf(-2.f * logf(u + 1e-7f)) * cosf(6.2831853f * v) * scale;
    h[i] = relu ? fmaxf(z, 0.f) : z;
  }
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}
static double maxdiff(const float* a, const float* b, size_t n, double* scale) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double d = 0, s = 0;
  for (size_t i = 0; i < n; ++i) { d = std::max(d, (double)fabsf(ha[i] - hb[i])); s = std::max(s, (double)fabsf(ha[i])); }
  *scale = s;
  return d;
}
static double maxdiff_d(const double* a, const double* b, size_t n, double* scale) {
  std::vector<double> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost));
  double d = 0, s = 0;
  for (size_t i = 0; i < n; ++i) { d = std::max(d, fabs(ha[i] - hb[i])); s = std::max(s, fabs(ha[i])); }
  *scale = s;
  return d;
}

template <typename F>
static void chain(const char* name, F fn, int n = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) fn();
  CK(hipDeviceSynchronize());
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < n; ++i) fn();
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  printf("  chain of %d x %-30s %7.2f us per launch (stream order, back to back)\n", n, name, best * 1e3 / n);
  fflush(stdout);
}

template <typename F>
static void timeit(const char* name, F fn, int iters = 40) {
  for (int i = 0; i < 5; ++i) fn();
  CK(hipDeviceSynchronize());
  std::vector<float> ms(iters);
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < iters; ++i) {
    sgmcmc_timing::e0 = ev[2 * i]; sgmcmc_timing::e1 = ev[2 * i + 1];
    fn();
  }
  CK(hipDeviceSynchronize());
  double sum = 0; float mn = 1e9;
  for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1])); sum += ms[i]; mn = std::min(mn, ms[i]); }
  for (auto& e : ev) CK(hipEventDestroy(e));
  printf("  %-34s avg %7.2f us  min %7.2f us\n", name, 1e3 * sum / iters, 1e3 * mn);
  fflush(stdout);
}

static void dump_stamps(const char* what, int b0, int b1) {
#ifdef NO_STAMPS
  return;
#endif
  std::vector<long long> t(1024 * 16);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), t.size() * 8));
  long long t0 = t[b0 * 16], t4 = 0;
  const int n = b1 - b0;
  for (int b = b0; b < b1; ++b) { t0 = std::min(t0, t[b * 16]); t4 = std::max(t4, t[b * 16 + 4]); }
  double m[16] = {0};
  for (int b = b0; b < b1; ++b) for (int k = 0; k < 16; ++k) m[k] += (double)(t[b * 16 + k] - t0) / n;
  printf("  %s stamps (10 ns ticks, mean over workgroups %d..%d): start %.0f | item0: band in LDS %.0f, MFMAs issued %.0f, item done %.0f | "
         "item1: begin %.0f, band in LDS %.0f, MFMAs issued %.0f, results ready %.0f, stored %.0f, item done %.0f | end %.0f, last end %lld\n",
         what, b0, b1, m[0], m[1], m[2], m[3], m[8], m[9], m[10], m[11], m[12], m[13], m[4], t4 - t0);
}

template <int C, int HW>
static void run_shape(int n_img, int n_wg) {
  constexpr int R2 = 4;
  using G2 = conv2::Cfg<C, HW, R2>;
  using W2 = conv2::WCfg<C, HW, R2>;
  const size_t act = (size_t)n_img * C * HW * HW, wn = (size_t)C * C * 9;
  printf("== C=%d HW=%d n=%d  (v2: %d workgroups, LDS fwd %zu B, bwd %zu B)\n", C, HW, n_img, n_wg, G2::LDS_BYTES, conv2::bwd_lds<C, HW, R2>());
  float *x = dalloc(act), *dy = dalloc(act), *out = dalloc(act), *w = dalloc(wn);
  float *y1 = dalloc(act), *y2 = dalloc(act), *dx1 = dalloc(act), *dx2 = dalloc(act), *dw1 = dalloc(wn), *dw2 = dalloc(wn);
  float *ffwd = dalloc(wn), *fdg = dalloc(wn), *mean = dalloc(C), *invstd = dalloc(C);
  fill(x, act, 1 + C, 1.f); fill(dy, act, 2 + C, 1.f); fill(out, act, 3 + C, 1.f, true); fill(w, wn, 4 + C, sqrtf(2.f / (9 * C)));
  fill(mean, C, 5, 0.1f); fill(invstd, C, 6, 0.1f);
  const int sl1 = n_img * (HW / 8), sl2 = n_img * (HW / R2);
  double *st1, *st2, *pa1, *pa2;
  CK(hipMalloc(&st1, (size_t)C * sl1 * 16)); CK(hipMalloc(&st2, (size_t)C * sl2 * 16));
  CK(hipMalloc(&pa1, (size_t)C * sl1 * 16)); CK(hipMalloc(&pa2, (size_t)C * sl2 * 16));
  const size_t scr1 = (size_t)((n_img * (HW / 8) + 1) / 2) * wn;
  float* part1 = dalloc(scr1);
  float* part2 = dalloc((size_t)(n_wg / 2) * W2::SLAB * 1 + wn);   // (ct * P + p) slabs, P = n_wrw / CT
  hipStream_t s = nullptr;

  // ---- weight fragments
  conv2::FragJobs J{};
  J.n = 1; J.job[0] = {w, ffwd, fdg, C, G2::KS}; J.first_block[0] = 0; J.first_block[1] = (int)((wn + 255) / 256);
  auto frag = [&] { SGMCMC_LAUNCH(conv2::weight_frag_kernel, dim3(J.first_block[1]), dim3(256), 0, s, J); };
  frag();
  CK(hipDeviceSynchronize());

  // ---- forward (+ stats)
  auto f1 = [&] { int e = conv::launch_conv3x3<C, HW, 8>(x, w, y1, n_img, false, st1, s); if (e) { printf("v1 fwd err %d\n", e); exit(1); } };
  auto k2 = conv2::fwd_kernel<C, HW, R2, true>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2::LDS_BYTES));
  auto f2 = [&] { SGMCMC_LAUNCH(k2, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); };
  f1(); f2(); CK(hipDeviceSynchronize());
  double sc, d = maxdiff(y1, y2, act, &sc);
  printf("  fwd   max|y1-y2| = %.3g (scale %.3g)\n", d, sc);
  {  // stats: compare per-channel totals of sums
    std::vector<double> h1((size_t)C * sl1 * 2), h2((size_t)C * sl2 * 2);
    CK(hipMemcpy(h1.data(), st1, h1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), st2, h2.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int c = 0; c < C; ++c) {
      double a = 0, b = 0;
      for (int i = 0; i < sl1; ++i) a += h1[((size_t)c * sl1 + i) * 2];
      for (int i = 0; i < sl2; ++i) b += h2[((size_t)c * sl2 + i) * 2];
      worst = std::max(worst, fabs(a - b));
    }
    printf("  stats max |sum1 - sum2| per channel = %.3g\n", worst);
  }
  timeit("v1 fwd+stats", f1);
  timeit("v2 fwd+stats", f2);
  chain("v1 fwd+stats", f1);
  chain("v2 fwd+stats", f2);
  {  // where a workgroup's time goes: wall-clock stamps (100 MHz) of every workgroup of the last launch
    dump_stamps("v2 fwd", 0, n_wg);
  }
  timeit("v2 weight_frag", frag);
  {
    auto k_nomfma = conv2::fwd_kernel<C, HW, R2, true, 1>;
    auto k_noload = conv2::fwd_kernel<C, HW, R2, true, 2>;
    auto k_nostore = conv2::fwd_kernel<C, HW, R2, true, 4>;
    auto k_mfmaonly = conv2::fwd_kernel<C, HW, R2, false, 6>;
    auto k_memonly = conv2::fwd_kernel<C, HW, R2, false, 1>;
    for (const void* k : {(const void*)k_nomfma, (const void*)k_noload, (const void*)k_nostore, (const void*)k_mfmaonly, (const void*)k_memonly})
      CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G2::LDS_BYTES));
    timeit("v2 fwd, no MFMA", [&] { SGMCMC_LAUNCH(k_nomfma, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); });
    timeit("v2 fwd, no input loads", [&] { SGMCMC_LAUNCH(k_noload, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); });
    timeit("v2 fwd, no output stores", [&] { SGMCMC_LAUNCH(k_nostore, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); });
    timeit("v2 fwd, MFMA only (no stats)", [&] { SGMCMC_LAUNCH(k_mfmaonly, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); });
    timeit("v2 fwd, memory only (no stats)", [&] { SGMCMC_LAUNCH(k_memonly, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, x, ffwd, y2, st2, n_img); });
    timeit("empty kernel, same grid + LDS", [&] { SGMCMC_LAUNCH(empty_kernel, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, y2); });
    timeit("empty kernel, 1 workgroup", [&] { SGMCMC_LAUNCH(empty_kernel, dim3(1), dim3(256), 0, s, y2); });
    timeit("hipMemcpyDtoD of the activations", [&] { SGMCMC_LAUNCH(empty_kernel, dim3(1), dim3(64), 0, s, y2); }, 5);
  }

  // ---- backward with the SUMS epilogue
  conv::BwdEpilogue E1{}, E2{};
  E1.s_y = x; E1.s_out = out; E1.s_mean = mean; E1.s_invstd = invstd; E1.s_partial = pa1;
  E2 = E1; E2.s_partial = pa2; E2.n_slices = sl2;
  int slabs = 0;
  auto b1 = [&] { int e = conv::launch_bwd<C, HW, 8>(x, w, dy, dx1, dw1, part1, n_img, &slabs, s, E1); if (e) { printf("v1 bwd err %d\n", e); exit(1); } };
  auto r1 = [&] { SGMCMC_LAUNCH(conv::wrw_reduce_kernel, dim3(conv::reduce_blocks(slabs, (int)wn)), dim3(256), 0, s, part1, slabs, (int)wn, dw1, 9); };
  auto kb = conv2::bwd_kernel<C, HW, R2, false, true>;
  const size_t lds2 = conv2::bwd_lds<C, HW, R2>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  const int n_wrw = n_wg / 2, P = n_wrw / G2::CT;
  auto b2 = [&] { SGMCMC_LAUNCH(kb, dim3(n_wg), dim3(256), lds2, s, x, fdg, dy, dx2, part2, n_wrw, n_img, E2); };
  auto r2 = [&] {
    for (int ct = 0; ct < G2::CT; ++ct)
      SGMCMC_LAUNCH(conv::wrw_reduce_kernel, dim3(conv::reduce_blocks(P, W2::SLAB)), dim3(256), 0, s,
                    part2 + (size_t)ct * P * W2::SLAB, P, W2::SLAB, dw2 + (size_t)ct * W2::SLAB, 9);
  };
  b1(); r1(); b2(); r2(); CK(hipDeviceSynchronize());
  d = maxdiff(dx1, dx2, act, &sc); printf("  bwd   max|dx1-dx2| = %.3g (scale %.3g)\n", d, sc);
  d = maxdiff(dw1, dw2, wn, &sc); printf("  bwd   max|dw1-dw2| = %.3g (scale %.3g)\n", d, sc);
  {
    std::vector<double> h1((size_t)C * sl1 * 2), h2((size_t)C * sl2 * 2);
    CK(hipMemcpy(h1.data(), pa1, h1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h2.data(), pa2, h2.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int c = 0; c < C; ++c) for (int q = 0; q < 2; ++q) {
      double a = 0, b = 0;
      for (int i = 0; i < sl1; ++i) a += h1[((size_t)c * sl1 + i) * 2 + q];
      for (int i = 0; i < sl2; ++i) b += h2[((size_t)c * sl2 + i) * 2 + q];
      worst = std::max(worst, fabs(a - b)); scale = std::max(scale, fabs(a));
    }
    printf("  sums  max |v1 - v2| per channel = %.3g (scale %.3g)\n", worst, scale);
  }
  chain("v1 bwd+sums", b1);
  chain("v2 bwd+sums", b2);
  {
    // the same launches on ROTATING operand sets (R sets x 4 tensors: far more than the 8 x 4 MB of L2), so that every
    // launch finds its operands where a step's kernel finds them -- in the Infinity Cache / HBM, not in its XCD's L2
    const int R = (int)std::max<size_t>(3, (size_t)(160u << 20) / (4 * act * sizeof(float)));
    std::vector<float*> xs(R), dys(R), outs(R), ys(R);
    for (int r = 0; r < R; ++r) {
      xs[r] = dalloc(act); dys[r] = dalloc(act); outs[r] = dalloc(act); ys[r] = dalloc(act);
      CK(hipMemcpy(xs[r], x, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(dys[r], dy, act * 4, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(outs[r], out, act * 4, hipMemcpyDeviceToDevice));
    }
    int k = 0;
    printf("  -- rotating over %d operand sets (%.0f MB): L2-cold launches\n", R, R * 4.0 * act * 4 / 1e6);
    chain("v1 fwd+stats, cold", [&] { k = (k + 1) % R; conv::launch_conv3x3<C, HW, 8>(xs[k], w, ys[k], n_img, false, st1, s); });
    chain("v2 fwd+stats, cold", [&] { k = (k + 1) % R; SGMCMC_LAUNCH(k2, dim3(n_wg), dim3(256), G2::LDS_BYTES, s, xs[k], ffwd, ys[k], st2, n_img); });
    chain("v1 bwd+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = E1; E.s_y = xs[k]; E.s_out = outs[k];
                                      conv::launch_bwd<C, HW, 8>(xs[k], w, dys[k], ys[k], dw1, part1, n_img, &slabs, s, E); });
    chain("v2 bwd+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = E2; E.s_y = xs[k]; E.s_out = outs[k];
                                      SGMCMC_LAUNCH(kb, dim3(n_wg), dim3(256), lds2, s, xs[k], fdg, dys[k], ys[k], part2, n_wrw, n_img, E); });
    conv::BwdEpilogue Ea1 = E1, Ea2 = E2;
    auto kba = conv2::bwd_kernel<C, HW, R2, true, true>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kba), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    chain("v1 bwd+add+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = Ea1; E.s_y = xs[k]; E.s_out = outs[k]; E.e_dout = dys[(k + 1) % R]; E.e_out = outs[(k + 1) % R];
                                          conv::launch_bwd<C, HW, 8>(xs[k], w, dys[k], ys[k], dw1, part1, n_img, &slabs, s, E); });
    chain("v2 bwd+add+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = Ea2; E.s_y = xs[k]; E.s_out = outs[k]; E.e_dout = dys[(k + 1) % R]; E.e_out = outs[(k + 1) % R];
                                          SGMCMC_LAUNCH(kba, dim3(n_wg), dim3(256), lds2, s, xs[k], fdg, dys[k], ys[k], part2, n_wrw, n_img, E); });
    for (int r = 0; r < R; ++r) { CK(hipFree(xs[r])); CK(hipFree(dys[r])); CK(hipFree(outs[r])); CK(hipFree(ys[r])); }
  }
  timeit("v1 bwd+sums", b1);
  timeit("v2 bwd+sums (256 wrw + 256 dgrad)", b2);
  // the halves alone
  auto kd = conv2::bwd_kernel<C, HW, R2, false, true>;
  auto d2 = [&] { SGMCMC_LAUNCH(kd, dim3(n_wg), dim3(256), lds2, s, x, fdg, dy, dx2, part2, 0, n_img, E2); };
  auto w2 = [&] { SGMCMC_LAUNCH(kd, dim3(n_wg / 2), dim3(256), lds2, s, x, fdg, dy, dx2, part2, n_wg / 2, n_img, E2); };
  auto d2h = [&] { SGMCMC_LAUNCH(kd, dim3(n_wg / 2), dim3(256), lds2, s, x, fdg, dy, dx2, part2, 0, n_img, E2); };
  timeit("v2 dgrad+sums alone (512 wg)", d2);
  timeit("v2 dgrad+sums alone (256 wg)", d2h);
  dump_stamps("v2 dgrad+sums 256wg", 0, n_wg / 2);
  b2(); CK(hipDeviceSynchronize());
  dump_stamps("v2 merged bwd, dgrad half", n_wg / 2, n_wg);
  timeit("v2 wrw alone (256 wg)", w2);
  auto d1 = [&] { conv::launch_conv3x3<C, HW, 8>(dy, w, dx1, n_img, true, nullptr, s); };
  timeit("v1 dgrad alone", d1);
  auto w1 = [&] { conv::launch_wrw<C, HW, 8>(x, dy, dw1, part1, n_img, s); };
  timeit("v1 wrw alone (first launch)", w1);
  CK(hipDeviceSynchronize());
  for (float* p : {x, dy, out, w, y1, y2, dx1, dx2, dw1, dw2, ffwd, fdg, mean, invstd, part1, part2}) CK(hipFree(p));
  CK(hipFree(st1)); CK(hipFree(st2)); CK(hipFree(pa1)); CK(hipFree(pa2));
}

int main(int argc, char** argv) {
  const int n_img = argc > 1 ? atoi(argv[1]) : 128;
  const int n_wg = argc > 2 ? atoi(argv[2]) : 512;
  {
    float* o = dalloc(1024);
    hipStream_t s = nullptr;
    printf("== MFMA rate (v_mfma_f32_16x16x4_f32 from registers, 512 workgroups x 4 waves): 2048 flop per wave-MFMA\n");
    timeit("144 MFMAs per wave (= one trunk conv)", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<144>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    timeit("288 MFMAs per wave (= one conv bwd)", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<288>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    timeit("1152 MFMAs per wave", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<1152>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    timeit("9216 MFMAs per wave", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<9216>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    timeit("empty kernel, 512 workgroups", [&] { SGMCMC_LAUNCH(empty_kernel, dim3(512), dim3(256), 0, s, o); });
    chain("empty", [&] { SGMCMC_LAUNCH(empty_kernel, dim3(512), dim3(256), 0, s, o); });
    chain("144 MFMAs", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<144>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    chain("288 MFMAs", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<288>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    chain("1152 MFMAs", [&] { SGMCMC_LAUNCH(mfma_rate_kernel<1152>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f); });
    // back-to-back chain of 100 empty kernels between two events: the per-launch cost when nothing is timed per kernel
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(512), dim3(256), 0, s, o);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  chain of 200 empty kernels (stream order): %.2f us per kernel\n", ms * 1e3 / 200);
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(mfma_rate_kernel<144>, dim3(512), dim3(256), 0, s, o, 1.f, 2.f);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  chain of 200 x 144-MFMA kernels (stream order): %.2f us per kernel\n", ms * 1e3 / 200);
    }
    CK(hipFree(o));
  }
  run_shape<16, 32>(n_img, n_wg);
  run_shape<32, 16>(n_img, n_wg);
  run_shape<64, 8>(n_img, n_wg);
  return 0;
}
