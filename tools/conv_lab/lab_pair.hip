// Lab (round 6): a BatchNorm backward and the PREVIOUS convolution's weight gradient in one launch -- memory-bound blocks beside
// matrix-pipe-bound workgroups -- against today's merged data + weight gradient launch followed by the BatchNorm launch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
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
#ifndef SGMCMC_WT_STORES
#define SGMCMC_WT_STORES 15
#endif
using sgmcmc_f32x4 = __attribute__((ext_vector_type(4))) float;
template <int FAMILY>
__device__ __forceinline__ void sgmcmc_store4(float* __restrict__ base, float* __restrict__ p, float a, float b, float c, float d) {
  if constexpr ((SGMCMC_WT_STORES & FAMILY) != 0) {
    const uint64_t off = (uint64_t)(reinterpret_cast<char*>(p) - reinterpret_cast<char*>(base));
    if (off < 0xfffffff0ull) {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(sgmcmc_f32x4{a, b, c, d}, r, (int)(uint32_t)off, 0, 17);
      return;
    }
  }
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

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
static std::vector<float> host(const float* d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }
static size_t bits_differ(const float* a, const float* b, size_t n) {
  auto ha = host(a, n), hb = host(b, n);
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) k += memcmp(&ha[i], &hb[i], 4) != 0;
  return k;
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
  printf("  chain of %d x %-34s %7.2f us per launch\n", n, name, best * 1e3 / n);
  fflush(stdout);
}
template <typename F>
static void timeit(const char* name, F fn, int iters = 40) {
  for (int i = 0; i < 5; ++i) fn();
  CK(hipDeviceSynchronize());
  std::vector<float> ms(iters);
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < iters; ++i) { sgmcmc_timing::e0 = ev[2 * i]; sgmcmc_timing::e1 = ev[2 * i + 1]; fn(); }
  CK(hipDeviceSynchronize());
  double sum = 0; float mn = 1e9;
  for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1])); sum += ms[i]; mn = std::min(mn, ms[i]); }
  for (auto& e : ev) CK(hipEventDestroy(e));
  printf("  %-40s avg %7.2f us  min %7.2f us\n", name, 1e3 * sum / iters, 1e3 * mn);
  fflush(stdout);
}


#include "conv_hip.inc"
#include "bn_hip.inc"

// ---- the experiment: a BatchNorm backward (memory bound, matrix pipe idle) and a convolution's weight gradient (matrix
// pipe bound) in ONE launch -- weight-gradient workgroups first, BatchNorm blocks behind them
template <int C, int HW>
__global__ __launch_bounds__(256) void bn_wrw_kernel(int n_wrw, const float* __restrict__ wx, const float* __restrict__ wdy,
                                                     float* __restrict__ part, int n_items,
                                                     const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const double* __restrict__ partial,
                                                     int n_partials, bn::Geo g, float* __restrict__ dx, float* __restrict__ dgb) {
  const int b = (int)blockIdx.x;
  if (b < n_wrw) conv::conv3x3_wrw_body<C, HW, 8>(conv::xcd_remap<C>(b, n_wrw), wx, wdy, part, n_items);
  else bn::bwd_dx_body<false, false, false>(b - n_wrw, dy, y, x, gamma, mean, invstd, partial, n_partials, g, dx, nullptr, dgb, bn::ResSums{});
}

template <int C, int HW>
static void run_shape(int n_img) {
  const size_t act = (size_t)n_img * C * HW * HW, wn = (size_t)C * C * 9;
  using W = conv::WrwCfg<C, HW, 8>;
  printf("== C=%d HW=%d n=%d\n", C, HW, n_img);
  float *x = dalloc(act), *dy = dalloc(act), *out = dalloc(act), *w = dalloc(wn), *edo = dalloc(act);
  float *dx1 = dalloc(act), *dw1 = dalloc(wn), *mean = dalloc(C), *invstd = dalloc(C), *gamma = dalloc(C);
  float *bdy = dalloc(act), *by = dalloc(act), *bx = dalloc(act), *bdx = dalloc(act), *bdx2 = dalloc(act), *dgb = dalloc(2 * C), *dgb2 = dalloc(2 * C);
  fill(x, act, 1 + C, 1.f); fill(dy, act, 2 + C, 1.f); fill(out, act, 3 + C, 1.f, true); fill(w, wn, 4 + C, sqrtf(2.f / (9 * C)));
  fill(edo, act, 7 + C, 1.f); fill(mean, C, 5, 0.1f); fill(invstd, C, 6, 0.1f); fill(gamma, C, 9, 1.f);
  fill(bdy, act, 11 + C, 1.f); fill(by, act, 12 + C, 1.f, true); fill(bx, act, 13 + C, 1.f);
  const int sl = n_img * (HW / 8);
  double *pa1; CK(hipMalloc(&pa1, (size_t)C * sl * 16)); CK(hipMemset(pa1, 0, (size_t)C * sl * 16));
  const size_t scr = (size_t)sgmcmc_conv3x3_wrw_scratch_floats(n_img, C, HW);
  float *part1 = dalloc(scr), *part2 = dalloc(scr);
  hipStream_t s = nullptr;
  conv::BwdEpilogue E{};
  E.s_y = x; E.s_out = out; E.s_mean = mean; E.s_invstd = invstd; E.s_partial = pa1; E.mask_dx = 1; E.e_dout = edo;
  int slabs = 0;
  auto merged = [&] { conv::launch_bwd<C, HW, 8>(x, w, dy, dx1, dw1, part1, n_img, &slabs, s, E, 0); };
  auto dgrad = [&] { conv::launch_bwd<C, HW, 8>(x, w, dy, dx1, dw1, part1, n_img, &slabs, s, E, 1); };
  auto wrw = [&] { conv::launch_bwd<C, HW, 8>(x, w, dy, dx1, dw1, part1, n_img, &slabs, s, E, 2); };
  bn::Geo g; bn::geo(n_img, C, HW * HW, &g, 1);
  const int n_bn = g.C * g.S * g.G;
  auto bnk = [&] { SGMCMC_LAUNCH((bn::bwd_dx_kernel<false, false, false>), dim3(n_bn), dim3(256), 0, s, bdy, by, bx, gamma, mean, invstd, pa1, sl, g, bdx, nullptr, dgb, bn::ResSums{}); };
  const int n_items = n_img * W::BANDS, P = (n_items + W::ITEMS - 1) / W::ITEMS, n_wrw = P * W::CT;
  auto kp = bn_wrw_kernel<C, HW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W::LDS_BYTES));
  auto pair = [&] { SGMCMC_LAUNCH(kp, dim3(n_wrw + n_bn), dim3(256), W::LDS_BYTES, s, n_wrw, x, dy, part2, n_items, bdy, by, bx, gamma, mean, invstd, pa1, sl, g, bdx2, dgb2); };
  merged(); wrw(); bnk(); pair(); CK(hipDeviceSynchronize());
  printf("  pair vs separate: slabs differing words %zu, bn dx differing words %zu (n_wrw %d + n_bn %d blocks, LDS %zu)\n",
         bits_differ(part1, part2, (size_t)P * wn), bits_differ(bdx, bdx2, act), n_wrw, n_bn, (size_t)W::LDS_BYTES);
  timeit("merged dgrad+wrw (today)", merged);
  timeit("dgrad alone (add + sums + mask)", dgrad);
  timeit("wrw alone", wrw);
  timeit("bn bwd_dx alone", bnk);
  timeit("bn bwd_dx || wrw in one launch", pair);
  chain("today:  merged ; bn", [&] { merged(); bnk(); }, 100);
  chain("paired: dgrad ; bn||wrw", [&] { dgrad(); pair(); }, 100);
  {   // L2-cold: rotating activations
    const int R = (int)std::max<size_t>(3, (size_t)(200u << 20) / (9 * act * sizeof(float)));
    std::vector<float*> xs(R), dys(R), outs(R), es(R), b1(R), b2(R), b3(R), o1(R), o2(R);
    for (int r = 0; r < R; ++r) {
      for (auto v : {&xs, &dys, &outs, &es, &b1, &b2, &b3, &o1, &o2}) (*v)[r] = dalloc(act);
      CK(hipMemcpy(xs[r], x, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(dys[r], dy, act * 4, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(outs[r], out, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(es[r], edo, act * 4, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(b1[r], bdy, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(b2[r], by, act * 4, hipMemcpyDeviceToDevice));
      CK(hipMemcpy(b3[r], bx, act * 4, hipMemcpyDeviceToDevice));
    }
    int k = 0;
    printf("  -- rotating over %d operand sets\n", R);
    chain("today:  merged ; bn, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue Ek = E; Ek.s_y = xs[k]; Ek.s_out = outs[k]; Ek.e_dout = es[k];
        conv::launch_bwd<C, HW, 8>(xs[k], w, dys[k], o1[k], dw1, part1, n_img, &slabs, s, Ek, 0);
        SGMCMC_LAUNCH((bn::bwd_dx_kernel<false, false, false>), dim3(n_bn), dim3(256), 0, s, b1[k], b2[k], b3[k], gamma, mean, invstd, pa1, sl, g, o2[k], nullptr, dgb, bn::ResSums{}); }, 100);
    chain("paired: dgrad ; bn||wrw, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue Ek = E; Ek.s_y = xs[k]; Ek.s_out = outs[k]; Ek.e_dout = es[k];
        conv::launch_bwd<C, HW, 8>(xs[k], w, dys[k], o1[k], dw1, part1, n_img, &slabs, s, Ek, 1);
        SGMCMC_LAUNCH(kp, dim3(n_wrw + n_bn), dim3(256), W::LDS_BYTES, s, n_wrw, xs[k], dys[k], part2, n_items, b1[k], b2[k], b3[k], gamma, mean, invstd, pa1, sl, g, o2[k], dgb2); }, 100);
  }
}

int main(int argc, char** argv) {
  const int n_img = argc > 1 ? atoi(argv[1]) : 128;
  run_shape<16, 32>(n_img);
  run_shape<32, 16>(n_img);
  run_shape<64, 8>(n_img);
  return 0;
}
