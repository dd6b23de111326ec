// Standalone lab of the UNIFORM backward convolution (csrc/conv_uni_hip.inc) against the merged two-role launch
// (conv::launch_bwd): dx and the BatchNorm-backward sums must be bit-identical, dw agree to summation order; every launch
// timed through its dispatch packet, as a back-to-back chain, and on rotating (L2-cold) operand sets.
//   hipcc --offload-arch=gfx950 -O3 ... lab_uni.hip -o lab_uni ; ./lab_uni [n_img]
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

#ifdef UNI_STAMPS
__device__ unsigned long long g_ustamps[1024 * 16];
#define CONVU_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_ustamps[blockIdx.x * 16 + (k)] = wall_clock64(); } while (0)
#endif
#include "conv_hip.inc"
#include "conv_uni_hip.inc"

static void dump_ustamps(const char* what, int n_wg) {
#ifdef UNI_STAMPS
  std::vector<unsigned long long> t(1024 * 16);
  CK(hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_ustamps), t.size() * 8));
  unsigned long long t0 = ~0ull, tend = 0;
  for (int b = 0; b < n_wg; ++b) { t0 = std::min(t0, t[b * 16]); tend = std::max(tend, t[b * 16 + 8]); }
  double m[9] = {0};
  for (int b = 0; b < n_wg; ++b) for (int k = 0; k < 9; ++k) m[k] += (double)(t[b * 16 + k] - t0) / n_wg;
  printf("  %s stamps, us after the first workgroup's start (mean over %d workgroups): start %.2f | A in LDS %.2f | dgrad half 0 issued %.2f | B in LDS %.2f | "
         "dgrad issued %.2f | x in LDS %.2f | wrw + epilogue issued %.2f | barrier %.2f | end %.2f | last end %.2f\n", what, n_wg,
         m[0] / 100, m[1] / 100, m[2] / 100, m[3] / 100, m[4] / 100, m[5] / 100, m[6] / 100, m[7] / 100, m[8] / 100, (double)(tend - t0) / 100);
#endif
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

// float64 reference of dw on the host for a few entries
template <int C, int HW>
static double dw_ref(const std::vector<float>& x, const std::vector<float>& dy, int n_img, int co, int ci, int r, int s) {
  double acc = 0;
  for (int n = 0; n < n_img; ++n)
    for (int py = 0; py < HW; ++py) for (int px = 0; px < HW; ++px) {
      const int qy = py + r - 1, qx = px + s - 1;
      if (qy < 0 || qy >= HW || qx < 0 || qx >= HW) continue;
      acc += (double)dy[(((size_t)n * C + co) * HW + py) * HW + px] * (double)x[(((size_t)n * C + ci) * HW + qy) * HW + qx];
    }
  return acc;
}

template <int C, int HW>
static void run_shape(int n_img) {
  const size_t act = (size_t)n_img * C * HW * HW, wn = (size_t)C * C * 9;
  using U = convu::Cfg<C, HW>;
  printf("== C=%d HW=%d n=%d  (uniform: %d workgroups x 512 threads, LDS %zu B)\n", C, HW, n_img, convu::n_slabs<C, HW>(n_img), U::LDS_BYTES);
  float *x = dalloc(act), *dy = dalloc(act), *out = dalloc(act), *w = dalloc(wn), *edo = dalloc(act), *eo = dalloc(act);
  float *dx1 = dalloc(act), *dx2 = dalloc(act), *dw1 = dalloc(wn), *dw2 = dalloc(wn), *mean = dalloc(C), *invstd = dalloc(C);
  fill(x, act, 1 + C, 1.f); fill(dy, act, 2 + C, 1.f); fill(out, act, 3 + C, 1.f, true); fill(w, wn, 4 + C, sqrtf(2.f / (9 * C)));
  fill(edo, act, 7 + C, 1.f); fill(eo, act, 8 + C, 1.f, true);
  fill(mean, C, 5, 0.1f); fill(invstd, C, 6, 0.1f);
  const int sl = n_img * (HW / 8);
  double *pa1, *pa2;
  CK(hipMalloc(&pa1, (size_t)C * sl * 16)); CK(hipMalloc(&pa2, (size_t)C * sl * 16));
  CK(hipMemset(pa1, 0, (size_t)C * sl * 16)); CK(hipMemset(pa2, 0xff, (size_t)C * sl * 16));
  const size_t scr = (size_t)sgmcmc_conv3x3_wrw_scratch_floats(n_img, C, HW);
  float *part1 = dalloc(scr), *part2 = dalloc(scr);
  hipStream_t s = nullptr;
  for (int variant = 0; variant < 5; ++variant) {
    const int epi = variant == 4 ? 2 : (variant & 1);      // 2: e_dout added unmasked (what the step does: PREMASK)
    const bool sums = variant & 2 || variant == 4;
    conv::BwdEpilogue E1{}, E2{};
    if (sums) { E1.s_y = x; E1.s_out = out; E1.s_mean = mean; E1.s_invstd = invstd; E1.s_partial = pa1; E1.mask_dx = 1; }
    if (epi) { E1.e_dout = edo; E1.e_out = epi == 1 ? eo : nullptr; }
    E2 = E1; E2.s_partial = sums ? pa2 : nullptr;
    int slabs1 = 0;
    const int slabs2 = convu::n_slabs<C, HW>(n_img);
    auto b1 = [&] { int e = conv::launch_bwd<C, HW, 8>(x, w, dy, dx1, dw1, part1, n_img, &slabs1, s, E1); if (e) { printf("v1 err %d\n", e); exit(1); } };
    auto b2 = [&] { int e = convu::launch<C, HW>(x, w, dy, dx2, part2, n_img, s, E2); if (e) { printf("uni err %d\n", e); exit(1); } };
    auto r1 = [&] { SGMCMC_LAUNCH(conv::wrw_reduce_kernel, dim3(conv::reduce_blocks(slabs1, (int)wn)), dim3(256), 0, s, part1, slabs1, (int)wn, dw1, 9); };
    auto r2 = [&] { SGMCMC_LAUNCH(conv::wrw_reduce_kernel, dim3(conv::reduce_blocks(slabs2, (int)wn)), dim3(256), 0, s, part2, slabs2, (int)wn, dw2, 9); };
    CK(hipMemset(dx2, 0xff, act * 4));
    b1(); r1(); b2(); r2(); CK(hipDeviceSynchronize());
    printf(" variant add=%d sums=%d: dx differing words %zu / %zu", (int)epi, (int)sums, bits_differ(dx1, dx2, act), act);
    if (sums) printf(", sums differing words %zu", bits_differ((float*)pa1, (float*)pa2, (size_t)C * sl * 4));
    {
      auto h1 = host(dw1, wn), h2 = host(dw2, wn);
      double d = 0, sc = 0;
      for (size_t i = 0; i < wn; ++i) { d = std::max(d, (double)fabsf(h1[i] - h2[i])); sc = std::max(sc, (double)fabsf(h1[i])); }
      printf(", max|dw1-dw2| %.3g (scale %.3g)", d, sc);
      if (variant == 0) {
        auto hx = host(x, act), hd = host(dy, act);
        double e1 = 0, e2 = 0;
        const int probes[6][4] = {{0, 0, 0, 0}, {C - 1, C - 1, 2, 2}, {3, 7, 1, 1}, {C / 2, 1, 0, 2}, {5, C - 2, 2, 0}, {C - 1, 0, 1, 2}};
        for (auto& p : probes) {
          const double ref = dw_ref<C, HW>(hx, hd, n_img, p[0], p[1], p[2], p[3]);
          const size_t i = ((size_t)p[0] * C + p[1]) * 9 + p[2] * 3 + p[3];
          e1 = std::max(e1, fabs(h1[i] - ref)); e2 = std::max(e2, fabs(h2[i] - ref));
        }
        printf(", vs float64 on 6 entries: merged %.3g uniform %.3g", e1, e2);
      }
    }
    printf("\n");
    char nm[64];
    snprintf(nm, sizeof nm, "merged  add=%d sums=%d", (int)epi, (int)sums); timeit(nm, b1);
    snprintf(nm, sizeof nm, "uniform add=%d sums=%d", (int)epi, (int)sums); timeit(nm, b2);
    b2(); CK(hipDeviceSynchronize()); dump_ustamps(nm, std::min(1024, slabs2));
    if (variant >= 2) {
      snprintf(nm, sizeof nm, "merged  add=%d sums=%d", (int)epi, (int)sums); chain(nm, b1);
      snprintf(nm, sizeof nm, "uniform add=%d sums=%d", (int)epi, (int)sums); chain(nm, b2);
      timeit("reduce of the merged launch's slabs", r1);
      timeit("reduce of the uniform launch's slabs", r2);
    }
    if (variant == 4) {
      const int R = (int)std::max<size_t>(3, (size_t)(160u << 20) / (6 * act * sizeof(float)));
      std::vector<float*> xs(R), dys(R), outs(R), ys(R), es(R), eos(R);
      for (int r = 0; r < R; ++r) {
        xs[r] = dalloc(act); dys[r] = dalloc(act); outs[r] = dalloc(act); ys[r] = dalloc(act); es[r] = dalloc(act); eos[r] = dalloc(act);
        CK(hipMemcpy(xs[r], x, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(dys[r], dy, act * 4, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(outs[r], out, act * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(es[r], edo, act * 4, hipMemcpyDeviceToDevice));
        CK(hipMemcpy(eos[r], eo, act * 4, hipMemcpyDeviceToDevice));
      }
      int k = 0;
      printf("  -- rotating over %d operand sets: L2-cold launches\n", R);
      chain("merged  add+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = E1; E.s_y = xs[k]; E.s_out = outs[k]; E.e_dout = es[k];
                                          conv::launch_bwd<C, HW, 8>(xs[k], w, dys[k], ys[k], dw1, part1, n_img, &slabs1, s, E); });
      chain("uniform add+sums, cold", [&] { k = (k + 1) % R; conv::BwdEpilogue E = E2; E.s_y = xs[k]; E.s_out = outs[k]; E.e_dout = es[k];
                                          convu::launch<C, HW>(xs[k], w, dys[k], ys[k], part2, n_img, s, E); });
      for (int r = 0; r < R; ++r) for (float* p : {xs[r], dys[r], outs[r], ys[r], es[r], eos[r]}) CK(hipFree(p));
    }
  }
  for (float* p : {x, dy, out, w, edo, eo, dx1, dx2, dw1, dw2, mean, invstd, part1, part2}) CK(hipFree(p));
  CK(hipFree(pa1)); CK(hipFree(pa2));
}

int main(int argc, char** argv) {
  const int n_img = argc > 1 ? atoi(argv[1]) : 128;
  run_shape<16, 32>(n_img);
  run_shape<32, 16>(n_img);
  if (argc > 2) { run_shape<16, 32>(5); run_shape<32, 16>(1); }
  return 0;
}
