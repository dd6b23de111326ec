// Micro-benchmark: what does it cost a 512-workgroup launch to leave its per-channel partial sums as INTEGER atomics
// (order-independent, hence bit-reproducible) instead of per-workgroup partial stores?
//   store : 32 doubles per workgroup to its own slot                      (what the BatchNorm statistics do today)
//   xcd   : 64 int64 atomics (2 limbs x 32 sums), workgroup scope, to the slot of the XCD the workgroup runs on
//   agent : the same atomics, agent scope, per-XCD slots / ONE slot set for the device
// and does the per-XCD variant give the exact total?  (HW_REG_XCC_ID names the XCD.)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef PADV
#define PADV 16
#endif
constexpr int NSUM = 32, PAD = PADV;     // 32 sums; PAD 16: every limb in its own 128-byte line, 1: dense

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ float work(const float* __restrict__ x, int n) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += x[(size_t)blockIdx.x * n + i];
  return s;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ x, int n, double* __restrict__ part,
                                         long long* __restrict__ acc, int* __restrict__ seen) {
  __shared__ float sh[256];
  sh[threadIdx.x] = work(x, n);
  __syncthreads();
  if (threadIdx.x < NSUM) {
    double v = 0.0;
    for (int i = 0; i < 8; ++i) v += (double)sh[threadIdx.x * 8 + i];
    if (MODE == 0) {
      part[(size_t)blockIdx.x * NSUM + threadIdx.x] = v;
    } else {
      const double coarse = rint(v * 256.0);                    // multiples of 2^-8
      const long long hi = (long long)coarse;
      const long long lo = (long long)rint((v - coarse / 256.0) * 0x1p60);
      const int slot = MODE == 2 ? 0 : xcc_id();
      long long* a = acc + ((size_t)slot * NSUM + threadIdx.x) * 2 * PAD;
      if (MODE == 1) {
        __hip_atomic_fetch_add(a, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + PAD, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        __hip_atomic_fetch_add(a, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + PAD, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (threadIdx.x == 0 && seen) seen[blockIdx.x] = xcc_id();
    }
  }
}

// the consumer's view: total of sum c over the slots, in slot order
__global__ void total(const long long* __restrict__ acc, int slots, double* __restrict__ out) {
  const int c = threadIdx.x;
  if (c >= NSUM) return;
  long long hi = 0, lo = 0;
  for (int s = 0; s < slots; ++s) { hi += acc[((size_t)s * NSUM + c) * 2 * PAD]; lo += acc[((size_t)s * NSUM + c) * 2 * PAD + PAD]; }
  out[c] = (double)hi / 256.0 + (double)lo * 0x1p-60;
}

template <typename F>
static void timeit(const char* name, F fn, int iters = 200) {
  for (int i = 0; i < 10; ++i) fn(nullptr, nullptr);
  CK(hipDeviceSynchronize());
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int i = 0; i < iters; ++i) fn(ev[2 * i], ev[2 * i + 1]);
  CK(hipDeviceSynchronize());
  double sum = 0; float mn = 1e9, ms;
  for (int i = 0; i < iters; ++i) { CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); sum += ms; mn = std::min(mn, ms); }
  for (auto& e : ev) CK(hipEventDestroy(e));
  printf("  %-44s avg %7.2f us  min %7.2f us\n", name, 1e3 * sum / iters, 1e3 * mn);
}

int main() {
  const int grid = 512;
  for (int n : {256, 4096}) {
    float* x; double *part, *out; long long* acc; int* seen;
    CK(hipMalloc(&x, (size_t)grid * n * 4)); CK(hipMalloc(&part, (size_t)grid * NSUM * 8)); CK(hipMalloc(&out, NSUM * 8));
    CK(hipMalloc(&acc, 16 * NSUM * 2 * PAD * 8)); CK(hipMalloc(&seen, grid * 4));
    std::vector<float> h((size_t)grid * n);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 997.f - 0.5f;
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("grid %d, %d floats per workgroup\n", grid, n);
    timeit("partial stores (32 doubles / workgroup)", [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, b, 0, x, n, part, acc, (int*)nullptr);
      else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, (int*)nullptr); });
    timeit("int64 atomics, per-XCD slots, workgroup scope", [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, b, 0, x, n, part, acc, (int*)nullptr);
      else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, (int*)nullptr); });
    timeit("int64 atomics, per-XCD slots, agent scope", [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, a, b, 0, x, n, part, acc, (int*)nullptr);
      else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, (int*)nullptr); });
    timeit("int64 atomics, one slot set, agent scope", [&](hipEvent_t a, hipEvent_t b) {
      if (a) hipExtLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, b, 0, x, n, part, acc, (int*)nullptr);
      else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, (int*)nullptr); });
    // exactness: per-XCD slots vs agent-scope single slot vs the host's sum of the stored partials, 20 repetitions
    std::vector<double> ref(NSUM), got(NSUM), hp((size_t)grid * NSUM);
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, (int*)nullptr);
    CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
    for (int c = 0; c < NSUM; ++c) { long double s = 0; for (int b = 0; b < grid; ++b) s += hp[(size_t)b * NSUM + c]; ref[c] = (double)s; }
    int bad = 0, differ = 0; std::vector<double> first;
    for (int mode = 1; mode <= 2; ++mode)
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipMemset(acc, 0, 16 * NSUM * 2 * PAD * 8));
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, seen);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, x, n, part, acc, seen);
        hipLaunchKernelGGL(total, dim3(1), dim3(64), 0, 0, acc, 16, out);
        CK(hipMemcpy(got.data(), out, NSUM * 8, hipMemcpyDeviceToHost));
        if (first.empty()) first = got;
        for (int c = 0; c < NSUM; ++c) { if (fabs(got[c] - ref[c]) > 1e-9 * (1 + fabs(ref[c]))) ++bad; if (got[c] != first[c]) ++differ; }
      }
    std::vector<int> hs(grid); CK(hipMemcpy(hs.data(), seen, grid * 4, hipMemcpyDeviceToHost));
    int hist[16] = {0}; for (int b = 0; b < grid; ++b) hist[hs[b] & 15]++;
    printf("  totals off the reference: %d, totals that differ between repetitions / modes: %d; workgroups per XCC_ID:", bad, differ);
    for (int i = 0; i < 16; ++i) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf("\n");
    hipFree(x); hipFree(part); hipFree(out); hipFree(acc); hipFree(seen);
  }
  return 0;
}
