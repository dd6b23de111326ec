// sgmcmc_hip.hip -- gfx950 (MI355X / CDNA4) kernels of the SG-MCMC leapfrog engine
// and the extern "C" entry points declared in include/sgmcmc_hip.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared  (see __graft_entry__.build)
// -ffp-contract=off: the element-wise update and the noise transform are SPECIFIED
// operation by operation (DESIGN.md "Arithmetic spec"); every fused multiply-add is an
// explicit fmaf()/fma(), everything else rounds individually, so theta/m/v are
// bit-reproducible against the C oracle.
//
// Kernel shape (HBM-bound streaming pass, 28 B/element for an ordinary fp32 step):
//   grid  = one workgroup per arena chunk of SGMCMC_CHUNK = 4096 elements
//   block = 256 threads = 4 wavefronts of 64; thread t owns items t, t+256, t+512, t+768,
//           an item being 4 consecutive elements = one 16-byte access per array (fp32),
//           so each wave instruction moves a contiguous, aligned 1 KiB.
//   All 16 loads of a thread (4 arrays x 4 items) are issued before the first use.
//   Noise: one Philox4x32-10 call per item yields its 4 normals in registers (0 B of HBM).
//   Dots : exact fp64 products, fp64 accumulate per thread -> wave shuffle reduce ->
//          LDS across the 4 waves in wave order -> one fp64 partial per (chunk, quantity).
//          A second small kernel sums a segment's partials in a fixed order and applies
//          the per-segment energy / temperature bookkeeping.  No float atomics anywhere:
//          results are bitwise run-to-run reproducible and independent of launch timing.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>

#include "sgmcmc_hip.h"

// Every kernel launch of this library goes through SGMCMC_LAUNCH.  Normally that is hipLaunchKernelGGL;
// after sgmcmc_time_next_launch(e0, e1) the NEXT launch carries the two events in its dispatch packet
// (hipExtLaunchKernelGGL), so hipEventElapsedTime(e0, e1) is that kernel's own execution time -- what
// rocprofv3 reports per dispatch -- without the ~3 us a pair of hipEventRecord calls adds around it.
namespace sgmcmc_timing {
static hipEvent_t e0 = nullptr, e1 = nullptr;
}
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

// Activation-sized outputs are stored WRITE-THROUGH (global_store ... sc0 sc1): the lines leave the XCD's L2 while the
// kernel still runs, instead of in the write-back at its end that the next launch of the chain waits for -- measured in
// the googleresnet step (round 5, tools/ab_table.sh): forward convolutions -0.9 us per launch.  Their consumer is the NEXT
// launch, on whatever XCD: it never could count on finding the line in its own L2.  SGMCMC_WT_STORES is a mask of kernel
// families (A/B builds): 1 trunk convolutions, 2 BatchNorm, 4 down-sampling / stem convolutions (forward), 8 head / pooling,
// 16 the 50-channel classifier's layers, 32 persistent convolutions, 64 the trunk's data gradients (measured +0.3 us per
// merged backward launch: off; the down-sampling block's data gradient +1 us: always plain).
#ifndef SGMCMC_WT_STORES
#define SGMCMC_WT_STORES 15
#endif
using sgmcmc_f32x4 = __attribute__((ext_vector_type(4))) float;
// `base`: the tensor's base pointer as the kernel received it (uniform: the buffer descriptor lives in scalar registers);
// `p`: this lane's 16-byte aligned element pointer inside that tensor.  The store goes through the compiler's own buffer
// -store builtin (cache policy sc0 sc1), NOT through inline assembly: an asm store is invisible to the hazard recogniser
// -- round 5's first A/B build stored accumulators that the matrix pipe had not written back yet (tools/lab/conv_bits.py).
template <int FAMILY>
__device__ __forceinline__ void sgmcmc_store4(float* __restrict__ base, float* __restrict__ p, float a, float b, float c, float d) {
  if constexpr ((SGMCMC_WT_STORES & FAMILY) != 0) {
    const uint64_t off = (uint64_t)(reinterpret_cast<char*>(p) - reinterpret_cast<char*>(base));
    if (off < 0xfffffff0ull) {           // (a descriptor addresses 4 GiB; beyond it: a plain store)
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(sgmcmc_f32x4{a, b, c, d}, r, (int)(uint32_t)off, 0, /*sc0 | sc1*/ 17);
      return;
    }
  }
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// Every entry point reports ITS launches through hipGetLastError().  That state is process-wide and
// sticky: a failed pointer query inside another library (ATen's pinned-memory checks leave
// hipErrorInvalidValue behind) would otherwise surface here as if one of our launches had failed.
#define SGMCMC_FRESH_ERROR_STATE() (void)hipGetLastError()

namespace {

constexpr int kThreads = 256;
constexpr int kItemElems = kThreads * 4;  // elements covered by one item of every thread (1024)
static_assert(SGMCMC_CHUNK % kItemElems == 0 && SGMCMC_CHUNK_SMALL == kItemElems, "chunk geometry");
__device__ __forceinline__ int items_of(const sgmcmc_layout& L) { return (int)(L.chunk_elems / kItemElems); }

// ------------------------------------------------------------------ noise
// Philox4x32-10; counter layout and transforms: DESIGN.md "Noise".
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float spec_logf(float u) {
  const uint32_t bits = __float_as_uint(u);
  int e = (int)(bits >> 23) - 126;
  const float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F000000u);
  float x;
  if (m < 0.70710678f) { e -= 1; x = (m + m) - 1.0f; } else { x = m - 1.0f; }
  const float z = x * x;
  float p = 7.0376836292E-2f;
  p = fmaf(p, x, -1.1514610310E-1f);
  p = fmaf(p, x, 1.1676998740E-1f);
  p = fmaf(p, x, -1.2420140846E-1f);
  p = fmaf(p, x, 1.4249322787E-1f);
  p = fmaf(p, x, -1.6668057665E-1f);
  p = fmaf(p, x, 2.0000714765E-1f);
  p = fmaf(p, x, -2.4999993993E-1f);
  p = fmaf(p, x, 3.3333331174E-1f);
  float y = (p * x) * z;
  const float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(z, -0.5f, y);
  float r = x + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

__device__ __forceinline__ void spec_sincos2pi(uint32_t k23, float& s_out, float& c_out) {
  const uint32_t odd = 2u * k23 + 1u;
  const uint32_t q = (odd + (1u << 21)) >> 22;
  const int32_t ri = (int32_t)odd - (int32_t)(q << 22);
  const float r = (float)ri * 5.9604644775390625e-08f;
  const float phi = r * 6.2831855f;
  const float z = phi * phi;
  float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float s = fmaf(ps * z, phi, phi);
  float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float c = fmaf(pc, z * z, fmaf(z, -0.5f, 1.0f));
  const uint32_t qq = q & 3u;
  s_out = (qq == 0) ? s : (qq == 1) ? c : (qq == 2) ? -s : -c;
  c_out = (qq == 0) ? c : (qq == 1) ? -s : (qq == 2) ? -c : s;
}

__device__ __forceinline__ float spec_uniform(uint32_t x) {
  return (float)(2u * (x >> 9) + 1u) * 5.9604644775390625e-08f;
}

__device__ __forceinline__ void spec_normal4(uint64_t seed, uint32_t stream, uint64_t draw,
                                             uint32_t purpose, uint64_t quad, float (&z)[4]) {
  uint32_t x[4];
  philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), (uint32_t)draw,
                (purpose << 28) | ((stream & 0xFFFu) << 16) | (uint32_t)((draw >> 32) & 0xFFFFu),
                (uint32_t)seed, (uint32_t)(seed >> 32), x);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float t = -2.0f * spec_logf(spec_uniform(x[2 * h]));
    t = t < 0.0f ? 0.0f : t;
    const float rad = sqrtf(t);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    float s, c;
    spec_sincos2pi(x[2 * h + 1] >> 9, s, c);
    z[2 * h] = rad * c;
    z[2 * h + 1] = rad * s;
  }
}

// ------------------------------------------------------------------ helpers
template <typename T> __device__ __forceinline__ T fma_t(T a, T b, T c);
template <> __device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return fmaf(a, b, c); }
template <> __device__ __forceinline__ double fma_t<double>(double a, double b, double c) { return fma(a, b, c); }

template <typename T> struct Item { T x[4]; };

template <typename T> __device__ __forceinline__ Item<T> load_item(const T* __restrict__ p);
template <> __device__ __forceinline__ Item<float> load_item<float>(const float* __restrict__ p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return Item<float>{{v.x, v.y, v.z, v.w}};
}
template <> __device__ __forceinline__ Item<double> load_item<double>(const double* __restrict__ p) {
  const double2 a = *reinterpret_cast<const double2*>(p);
  const double2 b = *reinterpret_cast<const double2*>(p + 2);
  return Item<double>{{a.x, a.y, b.x, b.y}};
}
template <typename T> __device__ __forceinline__ void store_item(T* __restrict__ p, const Item<T>& v);
template <> __device__ __forceinline__ void store_item<float>(float* __restrict__ p, const Item<float>& v) {
  *reinterpret_cast<float4*>(p) = make_float4(v.x[0], v.x[1], v.x[2], v.x[3]);
}
template <> __device__ __forceinline__ void store_item<double>(double* __restrict__ p, const Item<double>& v) {
  *reinterpret_cast<double2*>(p) = make_double2(v.x[0], v.x[1]);
  *reinterpret_cast<double2*>(p + 2) = make_double2(v.x[2], v.x[3]);
}
// streaming (non-temporal) forms: each byte of a big arena is touched once per launch, so it
// should not displace anything in L2 / Infinity Cache
template <typename T> __device__ __forceinline__ Item<T> load_item_nt(const T* __restrict__ p);
template <> __device__ __forceinline__ Item<float> load_item_nt<float>(const float* __restrict__ p) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
  return Item<float>{{v[0], v[1], v[2], v[3]}};
}
template <> __device__ __forceinline__ Item<double> load_item_nt<double>(const double* __restrict__ p) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 a = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
  const d2 b = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + 2));
  return Item<double>{{a[0], a[1], b[0], b[1]}};
}
template <typename T> __device__ __forceinline__ void store_item_nt(T* __restrict__ p, const Item<T>& v);
template <> __device__ __forceinline__ void store_item_nt<float>(float* __restrict__ p, const Item<float>& v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 x = {v.x[0], v.x[1], v.x[2], v.x[3]};
  __builtin_nontemporal_store(x, reinterpret_cast<f4*>(p));
}
template <> __device__ __forceinline__ void store_item_nt<double>(double* __restrict__ p, const Item<double>& v) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  d2 a = {v.x[0], v.x[1]}, b = {v.x[2], v.x[3]};
  __builtin_nontemporal_store(a, reinterpret_cast<d2*>(p));
  __builtin_nontemporal_store(b, reinterpret_cast<d2*>(p + 2));
}

template <typename T>
__device__ __forceinline__ Item<T> load_guarded(const T* __restrict__ p, int n) {
  Item<T> r;
#pragma unroll
  for (int l = 0; l < 4; ++l) r.x[l] = l < n ? p[l] : T(0);
  return r;
}
template <typename T>
__device__ __forceinline__ void store_guarded(T* __restrict__ p, const Item<T>& v, int n) {
#pragma unroll
  for (int l = 0; l < 4; ++l)
    if (l < n) p[l] = v.x[l];
}

template <typename T> __device__ __forceinline__ T clamp_grad(T g, T c) {
  // torch.clamp semantics: NaN stays NaN (inference.py:219-220)
  return g < -c ? -c : (g > c ? c : g);
}

// Deterministic block reduction of NS doubles: wave shuffle tree, then the 4 wave
// results summed in wave order by one thread per quantity.
template <int NS>
__device__ __forceinline__ void block_reduce_store(double (&acc)[NS], double* __restrict__ out) {
  __shared__ double sh[kThreads / 64][NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double x = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    acc[k] = x;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS; ++k) sh[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < NS) {
    double s = sh[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) s += sh[w][threadIdx.x];
    out[threadIdx.x] = s;
  }
}

struct ChunkCtx {
  int32_t seg, n_valid;
  int64_t seg_off, arena_off;
};
__device__ __forceinline__ ChunkCtx chunk_ctx(const sgmcmc_layout& L, int64_t chunk, const sgmcmc_chunk ce) {
  ChunkCtx c;
  c.seg = ce.seg;
  c.n_valid = ce.n_valid;
  c.seg_off = (chunk - L.segs[ce.seg].first_chunk) * L.chunk_elems;
  c.arena_off = chunk * L.chunk_elems;
  return c;
}
__device__ __forceinline__ ChunkCtx chunk_ctx(const sgmcmc_layout& L, int64_t chunk) {
  return chunk_ctx(L, chunk, L.chunks[chunk]);
}

// ------------------------------------------------------------------ step kernel
// One element of one transition.  KIND selects the integrator; see the arithmetic
// spec in DESIGN.md and the reference lines cited in include/sgmcmc_hip.h.
template <typename T, int KIND>
struct Coef {
  T grad_lr, stepsz, mom_decay, noise_std, alpha, one_m_alpha, clampv;
  bool has_decay, has_noise, do_clamp, is_final, no_mom;
};

template <typename T, int KIND>
__device__ __forceinline__ void update_elem(const Coef<T, KIND>& C, T xi, T gi, T mo, T th, T vv,
                                            T& mn, T& th_new, T& v_new, double (&acc)[SGMCMC_NSUMS]) {
  T mt = mo;  // the momentum the kinetic-temperature metric is taken from (sums[3])
  if (KIND == SGMCMC_VERLET) {
    mn = xi * C.noise_std;                              // verlet_sgld.py:163
    mn = fma_t<T>(gi, C.grad_lr, mn);                   // :164-165
    if (C.has_decay) mn = fma_t<T>(mo, C.mom_decay, mn);  // :166-167
  } else if (KIND == SGMCMC_HMC) {
    mn = fma_t<T>(gi, C.grad_lr, mo);                   // hmc.py:64-65
  } else {
    if (C.no_mom) { mn = gi * C.grad_lr; mt = mn; }     // sgld.py:134
    else { mn = mo * C.mom_decay; mn = fma_t<T>(gi, C.grad_lr, mn); }  // sgld.py:131
    if (C.has_noise) mn = fma_t<T>(xi, C.noise_std, mn);  // sgld.py:142
  }
  const double gd = (double)gi, mod = (double)mo, mnd = (double)mn, mtd = (double)mt;
  acc[0] = fma(gd, gd, acc[0]);
  acc[1] = fma(gd, mod, acc[1]);
  acc[2] = fma(gd, mnd, acc[2]);
  acc[3] = fma(mtd, mtd, acc[3]);
  acc[4] = fma(mnd, mnd, acc[4]);
  acc[5] = fma((double)th, gd, acc[5]);
  th_new = fma_t<T>(mn, C.stepsz, th);                  // verlet_sgld.py:192-193
  v_new = vv * C.alpha + (C.one_m_alpha * gi) * gi;     // :195-197
}

// Optional in-flight assembly of the gradient (fused dense path): likelihood part as per-slice
// partials, prior part in closed form.  `gpart == nullptr`: g is taken from the segment as is.
struct GradParts {
  const float* gpart;       // [n_slices][stride], element (seg, j) at noise_base_seg + j
  int n_slices;
  int64_t stride;
  const float* loss_part;   // [n_slices]
  const float* correct_part;
  int batch;
  double num_data;
};

__device__ __forceinline__ double prior_log_norm(const sgmcmc_segment& s, double scale);  // defined with the prior kernels

// value x of a hierarchical scale given the raw hyper-parameter s (see SGMCMC_PRIOR_* in the header) and dx/ds
// (out of line: rare path with heavy double-precision math; inlined it costs every caller ~80 VGPRs)
__device__ __attribute__((noinline)) double hyper_value(const sgmcmc_segment& h, double s, double& dxds) {
  const double sig = 1.0 / (1.0 + exp(-s)), sp = s > 30.0 ? s : log1p(exp(s));
  if (h.prior_kind == SGMCMC_PRIOR_GAMMA_SOFTPLUS || h.prior_kind == SGMCMC_PRIOR_IMPROPER_SOFTPLUS) { dxds = sig; return sp; }
  if (h.prior_kind == SGMCMC_PRIOR_HALFCAUCHY_SOFTPLUS) { dxds = sig * h.prior_loc; return sp * h.prior_loc; }
  // UNIFORM_CDF: low + (high - low) Phi(s)
  const double w = h.prior_scale - h.prior_loc;
  dxds = w * 0.3989422804014326779 * exp(-0.5 * s * s);
  return h.prior_loc + w * 0.5 * erfc(-s * 0.7071067811865475244);
}

__device__ __attribute__((noinline)) double pow_noinline(double a, double b) { return pow(a, b); }

// d log p / ds and log p (normalised) of a hyper-prior kind at raw parameter s; loc / scale as in the header
__device__ __attribute__((noinline)) double hyper_prior_dlogp(int kind, double loc, double scale, double s, double& l) {
  const double sig = 1.0 / (1.0 + exp(-s)), x = s > 30.0 ? s : log1p(exp(s));
  if (kind == SGMCMC_PRIOR_GAMMA_SOFTPLUS) {             // loc = concentration, scale = rate
    l = loc * log(scale) - lgamma(loc) + (loc - 1.0) * log(x) - scale * x;
    return ((loc - 1.0) / x - scale) * sig;
  }
  if (kind == SGMCMC_PRIOR_IMPROPER_SOFTPLUS) { l = 0.0; return 0.0; }      // no density of its own
  if (kind == SGMCMC_PRIOR_HALFCAUCHY_SOFTPLUS) {        // loc = multiplier, scale = gamma
    const double z = x * loc / scale;
    l = log(2.0 / (3.14159265358979323846 * scale)) - log1p(z * z);
    return -2.0 * z / (1.0 + z * z) * (loc / scale) * sig;
  }
  l = -log(scale - loc);                                 // UNIFORM_CDF: constant density
  return 0.0;
}

template <typename T>
__device__ __forceinline__ double linked_scale(const sgmcmc_layout& L, const sgmcmc_segment* sp, double& dxds) {
  dxds = 0.0;
  if (sp->scale_link <= 0) return sp->prior_scale;
  const sgmcmc_segment& h = L.segs[sp->scale_link - 1];
  return hyper_value(h, (double)*(const T*)h.theta, dxds);
}

// the scale a segment's prior is evaluated with right now: its own constant, or the hyper segment's value
__device__ __forceinline__ double current_scale(const sgmcmc_layout& L, const sgmcmc_segment& s) {
  if (s.scale_link <= 0) return s.prior_scale;
  double dxds;
  return L.dtype == SGMCMC_F32 ? linked_scale<float>(L, &s, dxds) : linked_scale<double>(L, &s, dxds);
}

// d/dtheta[-log p(theta)/N] added to g, and (optionally) the un-normalised log-density into lp and
// d/dscale[log p(theta)] into dls (hierarchical scales)
template <typename T>
struct PriorCoef {
  int kind;
  bool linked;
  double loc, scale, df;
  T locT, c_normal, c_laplace, c_t_num, c_t_den, c_gn, inv_scaleT, beta_m1;
  template <bool FULL = true>
  __device__ __forceinline__ void init(const sgmcmc_layout& L, const sgmcmc_segment* sp, double num_data) {
    kind = sp->prior_kind;
    loc = sp->prior_loc;
    double dxds;
    linked = FULL && sp->scale_link > 0;
    scale = FULL ? linked_scale<T>(L, sp, dxds) : sp->prior_scale;
    df = kind == SGMCMC_PRIOR_CAUCHY ? 1.0 : sp->prior_df;
    locT = (T)loc;
    c_normal = (T)(1.0 / (scale * scale * num_data));
    c_laplace = (T)(1.0 / (scale * num_data));
    c_t_num = (T)((df + 1.0) / num_data);
    c_t_den = (T)(df * scale * scale);
    c_gn = (T)(df / (scale * num_data));   // GENNORM: df holds beta
    c_hyper = 1.0 / num_data;
    inv_scaleT = (T)(1.0 / scale);
    beta_m1 = (T)(df - 1.0);
  }
  // FULL = false (the fused dense step kernel): only the four constant-scale families of BASELINE's configs;
  // the host keeps models with other priors off that path (FusedDenseLeapfrog.supported)
  template <bool FULL = true>
  __device__ __forceinline__ T apply(T g, T th, bool calc_logp, double& lp, double& dls) const {
    if (kind == SGMCMC_PRIOR_NONE) return g;
    if (FULL && kind >= SGMCMC_PRIOR_GAMMA_SOFTPLUS) {
      // hyper-priors: th is the raw parameter s; everything in double (one element per tensor)
      double l;
      const double dlp = hyper_prior_dlogp(kind, loc, scale, (double)th, l);
      if (calc_logp) lp += l;
      return (T)((double)g - dlp * c_hyper);
    }
    const T d = th - locT;
    if (kind == SGMCMC_PRIOR_NORMAL) {
      g = fma_t<T>(d, c_normal, g);
    } else if (kind == SGMCMC_PRIOR_LAPLACE) {
      const T sgn = d > T(0) ? T(1) : (d < T(0) ? T(-1) : T(0));
      g = fma_t<T>(sgn, c_laplace, g);
    } else if (FULL && kind == SGMCMC_PRIOR_GENNORM) {
      const T sgn = d > T(0) ? T(1) : (d < T(0) ? T(-1) : T(0));
      const T z = (d < T(0) ? -d : d) * inv_scaleT;
      g = fma_t<T>(sgn * (T)pow_noinline((double)z, (double)beta_m1), c_gn, g);
    } else {  // Student-t (df) and Cauchy (= Student-t with df = 1, set by PriorCoef::init)
      g = fma_t<T>(d, c_t_num / fma_t<T>(d, d, c_t_den), g);
    }
    if (calc_logp || (FULL && linked)) {
      const double dd = (double)th - loc, z = dd / scale;
      if (kind == SGMCMC_PRIOR_NORMAL) { lp += -0.5 * z * z; dls += (z * z - 1.0) / scale; }
      else if (kind == SGMCMC_PRIOR_LAPLACE) { lp += -fabs(z); dls += (fabs(z) - 1.0) / scale; }
      else if (FULL && kind == SGMCMC_PRIOR_GENNORM) lp += -pow_noinline(fabs(z), df);
      else { lp += -0.5 * (df + 1.0) * log1p(z * z / df); dls += ((df + 1.0) * z * z / (df + z * z) - 1.0) / scale; }
    }
    return g;
  }
  double c_hyper;  // 1/N for the hyper-prior kinds
};

// fixed-order sum of the slices' partial gradients for the 4 elements at packed offset q
template <typename T>
__device__ __forceinline__ Item<T> sum_parts(const float* __restrict__ q, int n_slices, int64_t stride,
                                             int n) {
  Item<T> g{{T(0), T(0), T(0), T(0)}};
  if (n == 4) {  // partial rows are 16-byte aligned (packed offsets are multiples of 4)
#pragma unroll 8
    for (int sl = 0; sl < n_slices; ++sl) {
      const float4 v = *reinterpret_cast<const float4*>(q + (int64_t)sl * stride);
      g.x[0] += (T)v.x; g.x[1] += (T)v.y; g.x[2] += (T)v.z; g.x[3] += (T)v.w;
    }
  } else {
    for (int sl = 0; sl < n_slices; ++sl) {
#pragma unroll
      for (int l = 0; l < 4; ++l)
        if (l < n) g.x[l] += (T)q[(int64_t)sl * stride + l];
    }
  }
  return g;
}

// (every thread of the workgroup calls; workgroup 0 works)  The slices' values are fetched by as many threads at once and
// added by thread 0 in slice order from LDS -- the same sums as a serial walk, without its n_slices dependent trips to
// L2 (32 x ~75 ns at the end of the ONE workgroup that the launch then waits for).
__device__ __forceinline__ void publish_batch_stats(const sgmcmc_layout& L, const GradParts& G) {
  if (blockIdx.x != 0) return;
  __shared__ float lp[2][kThreads];
  const int t = threadIdx.x;
  if (t < G.n_slices) { lp[0][t] = G.loss_part[t]; lp[1][t] = G.correct_part[t]; }
  __syncthreads();
  if (t == 0) {
    double l = 0.0, c = 0.0;
    const int m = G.n_slices < kThreads ? G.n_slices : kThreads;
    for (int sl = 0; sl < m; ++sl) { l += (double)lp[0][sl]; c += (double)lp[1][sl]; }
    for (int sl = m; sl < G.n_slices; ++sl) { l += (double)G.loss_part[sl]; c += (double)G.correct_part[sl]; }
    L.scalars[4] = l / (double)G.batch;
    L.scalars[5] = c / (double)G.batch;
  }
}

// PRIOR (without PARTS): g is autograd's gradient of the likelihood term; the closed-form prior gradient is added in
// flight (SGMCMC_INLINE_PRIOR) instead of by a prior_kernel launch before this one -- as PARTS has always done.
// What a graph-replay kernel has in flight BEFORE it knows its scalars (step_early): the chunk's table entries and,
// on the vector path, its items of m and v -- their addresses follow from the workgroup index alone (the arenas are
// padded to whole chunks, so the loads are in bounds for a ragged chunk too; its values are then not used).
template <typename T, int ITEMS>
struct Early {
  int64_t chunk;
  sgmcmc_chunk ce;
  Item<T> m[ITEMS], v[ITEMS];
};
// "All of these are needed HERE": the compiler otherwise sinks each scalar load of a table entry to its first use and
// waits for them one group at a time -- five dependent round trips to a cold L2 in a kernel whose arithmetic takes
// a microsecond.  An empty asm that names the values makes them one batch of loads and one wait.
__device__ __forceinline__ void needed_here(const sgmcmc_step_args& A) {
  asm volatile("" ::"s"(A.kind), "s"(A.flags), "s"(A.seg_begin), "s"(A.seg_end), "s"(A.chunk_begin), "s"(A.chunk_end),
               "s"(A.num_data), "s"(A.b2h2), "s"(A.bh), "s"(A.bhn), "s"(A.mom_decay), "s"(A.grad_v), "s"(A.noise_std),
               "s"(A.rmsprop_alpha), "s"(A.grad_clamp), "s"(A.seed), "s"(A.draw), "s"(A.stream));
}
__device__ __forceinline__ void needed_here(const sgmcmc_segment& S) {
  asm volatile("" ::"s"(S.theta), "s"(S.g), "s"(S.M), "s"(S.numel), "s"(S.first_chunk), "s"(S.noise_base),
               "s"(S.prior_kind), "s"(S.scale_link), "s"(S.prior_loc), "s"(S.prior_scale), "s"(S.prior_df));
}

template <typename T, int KIND, bool VEC, int ITEMS, bool PARTS, bool STREAM = false, bool PRIOR = false,
          bool EARLY = false>
__device__ __forceinline__ void step_body(const sgmcmc_layout& L, const sgmcmc_step_args& A,
                                          const GradParts& G, const Early<T, ITEMS>* E = nullptr) {
  const int64_t chunk = EARLY ? E->chunk : A.chunk_begin + blockIdx.x;
  const sgmcmc_chunk ce = EARLY ? E->ce : L.chunks[chunk];
  sgmcmc_segment S;
  if (EARLY) {
    S = L.segs[ce.seg];
    needed_here(S);
  }
  const sgmcmc_segment* sp = EARLY ? &S : &L.segs[ce.seg];
  ChunkCtx cx;
  cx.seg = ce.seg;
  cx.n_valid = ce.n_valid;
  cx.seg_off = (chunk - sp->first_chunk) * L.chunk_elems;
  cx.arena_off = chunk * L.chunk_elems;
  const double M = sp->M;
  if (!PARTS && sp->g == nullptr) {
    // raise_on_no_grad=False: a tensor without gradient is left untouched -- parameter, momentum,
    // square_avg and its running scalars (the reference's `continue`, sgld.py:96-100)
    if (threadIdx.x < SGMCMC_NSUMS) L.partials[chunk * SGMCMC_PSTRIDE + threadIdx.x] = 0.0;
    return;
  }

  Coef<T, KIND> C;
  C.grad_lr = (KIND == SGMCMC_SGLD) ? (T)(-A.bhn * M) : (T)(-.5 * A.grad_v * A.bhn * M);
  C.stepsz = (T)(A.bh * M);
  C.mom_decay = (T)A.mom_decay;
  C.noise_std = (T)A.noise_std;
  C.alpha = (T)A.rmsprop_alpha;
  C.one_m_alpha = (T)(1 - A.rmsprop_alpha);
  C.clampv = (T)A.grad_clamp;
  C.has_decay = A.mom_decay > 0;
  C.has_noise = A.noise_std > 0;
  C.do_clamp = A.grad_clamp > 0;
  C.is_final = (A.flags & SGMCMC_FINAL) != 0;
  C.no_mom = (A.flags & SGMCMC_NO_MOMENTUM) != 0;
  const bool save = (A.flags & SGMCMC_SAVE_STATE) != 0;
  // SGLD's final step modifies nothing (sgld.py:80-85); Verlet/HMC's writes m only.
  const bool write_m = !C.no_mom && !(KIND == SGMCMC_SGLD && C.is_final);
  const bool draw_noise = (KIND != SGMCMC_HMC) && C.has_noise && !(KIND == SGMCMC_SGLD && C.is_final);
  constexpr bool WITH_PRIOR = PARTS || PRIOR;
  const bool calc_logp = WITH_PRIOR && (A.flags & SGMCMC_CALC_METRICS);

  T* __restrict__ gp = (T*)sp->g + cx.seg_off;
  T* __restrict__ thp = (T*)sp->theta + cx.seg_off;
  T* __restrict__ mp = (T*)L.m + cx.arena_off;
  T* __restrict__ vp = (T*)L.v + cx.arena_off;
  T* __restrict__ pth = (T*)L.prev_theta + cx.arena_off;
  T* __restrict__ pg = (T*)L.prev_g + cx.arena_off;
  T* __restrict__ pm = (T*)L.prev_m + cx.arena_off;
  const uint64_t noise0 = (uint64_t)(sp->noise_base + cx.seg_off);
  const float* __restrict__ pp = PARTS ? G.gpart + sp->noise_base + cx.seg_off : nullptr;
  PriorCoef<T> PC;
  if (WITH_PRIOR) PC.template init<false>(L, sp, PARTS ? G.num_data : A.num_data);
  double dls_unused = 0.0;

  double acc[SGMCMC_NSUMS] = {0, 0, 0, 0, 0, 0};
  double lp = 0.0;

  if (VEC && cx.n_valid == ITEMS * kItemElems) {
    // full chunk: issue every load first, then compute
    Item<T> g[ITEMS], m[ITEMS], th[ITEMS], v[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = (it * kThreads + threadIdx.x) * 4;
      if (PARTS) g[it] = sum_parts<T>(pp + j, G.n_slices, G.stride, 4);
      else g[it] = STREAM ? load_item_nt<T>(gp + j) : load_item<T>(gp + j);
      th[it] = STREAM ? load_item_nt<T>(thp + j) : load_item<T>(thp + j);
      if (EARLY) {
        m[it] = C.no_mom ? Item<T>{{T(0), T(0), T(0), T(0)}} : E->m[it];
        v[it] = C.is_final ? Item<T>{{T(0), T(0), T(0), T(0)}} : E->v[it];
        continue;
      }
      if (!C.no_mom) m[it] = STREAM ? load_item_nt<T>(mp + j) : load_item<T>(mp + j);
      else m[it] = Item<T>{{T(0), T(0), T(0), T(0)}};
      if (!C.is_final) v[it] = STREAM ? load_item_nt<T>(vp + j) : load_item<T>(vp + j);
      else v[it] = Item<T>{{T(0), T(0), T(0), T(0)}};
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = (it * kThreads + threadIdx.x) * 4;
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (draw_noise) spec_normal4(A.seed, A.stream, A.draw, 0u, (noise0 + (uint64_t)j) >> 2, z);
      Item<T> mn, tn, vn;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        if (WITH_PRIOR) g[it].x[l] = PC.template apply<false>(g[it].x[l], th[it].x[l], calc_logp, lp, dls_unused);
        if (C.do_clamp) g[it].x[l] = clamp_grad<T>(g[it].x[l], C.clampv);
        update_elem<T, KIND>(C, (T)z[l], g[it].x[l], m[it].x[l], th[it].x[l], v[it].x[l], mn.x[l],
                             tn.x[l], vn.x[l], acc);
      }
      if (WITH_PRIOR) store_item<T>(gp + j, g[it]);  // p.grad holds the full gradient afterwards
      if (save) {
        store_item<T>(pth + j, th[it]);
        store_item<T>(pg + j, g[it]);
        if (!C.no_mom) store_item<T>(pm + j, m[it]);
      }
      if (write_m) { if (STREAM) store_item_nt<T>(mp + j, mn); else store_item<T>(mp + j, mn); }
      if (!C.is_final) {
        if (STREAM) { store_item_nt<T>(thp + j, tn); store_item_nt<T>(vp + j, vn); }
        else { store_item<T>(thp + j, tn); store_item<T>(vp + j, vn); }
      }
    }
  } else {
    // ragged tail chunk of a segment, or unaligned base pointers: guarded scalar accesses
    for (int it = 0; it < ITEMS; ++it) {
      const int j = (it * kThreads + threadIdx.x) * 4;
      const int n = cx.n_valid - j < 4 ? cx.n_valid - j : 4;
      if (n <= 0) break;
      Item<T> g = PARTS ? sum_parts<T>(pp + j, G.n_slices, G.stride, n) : load_guarded<T>(gp + j, n);
      Item<T> th = load_guarded<T>(thp + j, n);
      Item<T> m = C.no_mom ? Item<T>{{T(0), T(0), T(0), T(0)}} : load_guarded<T>(mp + j, n);
      Item<T> v = C.is_final ? Item<T>{{T(0), T(0), T(0), T(0)}} : load_guarded<T>(vp + j, n);
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (draw_noise) spec_normal4(A.seed, A.stream, A.draw, 0u, (noise0 + (uint64_t)j) >> 2, z);
      Item<T> mn, tn, vn;
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        if (l < n) {
          if (WITH_PRIOR) g.x[l] = PC.template apply<false>(g.x[l], th.x[l], calc_logp, lp, dls_unused);
          if (C.do_clamp) g.x[l] = clamp_grad<T>(g.x[l], C.clampv);
          update_elem<T, KIND>(C, (T)z[l], g.x[l], m.x[l], th.x[l], v.x[l], mn.x[l], tn.x[l],
                               vn.x[l], acc);
        }
      }
      if (WITH_PRIOR) store_guarded<T>(gp + j, g, n);
      if (save) {
        store_guarded<T>(pth + j, th, n);
        store_guarded<T>(pg + j, g, n);
        if (!C.no_mom) store_guarded<T>(pm + j, m, n);
      }
      if (write_m) store_guarded<T>(mp + j, mn, n);
      if (!C.is_final) {
        store_guarded<T>(thp + j, tn, n);
        store_guarded<T>(vp + j, vn, n);
      }
    }
  }
  if (WITH_PRIOR) {
    double a7[SGMCMC_NSUMS + 1] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], lp};
    block_reduce_store<SGMCMC_NSUMS + 1>(a7, L.partials + chunk * SGMCMC_PSTRIDE);
    if (PARTS) publish_batch_stats(L, G);
  } else {
    block_reduce_store<SGMCMC_NSUMS>(acc, L.partials + chunk * SGMCMC_PSTRIDE);
  }
}

// At BASELINE sizes (314 chunks of 1024 elements) the kernel is a chain of dependent round trips to memory that the
// kernel boundary has just made cold -- scalars -> chunk table -> segment table -> operands -- not a bandwidth problem.
// Everything whose address follows from the workgroup index alone is therefore requested up front, next to the scalars
// (which a graph-replay kernel reads from memory; `chunk_begin` is the host copy's, fixed when the graph is captured):
// the chunk's table entry and its items of m and v.  The segment's entry is then ONE batch of loads (needed_here);
// theta and g follow it.  Measured in the googleresnet step: 8.7 -> 7.4-7.9 us.  (A per-chunk pointer
// table, which lets theta and g leave one trip earlier still, measured no further gain and was not kept.)
template <typename T, bool VEC, int ITEMS>
__device__ __forceinline__ Early<T, ITEMS> step_early(const sgmcmc_layout& L, int64_t chunk_begin) {
  const int64_t chunk = chunk_begin + blockIdx.x;
  Early<T, ITEMS> E;
  E.chunk = chunk;
  if (VEC) {
    const T* __restrict__ mp = (const T*)L.m + chunk * L.chunk_elems;
    const T* __restrict__ vp = (const T*)L.v + chunk * L.chunk_elems;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int j = (it * kThreads + threadIdx.x) * 4;
      E.m[it] = load_item<T>(mp + j);
      E.v[it] = load_item<T>(vp + j);
    }
  }
  E.ce = L.chunks[chunk];
  return E;
}
template <typename T, int KIND, bool VEC, int ITEMS>
__global__ __launch_bounds__(kThreads) void step_kernel(sgmcmc_layout L, sgmcmc_step_args A) {
  const GradParts none = {nullptr, 0, 0, nullptr, nullptr, 1, 1.0};
  const Early<T, ITEMS> E = step_early<T, VEC, ITEMS>(L, A.chunk_begin);
  step_body<T, KIND, VEC, ITEMS, false, false, false, VEC>(L, A, none, &E);
}
// arenas beyond the 256 MiB Infinity Cache: every byte is touched once per launch, so the loads
// and stores are non-temporal (+10 % measured at 2^26..2^28 elements; -9 % if it would have fit)
template <typename T, int KIND>
__global__ __launch_bounds__(kThreads) void step_kernel_stream(sgmcmc_layout L, sgmcmc_step_args A) {
  const GradParts none = {nullptr, 0, 0, nullptr, nullptr, 1, 1.0};
  step_body<T, KIND, true, 4, false, true>(L, A, none);
}
// scalars fetched from device memory at run time (graph replay)
template <typename T, int KIND, bool VEC, int ITEMS>
__global__ __launch_bounds__(kThreads) void step_kernel_indirect(sgmcmc_layout L, const sgmcmc_step_args* Ap,
                                                                 int64_t chunk_begin) {
  const Early<T, ITEMS> E = step_early<T, VEC, ITEMS>(L, chunk_begin);
  const sgmcmc_step_args A = *Ap;
  needed_here(A);
  const GradParts none = {nullptr, 0, 0, nullptr, nullptr, 1, 1.0};
  step_body<T, KIND, VEC, ITEMS, false, false, false, VEC>(L, A, none, &E);
}
// ... with the closed-form prior gradient added in flight (SGMCMC_INLINE_PRIOR; float, lean priors)
template <int KIND, bool VEC, int ITEMS>
__global__ __launch_bounds__(kThreads) void step_kernel_indirect_prior(sgmcmc_layout L, const sgmcmc_step_args* Ap,
                                                                       int64_t chunk_begin) {
  const Early<float, ITEMS> E = step_early<float, VEC, ITEMS>(L, chunk_begin);
  const sgmcmc_step_args A = *Ap;
  needed_here(A);
  const GradParts none = {nullptr, 0, 0, nullptr, nullptr, 1, 1.0};
  step_body<float, KIND, VEC, ITEMS, false, false, true, VEC>(L, A, none, &E);
}
// ... and the gradient assembled in flight from per-slice partials + closed-form prior
template <typename T, int KIND, bool VEC, int ITEMS>
__global__ __launch_bounds__(kThreads) void step_kernel_parts(sgmcmc_layout L, const sgmcmc_step_args* Ap,
                                                              GradParts G, int64_t chunk_begin) {
  const Early<T, ITEMS> E = step_early<T, VEC, ITEMS>(L, chunk_begin);
  const sgmcmc_step_args A = *Ap;
  needed_here(A);
  step_body<T, KIND, VEC, ITEMS, true, false, false, VEC>(L, A, G, &E);
}

template <typename T, int KIND, bool VEC, int ITEMS>
__global__ __launch_bounds__(kThreads) void step_kernel_parts_val(sgmcmc_layout L, sgmcmc_step_args A,
                                                                  GradParts G) {
  const Early<T, ITEMS> E = step_early<T, VEC, ITEMS>(L, A.chunk_begin);
  step_body<T, KIND, VEC, ITEMS, true, false, false, VEC>(L, A, G, &E);
}

// ------------------------------------------------------------------ per-segment finalize
// Sums a segment's chunk partials in a fixed order (thread t takes chunks t, t+256, ...;
// then a fixed LDS tree) and applies the reference's scalar bookkeeping.
template <int NS>
__device__ __forceinline__ void segment_reduce(const double* __restrict__ partials, int64_t first,
                                               int64_t n, int stride, double (&out)[NS]) {
  __shared__ double sh[NS][kThreads];
  double a[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) a[k] = 0.0;
  for (int64_t c = threadIdx.x; c < n; c += kThreads) {
#pragma unroll
    for (int k = 0; k < NS; ++k) a[k] += partials[(first + c) * stride + k];
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) sh[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int off = kThreads / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
#pragma unroll
      for (int k = 0; k < NS; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) out[k] = sh[k][0];
}

__device__ __forceinline__ int64_t seg_chunks(const sgmcmc_layout& L, const sgmcmc_segment& s) {
  return (s.numel + L.chunk_elems - 1) / L.chunk_elems;
}

// Scalar bookkeeping of one segment after its six sums S are known (thread-serial).
// Returns the segment's contribution to delta_energy's loop: state.delta_energy + point energy
// evaluated with THIS transition's gradient / momentum (verlet_sgld.py:32-47, hmc.py:32-33).
// `pre` (optional): {delta_energy, prev_delta, point_energy} of the segment loaded EARLIER by the caller, so
// that those loads overlap the reduction of the chunk partials instead of following it.
__device__ __forceinline__ double segment_bookkeeping(const sgmcmc_layout& L, const sgmcmc_step_args& A,
                                                      int seg, const sgmcmc_segment& s,
                                                      const double (&S)[SGMCMC_NSUMS],
                                                      const double* pre = nullptr) {
  sgmcmc_seg_state* st = &L.state[seg];
  if (s.g == nullptr)   // skipped tensor: state untouched
    return pre ? pre[0] + pre[2] : st->delta_energy + st->point_energy;
  if (pre) { st->delta_energy = pre[0]; st->prev_delta = pre[1]; }   // (same values: keeps the code below one path)
#pragma unroll
  for (int k = 0; k < SGMCMC_NSUMS; ++k) st->sums[k] = S[k];
  const double d = (double)s.numel, M = s.M;
  const bool initial = A.flags & SGMCMC_INITIAL, final_ = A.flags & SGMCMC_FINAL;
  double point = 0.0;
  if (A.kind == SGMCMC_VERLET) {
    const double c_gm = -.5 * A.bhn * M;  // verlet_sgld.py:170
    const double curv = M * M * (A.num_data * A.num_data) * A.b2h2 / 8;  // :44-47
    if (initial) {
      st->delta_energy = -(curv * S[0]);   // :172
    } else {
      st->delta_energy += st->prev_delta;  // :174
      st->delta_energy += c_gm * S[1];     // :175
    }
    st->prev_delta = c_gm * S[2];  // :176
    if (A.flags & SGMCMC_CALC_METRICS) {
      st->est_temperature = (final_ ? S[4] : S[3]) / d;  // :181-187
      st->est_config_temp = S[5] * (A.num_data / d);     // :189
    }
    point = curv * S[0];
  } else if (A.kind == SGMCMC_HMC) {
    if (initial) st->delta_energy = -.5 * S[3];  // hmc.py:49-51
    if (A.flags & SGMCMC_CALC_METRICS) {
      st->est_temperature = (final_ ? S[4] : S[3]) / d;  // hmc.py:52-53,59,71
      st->est_config_temp = S[5] * (A.num_data / d);     // hmc.py:61
    }
    point = .5 * S[4];  // kinetic energy of the momentum this transition left behind
  } else {
    if (A.flags & SGMCMC_CALC_METRICS) {
      st->est_temperature = S[3] / d;                 // sgld.py:127-137
      st->est_config_temp = S[5] * (A.num_data / d);  // sgld.py:146
    }
  }
  st->point_energy = point;
  // non-finite gradient detector (raise_on_nan, sgld.py:101-104): sum g^2 is finite iff all g are
  if (!(S[0] - S[0] == 0.0)) L.scalars[1] = 1.0;
  return st->delta_energy + point;
}

__device__ __forceinline__ void finalize_step_body(const sgmcmc_layout& L, const sgmcmc_step_args& A) {
  const int seg = A.seg_begin + blockIdx.x;
  const sgmcmc_segment s = L.segs[seg];
  double S[SGMCMC_NSUMS];
  segment_reduce<SGMCMC_NSUMS>(L.partials, s.first_chunk, seg_chunks(L, s), SGMCMC_PSTRIDE, S);
  if (threadIdx.x != 0) return;
  segment_bookkeeping(L, A, seg, s, S);
}

// Small models (few chunks): ONE workgroup finalizes every segment -- thread t owns segments
// t, t+256, ... and sums their chunk partials serially in chunk order -- and also leaves
// scalars[3] = sum_s (delta_energy_s + point_energy_s) in segment order, i.e. the loop of
// VerletSGLD.delta_energy for the gradient this transition used, so a metric step needs no
// further reduction launches.
__device__ __forceinline__ void finalize_small_body(const sgmcmc_layout& L, const sgmcmc_step_args& A) {
  // A group of lanes owns segments seg_begin + g, + n_groups, ...: its lanes stride over the segment's chunk
  // partials (all loads independent), then a fixed shuffle tree; the group's first lane does the
  // bookkeeping.  Every segment is a chain of dependent loads, so the number of ROUNDS is what costs:
  // whole waves while there are enough of them, 16-lane groups (4 segments per wave) for many-tensor nets.
  const int n_segs = A.seg_end - A.seg_begin;
  const int gsz = n_segs * 64 <= (int)blockDim.x ? 64 : 16;
  const int lane = threadIdx.x & (gsz - 1), grp = threadIdx.x / gsz, n_grp = (int)blockDim.x / gsz;
  const bool with_lp = (A.flags & SGMCMC_WITH_LOG_PRIOR) && (A.flags & SGMCMC_CALC_METRICS);
  constexpr int kSmallSegs = 256;
  __shared__ double seg_e[kSmallSegs], seg_lp[kSmallSegs];  // each segment's energy term / log-prior
  for (int seg = A.seg_begin + grp; seg < A.seg_end; seg += n_grp) {
    const sgmcmc_segment s = L.segs[seg];
    const int64_t n = seg_chunks(L, s);
    double S[SGMCMC_NSUMS + 1] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t c = lane; c < n; c += gsz) {
      const double* __restrict__ q = L.partials + (s.first_chunk + c) * SGMCMC_PSTRIDE;
#pragma unroll
      for (int k = 0; k < SGMCMC_NSUMS + 1; ++k) S[k] += q[k];
    }
#pragma unroll
    for (int k = 0; k < SGMCMC_NSUMS + 1; ++k) {
      double x = S[k];
      for (int off = gsz >> 1; off > 0; off >>= 1) x += __shfl_down(x, off, gsz);
      S[k] = x;
    }
    if (lane == 0) {
      double S6[SGMCMC_NSUMS] = {S[0], S[1], S[2], S[3], S[4], S[5]};
      const double e = segment_bookkeeping(L, A, seg, s, S6);
      double lp = 0.0;
      if (with_lp)
        L.state[seg].aux = lp = s.prior_kind == SGMCMC_PRIOR_NONE ? 0.0 : S[6] + (double)s.numel * prior_log_norm(s, s.prior_scale);   // (fused dense path: no linked scales)
      if (seg - A.seg_begin < kSmallSegs) { seg_e[seg - A.seg_begin] = e; seg_lp[seg - A.seg_begin] = lp; }
    }
  }
  __syncthreads();  // the per-segment results written above are visible to thread 0 below
  if (threadIdx.x == 0) {
    double total = 0.0, lp_total = 0.0;  // segment order, as the reference's Python loop
    if (A.seg_end - A.seg_begin <= kSmallSegs) {  // from LDS: 2 x n_seg dependent global loads cost ~0.4 us each
      for (int i = 0; i < A.seg_end - A.seg_begin; ++i) {
        total += seg_e[i];
        if (with_lp) lp_total += seg_lp[i];
      }
    } else {
      for (int seg = A.seg_begin; seg < A.seg_end; ++seg) {
        total += L.state[seg].delta_energy + L.state[seg].point_energy;
        if (with_lp) lp_total += L.state[seg].aux;
      }
    }
    L.scalars[3] = total;
    if (with_lp) L.scalars[2] = lp_total;
  }
}

// The same bookkeeping over SEVERAL workgroups, for nets with many tensors (googleresnet: 65): a wave per
// segment, all segments in flight at once (one round of dependent loads instead of 65 / 64 groups x 2 rounds in
// a single workgroup), and the segment-ordered energy total by the LAST workgroup to arrive.  Hand-off without
// fences: every wave publishes its segments' energy terms with agent-scope (write-through) 8-byte stores into
// the spare slot partials[first_chunk][7], drains them (s_waitcnt vmcnt(0)), the workgroup takes a ticket with
// a device-scope atomic; the workgroup that draws the last ticket reads the terms back with agent-scope loads
// (MI355X_MICROARCH.md "Valid forms": 8-byte agent atomics on both sides) and resets the ticket.
// (bid, n_blocks): this workgroup's index among the workgroups doing the bookkeeping -- the whole grid for the
// kernels below, the tail of a staging launch for a deferred one (augment_hip.inc)
__device__ __forceinline__ void finalize_multi_body(const sgmcmc_layout& L, const sgmcmc_step_args& A, int bid,
                                                    int n_blocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = (int)blockDim.x >> 6;
  const int gw = bid * wpb + wave, n_w = n_blocks * wpb;
  for (int seg = A.seg_begin + gw; seg < A.seg_end; seg += n_w) {
    const sgmcmc_segment s = L.segs[seg];
    double pre[3] = {0.0, 0.0, 0.0};
    if (lane == 0) {
      const sgmcmc_seg_state* st = &L.state[seg];
      pre[0] = st->delta_energy; pre[1] = st->prev_delta; pre[2] = st->point_energy;
    }
    const int64_t n = seg_chunks(L, s);
    double S[SGMCMC_NSUMS] = {0, 0, 0, 0, 0, 0};
    for (int64_t c = lane; c < n; c += 64) {
      const double* __restrict__ q = L.partials + (s.first_chunk + c) * SGMCMC_PSTRIDE;
#pragma unroll
      for (int k = 0; k < SGMCMC_NSUMS; ++k) S[k] += q[k];
    }
#pragma unroll
    for (int k = 0; k < SGMCMC_NSUMS; ++k) {
      double x = S[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      S[k] = x;
    }
    if (lane == 0) {
      const double e = segment_bookkeeping(L, A, seg, s, S, pre);
      __hip_atomic_store(L.partials + s.first_chunk * SGMCMC_PSTRIDE + 7, e, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's published terms have left the CU
  __shared__ int is_last;
  __shared__ double terms[kThreads];
  __syncthreads();
  unsigned long long* ticket = reinterpret_cast<unsigned long long*>(L.scalars + 7);
  if (threadIdx.x == 0) {
    const unsigned long long t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    is_last = (t + 1 == (unsigned long long)n_blocks);
  }
  __syncthreads();
  if (!is_last) return;
  const int n_segs = A.seg_end - A.seg_begin;
  double total = 0.0;   // segment order, as the reference's Python loop (verlet_sgld.py:32-38)
  for (int base = 0; base < n_segs; base += (int)blockDim.x) {
    const int i = base + (int)threadIdx.x;
    if (i < n_segs)
      terms[threadIdx.x] = __hip_atomic_load(L.partials + L.segs[A.seg_begin + i].first_chunk * SGMCMC_PSTRIDE + 7,
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = n_segs - base < (int)blockDim.x ? n_segs - base : (int)blockDim.x;
      for (int j = 0; j < m; ++j) total += terms[j];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    L.scalars[3] = total;
    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(kThreads) void finalize_step_kernel(sgmcmc_layout L, sgmcmc_step_args A) {
  finalize_step_body(L, A);
}
__global__ __launch_bounds__(kThreads) void finalize_multi_kernel(sgmcmc_layout L, sgmcmc_step_args A) {
  finalize_multi_body(L, A, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(kThreads) void finalize_multi_kernel_indirect(sgmcmc_layout L,
                                                                           const sgmcmc_step_args* Ap) {
  const sgmcmc_step_args A = *Ap;
  finalize_multi_body(L, A, (int)blockIdx.x, (int)gridDim.x);
}
__global__ __launch_bounds__(kThreads) void finalize_step_kernel_indirect(sgmcmc_layout L,
                                                                          const sgmcmc_step_args* Ap) {
  const sgmcmc_step_args A = *Ap;
  finalize_step_body(L, A);
}

__global__ void finalize_small_kernel(sgmcmc_layout L, sgmcmc_step_args A) {
  finalize_small_body(L, A);
}
__global__ void finalize_small_kernel_indirect(sgmcmc_layout L,
                                                                           const sgmcmc_step_args* Ap) {
  const sgmcmc_step_args A = *Ap;
  finalize_small_body(L, A);
}

// ------------------------------------------------------------------ auxiliary kernels
template <typename T>
__global__ __launch_bounds__(kThreads) void sample_momentum_kernel(sgmcmc_layout L, double std_,
                                                                   double keep, uint64_t seed,
                                                                   uint32_t stream, uint64_t draw) {
  const int64_t chunk = blockIdx.x;
  const ChunkCtx cx = chunk_ctx(L, chunk);
  const sgmcmc_segment* __restrict__ sp = &L.segs[cx.seg];
  T* __restrict__ mp = (T*)L.m + cx.arena_off;
  const uint64_t noise0 = (uint64_t)(sp->noise_base + cx.seg_off);
  const T sd = (T)std_, sk = (T)sqrt(keep);
  for (int it = 0; it < items_of(L); ++it) {
    const int j = (it * kThreads + threadIdx.x) * 4;
    const int n = cx.n_valid - j < 4 ? cx.n_valid - j : 4;
    if (n <= 0) break;
    float z[4];
    spec_normal4(seed, stream, draw, 1u, (noise0 + (uint64_t)j) >> 2, z);
    Item<T> mo = (keep == 0.0) ? Item<T>{{T(0), T(0), T(0), T(0)}} : load_guarded<T>(mp + j, n);
    Item<T> mn;
#pragma unroll
    for (int l = 0; l < 4; ++l)  // sgld.py:66-69
      mn.x[l] = (keep == 0.0) ? (T)z[l] * sd : fma_t<T>((T)z[l], sd, mo.x[l] * sk);
    if (n == 4) store_item<T>(mp + j, mn); else store_guarded<T>(mp + j, mn, n);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void restore_kernel(sgmcmc_layout L, int restore_m) {
  const int64_t chunk = blockIdx.x;
  const ChunkCtx cx = chunk_ctx(L, chunk);
  const sgmcmc_segment* __restrict__ sp = &L.segs[cx.seg];
  T* __restrict__ gp = (T*)sp->g + cx.seg_off;
  T* __restrict__ thp = (T*)sp->theta + cx.seg_off;
  T* __restrict__ mp = (T*)L.m + cx.arena_off;
  const T* __restrict__ pth = (const T*)L.prev_theta + cx.arena_off;
  const T* __restrict__ pg = (const T*)L.prev_g + cx.arena_off;
  const T* __restrict__ pm = (const T*)L.prev_m + cx.arena_off;
  for (int it = 0; it < items_of(L); ++it) {
    const int j = (it * kThreads + threadIdx.x) * 4;
    const int n = cx.n_valid - j < 4 ? cx.n_valid - j : 4;
    if (n <= 0) break;
    store_guarded<T>(thp + j, load_guarded<T>(pth + j, n), n);  // verlet_sgld.py:64
    if (sp->g) store_guarded<T>(gp + j, load_guarded<T>(pg + j, n), n);    // :65
    if (restore_m) store_guarded<T>(mp + j, load_guarded<T>(pm + j, n), n);  // :66-69
  }
}

// which: 0 = sum v, 1 = m.m, 2 = g.g (clamped like the step kernel sees it)
template <typename T>
__global__ __launch_bounds__(kThreads) void dot_kernel(sgmcmc_layout L, int which, double clampv) {
  const int64_t chunk = blockIdx.x;
  const ChunkCtx cx = chunk_ctx(L, chunk);
  const sgmcmc_segment* __restrict__ sp = &L.segs[cx.seg];
  const T* __restrict__ src = which == 2 ? (const T*)sp->g + cx.seg_off
                                         : (const T*)(which == 1 ? L.m : L.v) + cx.arena_off;
  double acc[1] = {0.0};
  const bool absent = which == 2 && sp->g == nullptr;   // no gradient: contributes nothing
  for (int it = 0; it < items_of(L) && !absent; ++it) {
    const int j = (it * kThreads + threadIdx.x) * 4;
    const int n = cx.n_valid - j < 4 ? cx.n_valid - j : 4;
    if (n <= 0) break;
    const Item<T> x = load_guarded<T>(src + j, n);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      if (l < n) {
        T e = x.x[l];
        if (which == 2 && clampv > 0) e = clamp_grad<T>(e, (T)clampv);
        acc[0] = which == 0 ? acc[0] + (double)e : fma((double)e, (double)e, acc[0]);
      }
    }
  }
  block_reduce_store<1>(acc, L.partials + chunk * SGMCMC_PSTRIDE);
}

// mode 0: state.aux = segment sum ; mode 1 (Verlet) / 2 (HMC): state.point_energy
__global__ __launch_bounds__(kThreads) void finalize_dot_kernel(sgmcmc_layout L, int mode,
                                                                double num_data, double b2h2) {
  const int seg = blockIdx.x;
  const sgmcmc_segment s = L.segs[seg];
  double S[1];
  segment_reduce<1>(L.partials, s.first_chunk, seg_chunks(L, s), SGMCMC_PSTRIDE, S);
  if (threadIdx.x != 0) return;
  sgmcmc_seg_state* st = &L.state[seg];
  if (mode == 0) st->aux = S[0];
  else if (mode == 1) st->point_energy = (s.M * s.M * (num_data * num_data) * b2h2 / 8) * S[0];
  else st->point_energy = .5 * S[0];
}

// ------------------------------------------------------------------ fused priors
// d/dtheta of -log p(theta)/N and (optionally) log p(theta), element-wise families with scalar
// loc/scale/df (SURVEY.md Appendix A "Priors").  The gradient is applied in the working precision
// with one rounding for the factor and one fma into g; log p is evaluated and accumulated in fp64.
__device__ __forceinline__ double prior_log_norm(const sgmcmc_segment& s, double scale) {
  // per-element normalising constant of the density; the hyper-prior kinds
  // accumulate their full log-density element by element and need none
  if (s.prior_kind == SGMCMC_PRIOR_NORMAL) return -log(scale) - 0.9189385332046727418;
  if (s.prior_kind == SGMCMC_PRIOR_LAPLACE) return -log(2.0 * scale);
  if (s.prior_kind == SGMCMC_PRIOR_STUDENT_T)
    return -log(scale) - 0.5 * log(s.prior_df) - 0.5723649429247000870 -
           lgamma(0.5 * s.prior_df) + lgamma(0.5 * (s.prior_df + 1.0));
  if (s.prior_kind == SGMCMC_PRIOR_CAUCHY) return -log(scale) - 1.1447298858494001741;  // -ln(pi sigma)
  if (s.prior_kind == SGMCMC_PRIOR_GENNORM) return -log(2.0 * scale) - lgamma(1.0 / s.prior_df) + log(s.prior_df);
  return 0.0;
}

// FULL = false: only the constant-scale families Normal / Laplace / Student-t / Cauchy (BASELINE's configs): the
// lean variant the host selects unless some segment has a generalised-normal, hierarchical or hyper prior
template <typename T, bool FULL>
__device__ __forceinline__ void prior_body(const sgmcmc_layout& L, double num_data, bool calc_logp,
                                           const GradParts& G) {
  const int64_t chunk = blockIdx.x;
  const ChunkCtx cx = chunk_ctx(L, chunk);
  const sgmcmc_segment* __restrict__ sp = &L.segs[cx.seg];
  const bool parts = G.gpart != nullptr;
  const bool linked = FULL && sp->scale_link > 0;
  double acc[2] = {0.0, 0.0};   // log-density partial; d/dscale of it (hierarchical scales)
  if ((sp->prior_kind != SGMCMC_PRIOR_NONE || parts) && sp->g != nullptr) {
    T* __restrict__ gp = (T*)sp->g + cx.seg_off;
    const T* __restrict__ thp = (const T*)sp->theta + cx.seg_off;
    const float* __restrict__ pp = parts ? G.gpart + sp->noise_base + cx.seg_off : nullptr;
    PriorCoef<T> PC;
    PC.template init<FULL>(L, sp, num_data);
    for (int it = 0; it < items_of(L); ++it) {
      const int j = (it * kThreads + threadIdx.x) * 4;
      const int n = cx.n_valid - j < 4 ? cx.n_valid - j : 4;
      if (n <= 0) break;
      Item<T> g = parts ? sum_parts<T>(pp + j, G.n_slices, G.stride, n) : load_guarded<T>(gp + j, n);
      const Item<T> th = load_guarded<T>(thp + j, n);
#pragma unroll
      for (int l = 0; l < 4; ++l)
        if (l < n) g.x[l] = PC.template apply<FULL>(g.x[l], th.x[l], calc_logp, acc[0], acc[1]);
      store_guarded<T>(gp + j, g, n);
    }
  }
  if (calc_logp || linked) block_reduce_store<2>(acc, L.partials + chunk * SGMCMC_PSTRIDE + 6);
  if (parts) publish_batch_stats(L, G);
}

// Hierarchical scales: the hyper segment h (one element, raw parameter s) of every linked weight segment w gets
//   g_h += -(1/N) * (sum_j d/dscale log p(theta_wj)) * dx/ds        (the chain rule through x = value(s))
// from the per-chunk partials the prior kernel left in partials[.][7].  One workgroup; thread t owns the
// linked segments t, t + 256, ... and sums their chunks in chunk order (deterministic).
template <typename T>
__global__ __launch_bounds__(kThreads) void hyper_link_kernel(sgmcmc_layout L, double num_data) {
  for (int w = threadIdx.x; w < L.n_seg; w += kThreads) {
    const sgmcmc_segment s = L.segs[w];
    if (s.scale_link <= 0) continue;
    const sgmcmc_segment h = L.segs[s.scale_link - 1];
    if (h.g == nullptr) continue;
    double D = 0.0;
    const int64_t n = seg_chunks(L, s);
    for (int64_t c = 0; c < n; ++c) D += L.partials[(s.first_chunk + c) * SGMCMC_PSTRIDE + 7];
    double dxds;
    hyper_value(h, (double)*(const T*)h.theta, dxds);
    T* gh = (T*)h.g;
    *gh = (T)((double)*gh - D * dxds / num_data);
  }
}

template <typename T, bool FULL>
__global__ __launch_bounds__(kThreads) void prior_kernel(sgmcmc_layout L, double num_data,
                                                         int calc_logp, GradParts G) {
  prior_body<T, FULL>(L, num_data, calc_logp != 0, G);
}
template <typename T>
__global__ __launch_bounds__(kThreads) void prior_kernel_indirect(sgmcmc_layout L, double num_data,
                                                                  const sgmcmc_step_args* Ap,
                                                                  GradParts G) {
  prior_body<T, false>(L, num_data, (Ap->flags & SGMCMC_CALC_METRICS) != 0, G);
}

__global__ __launch_bounds__(kThreads) void finalize_prior_kernel(sgmcmc_layout L) {
  const int seg = blockIdx.x;
  const sgmcmc_segment s = L.segs[seg];
  double S[1];
  segment_reduce<1>(L.partials + 6, s.first_chunk, seg_chunks(L, s), SGMCMC_PSTRIDE, S);
  if (threadIdx.x != 0) return;
  L.state[seg].aux = s.prior_kind == SGMCMC_PRIOR_NONE ? 0.0 : S[0] + (double)s.numel * prior_log_norm(s, current_scale(L, s));
}

__global__ void total_prior_kernel(sgmcmc_layout L) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0;
  for (int s = 0; s < L.n_seg; ++s) total += L.state[s].aux;
  L.scalars[2] = total;
}

__global__ void total_energy_kernel(sgmcmc_layout L) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0;  // verlet_sgld.py:32-38, same association as the reference's Python loop
  for (int s = 0; s < L.n_seg; ++s) total += L.state[s].delta_energy + L.state[s].point_energy;
  L.scalars[0] = total;
}

__global__ void debug_normals_kernel(float* out, int64_t start, int64_t n, uint64_t seed,
                                     uint32_t stream, uint64_t draw, uint32_t purpose) {
  const int64_t q0 = start >> 2;
  const int64_t nq = ((start + n + 3) >> 2) - q0;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nq;
       q += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    spec_normal4(seed, stream, draw, purpose, (uint64_t)(q0 + q), z);
    for (int l = 0; l < 4; ++l) {
      const int64_t i = ((q0 + q) << 2) + l - start;
      if (i >= 0 && i < n) out[i] = z[l];
    }
  }
}

// mode 0: scalars by value; 1: scalars from device memory; 2: that + gradient assembled in flight
template <typename T, int KIND, bool VEC, int ITEMS>
void launch_step_mode(const sgmcmc_layout& L, const sgmcmc_step_args& A, const sgmcmc_step_args* Ad,
                      const GradParts* G, hipStream_t s) {
  const dim3 grid((unsigned)(A.chunk_end - A.chunk_begin)), block(kThreads);
  if (G && Ad) SGMCMC_LAUNCH((step_kernel_parts<T, KIND, VEC, ITEMS>), grid, block, 0, s, L, Ad, *G, A.chunk_begin);
  else if (G) SGMCMC_LAUNCH((step_kernel_parts_val<T, KIND, VEC, ITEMS>), grid, block, 0, s, L, A, *G);
  else if (Ad && (A.flags & SGMCMC_INLINE_PRIOR)) SGMCMC_LAUNCH((step_kernel_indirect_prior<KIND, VEC, ITEMS>), grid, block, 0, s, L, Ad, A.chunk_begin);
  else if (Ad) SGMCMC_LAUNCH((step_kernel_indirect<T, KIND, VEC, ITEMS>), grid, block, 0, s, L, Ad, A.chunk_begin);
  else if (VEC && ITEMS == 4 &&
           (double)L.n_chunks * (double)L.chunk_elems * sizeof(T) * 7.0 > 224.0 * 1024 * 1024)
    SGMCMC_LAUNCH((step_kernel_stream<T, KIND>), grid, block, 0, s, L, A);
  else SGMCMC_LAUNCH((step_kernel<T, KIND, VEC, ITEMS>), grid, block, 0, s, L, A);
}

template <typename T, bool VEC, int ITEMS>
void launch_step_kind(const sgmcmc_layout& L, const sgmcmc_step_args& A, const sgmcmc_step_args* Ad,
                      const GradParts* G, hipStream_t s) {
  switch (A.kind) {
    case SGMCMC_VERLET: launch_step_mode<T, SGMCMC_VERLET, VEC, ITEMS>(L, A, Ad, G, s); break;
    case SGMCMC_HMC: launch_step_mode<T, SGMCMC_HMC, VEC, ITEMS>(L, A, Ad, G, s); break;
    default: launch_step_mode<T, SGMCMC_SGLD, VEC, ITEMS>(L, A, Ad, G, s); break;
  }
}

// validates and launches the fused update kernel in the requested mode
int launch_step(const sgmcmc_layout* L, const sgmcmc_step_args* A, const sgmcmc_step_args* Ad,
                const GradParts* G, hipStream_t s) {
  if (!L || !A || A->chunk_end <= A->chunk_begin || A->seg_end <= A->seg_begin)
    return (int)hipErrorInvalidValue;
  if (A->kind < SGMCMC_VERLET || A->kind > SGMCMC_SGLD) return (int)hipErrorInvalidValue;
  if (L->chunk_elems != SGMCMC_CHUNK && L->chunk_elems != SGMCMC_CHUNK_SMALL) return (int)hipErrorInvalidValue;
  // the graph-replay kernels request their chunk's m and v before they know the transition's flags: both arenas have to
  // exist (padded to whole chunks) on every path, SGLD without momentum and final steps included
  if (Ad && (!L->m || !L->v)) return (int)hipErrorInvalidValue;
  // the in-flight prior exists for the graph-replay kernel on float32 arenas only
  if ((A->flags & SGMCMC_INLINE_PRIOR) && (!Ad || G || L->dtype != SGMCMC_F32)) return (int)hipErrorInvalidValue;
  // ... and, like the gradient-assembling kernels, carries the lean prior code only
  if (((A->flags & SGMCMC_INLINE_PRIOR) || G) && L->prior_flags != 0) return (int)hipErrorInvalidValue;
  const bool vec = !(A->flags & SGMCMC_UNALIGNED);
  const bool small = L->chunk_elems == SGMCMC_CHUNK_SMALL;
  if (L->dtype == SGMCMC_F32) {
    if (small) { if (vec) launch_step_kind<float, true, 1>(*L, *A, Ad, G, s); else launch_step_kind<float, false, 1>(*L, *A, Ad, G, s); }
    else { if (vec) launch_step_kind<float, true, 4>(*L, *A, Ad, G, s); else launch_step_kind<float, false, 4>(*L, *A, Ad, G, s); }
  } else if (L->dtype == SGMCMC_F64 && !G) {
    if (small) { if (vec) launch_step_kind<double, true, 1>(*L, *A, Ad, G, s); else launch_step_kind<double, false, 1>(*L, *A, Ad, G, s); }
    else { if (vec) launch_step_kind<double, true, 4>(*L, *A, Ad, G, s); else launch_step_kind<double, false, 4>(*L, *A, Ad, G, s); }
  } else {
    return (int)hipErrorInvalidValue;
  }
  return 0;
}

// how the bookkeeping of a SMALL_FINALIZE transition is spread: several workgroups (a wave per segment) for many
// tensors and no log-prior to finish, else one
inline int finalize_blocks(const sgmcmc_step_args& A) {
  const int n_segs = A.seg_end - A.seg_begin;
  return (n_segs > 8 && !(A.flags & SGMCMC_WITH_LOG_PRIOR)) ? (n_segs + 3) / 4 : 1;
}

void launch_finalize(const sgmcmc_layout* L, const sgmcmc_step_args* A, const sgmcmc_step_args* Ad,
                     hipStream_t s) {
  if (A->flags & SGMCMC_DEFER_FINALIZE) return;  // the caller runs it later (sgmcmc_finalize)
  const bool small = A->flags & SGMCMC_SMALL_FINALIZE;
  const int n_segs = A->seg_end - A->seg_begin;
  // many tensors and no log-prior to finish: a wave per segment over several workgroups, last arriver totals
  // (A_host carries the same SMALL_FINALIZE / WITH_LOG_PRIOR bits as the device copy: they select kernels)
  if (small && n_segs > 8 && !(A->flags & SGMCMC_WITH_LOG_PRIOR)) {
    const dim3 grid((unsigned)((n_segs + 3) / 4)), block(kThreads);
    if (Ad) SGMCMC_LAUNCH(finalize_multi_kernel_indirect, grid, block, 0, s, *L, Ad);
    else SGMCMC_LAUNCH(finalize_multi_kernel, grid, block, 0, s, *L, *A);
    return;
  }
  // small: one workgroup, one wave per segment in flight -- 16 waves once there are more than 4 segments
  // (every segment costs a chain of dependent loads, so the round count is what matters)
  const dim3 grid(small ? 1u : (unsigned)n_segs), block(small && n_segs > 4 ? 1024u : (unsigned)kThreads);
  if (Ad) {
    if (small) SGMCMC_LAUNCH(finalize_small_kernel_indirect, grid, block, 0, s, *L, Ad);
    else SGMCMC_LAUNCH(finalize_step_kernel_indirect, grid, block, 0, s, *L, Ad);
  } else {
    if (small) SGMCMC_LAUNCH(finalize_small_kernel, grid, block, 0, s, *L, *A);
    else SGMCMC_LAUNCH(finalize_step_kernel, grid, block, 0, s, *L, *A);
  }
}

}  // namespace

extern "C" {

int sgmcmc_abi_version(void) { return SGMCMC_ABI_VERSION; }
// The hash of the sources this binary was compiled from (bnn_priors_amd._hip.source_sha(), handed in by the build as
// -DSGMCMC_SOURCE_SHA="..."; a variant build appends its extra flags): the loader refuses a library whose hash is not the
// tree's, and measurements are stamped with THIS string, not with a hash of whatever files lie next to the binary.
#ifndef SGMCMC_SOURCE_SHA
#define SGMCMC_SOURCE_SHA "unstamped"
#endif
const char* sgmcmc_source_sha(void) { return SGMCMC_SOURCE_SHA; }

const char* sgmcmc_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int sgmcmc_step_timed(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream,
                      void* ev_start, void* ev_stop) {
  SGMCMC_FRESH_ERROR_STATE();
  hipStream_t s = (hipStream_t)stream;
  if (ev_start && ev_stop) { sgmcmc_timing::e0 = (hipEvent_t)ev_start; sgmcmc_timing::e1 = (hipEvent_t)ev_stop; }
  const int rc = launch_step(L, A, nullptr, nullptr, s);
  sgmcmc_timing::e0 = sgmcmc_timing::e1 = nullptr;
  if (rc) return rc;
  launch_finalize(L, A, nullptr, s);
  return (int)hipGetLastError();
}

int sgmcmc_step(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  return sgmcmc_step_timed(L, A, stream, nullptr, nullptr);
}

int sgmcmc_step_indirect(const sgmcmc_layout* L, const sgmcmc_step_args* A, const sgmcmc_step_args* Ad,
                         void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!Ad) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const int rc = launch_step(L, A, Ad, nullptr, s);
  if (rc) return rc;
  launch_finalize(L, A, Ad, s);
  return (int)hipGetLastError();
}

int sgmcmc_step_indirect_parts(const sgmcmc_layout* L, const sgmcmc_step_args* A,
                               const sgmcmc_step_args* Ad, const sgmcmc_grad_parts* P, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!Ad || !P || !P->gpart || P->n_slices <= 0 || P->batch <= 0 || !(P->num_data > 0))
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const GradParts G = {P->gpart, P->n_slices, P->stride, P->loss_part, P->correct_part, P->batch,
                       P->num_data};
  const int rc = launch_step(L, A, Ad, &G, s);
  if (rc) return rc;
  launch_finalize(L, A, Ad, s);
  return (int)hipGetLastError();
}

// by-value variant used by sgmcmc_dense_step_direct (csrc/mlp_hip.inc); internal: not part of the C ABI
__attribute__((visibility("hidden"))) int sgmcmc_step_parts_value(const sgmcmc_layout* L, const sgmcmc_step_args* A,
                            const sgmcmc_grad_parts* P, void* stream) {
  if (!P || !P->gpart || P->n_slices <= 0 || P->batch <= 0 || !(P->num_data > 0))
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const GradParts G = {P->gpart, P->n_slices, P->stride, P->loss_part, P->correct_part, P->batch,
                       P->num_data};
  const int rc = launch_step(L, A, nullptr, &G, s);
  if (rc) return rc;
  launch_finalize(L, A, nullptr, s);
  return (int)hipGetLastError();
}

int sgmcmc_finalize(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!L || !A || A->seg_end <= A->seg_begin) return (int)hipErrorInvalidValue;
  sgmcmc_step_args B = *A;
  B.flags &= ~(uint32_t)SGMCMC_DEFER_FINALIZE;
  launch_finalize(L, &B, nullptr, (hipStream_t)stream);
  return (int)hipGetLastError();
}

int sgmcmc_event_create(void** ev) {
  SGMCMC_FRESH_ERROR_STATE();
  hipEvent_t e;
  const hipError_t err = hipEventCreate(&e);
  if (err == hipSuccess) *ev = (void*)e;
  return (int)err;
}

int sgmcmc_event_destroy(void* ev) { return (int)hipEventDestroy((hipEvent_t)ev); }

int sgmcmc_time_next_launch(void* ev_start, void* ev_stop) {
  if (!ev_start || !ev_stop) return (int)hipErrorInvalidValue;
  sgmcmc_timing::e0 = (hipEvent_t)ev_start;
  sgmcmc_timing::e1 = (hipEvent_t)ev_stop;
  return 0;
}

int sgmcmc_event_record(void* ev, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  return (int)hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
}

int sgmcmc_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {
  SGMCMC_FRESH_ERROR_STATE();
  hipError_t err = hipEventSynchronize((hipEvent_t)ev_stop);
  if (err != hipSuccess) return (int)err;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

int sgmcmc_sample_momentum(const sgmcmc_layout* L, double std, double keep, uint64_t seed,
                           uint32_t stream, uint64_t draw, void* stream_) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!L || L->n_chunks <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream_;
  const dim3 grid((unsigned)L->n_chunks), block(kThreads);
  if (L->dtype == SGMCMC_F32)
    SGMCMC_LAUNCH(sample_momentum_kernel<float>, grid, block, 0, s, *L, std, keep, seed, stream, draw);
  else
    SGMCMC_LAUNCH(sample_momentum_kernel<double>, grid, block, 0, s, *L, std, keep, seed, stream, draw);
  return (int)hipGetLastError();
}

int sgmcmc_restore(const sgmcmc_layout* L, int restore_momentum, uint32_t flags, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  (void)flags;
  if (!L || L->n_chunks <= 0) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)L->n_chunks), block(kThreads);
  if (L->dtype == SGMCMC_F32)
    SGMCMC_LAUNCH(restore_kernel<float>, grid, block, 0, s, *L, restore_momentum);
  else
    SGMCMC_LAUNCH(restore_kernel<double>, grid, block, 0, s, *L, restore_momentum);
  return (int)hipGetLastError();
}

static int launch_dot(const sgmcmc_layout* L, int which, double clampv, hipStream_t s) {
  const dim3 grid((unsigned)L->n_chunks), block(kThreads);
  if (L->dtype == SGMCMC_F32) SGMCMC_LAUNCH(dot_kernel<float>, grid, block, 0, s, *L, which, clampv);
  else SGMCMC_LAUNCH(dot_kernel<double>, grid, block, 0, s, *L, which, clampv);
  return (int)hipGetLastError();
}

int sgmcmc_delta_energy(const sgmcmc_layout* L, int kind, double num_data, double b2h2,
                        double grad_clamp, uint32_t flags, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  (void)flags;
  if (!L || L->n_chunks <= 0 || (kind != SGMCMC_VERLET && kind != SGMCMC_HMC))
    return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  int err = launch_dot(L, kind == SGMCMC_HMC ? 1 : 2, grad_clamp, s);
  if (err) return err;
  SGMCMC_LAUNCH(finalize_dot_kernel, dim3((unsigned)L->n_seg), dim3(kThreads), 0, s, *L,
                     kind == SGMCMC_HMC ? 2 : 1, num_data, b2h2);
  SGMCMC_LAUNCH(total_energy_kernel, dim3(1), dim3(64), 0, s, *L);
  return (int)hipGetLastError();
}

int sgmcmc_segment_sum(const sgmcmc_layout* L, int which, uint32_t flags, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  (void)flags;
  if (!L || L->n_chunks <= 0 || which < 0 || which > 2) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  int err = launch_dot(L, which, 0.0, s);
  if (err) return err;
  SGMCMC_LAUNCH(finalize_dot_kernel, dim3((unsigned)L->n_seg), dim3(kThreads), 0, s, *L, 0, 0.0, 0.0);
  return (int)hipGetLastError();
}

int sgmcmc_prior_grad(const sgmcmc_layout* L, double num_data, int calc_log_prob, uint32_t flags,
                      void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!L || L->n_chunks <= 0 || !(num_data > 0)) return (int)hipErrorInvalidValue;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)L->n_chunks), block(kThreads);
  const GradParts none = {nullptr, 0, 0, nullptr, nullptr, 1, 1.0};
  flags |= L->prior_flags;
  const bool full = (flags & (SGMCMC_PRIOR_HAS_LINKS | SGMCMC_PRIOR_FULL)) != 0;
  if (L->dtype == SGMCMC_F32) {
    if (full) SGMCMC_LAUNCH((prior_kernel<float, true>), grid, block, 0, s, *L, num_data, calc_log_prob, none);
    else SGMCMC_LAUNCH((prior_kernel<float, false>), grid, block, 0, s, *L, num_data, calc_log_prob, none);
  } else {
    if (full) SGMCMC_LAUNCH((prior_kernel<double, true>), grid, block, 0, s, *L, num_data, calc_log_prob, none);
    else SGMCMC_LAUNCH((prior_kernel<double, false>), grid, block, 0, s, *L, num_data, calc_log_prob, none);
  }
  if (flags & SGMCMC_PRIOR_HAS_LINKS) {
    if (L->dtype == SGMCMC_F32) SGMCMC_LAUNCH(hyper_link_kernel<float>, dim3(1), block, 0, s, *L, num_data);
    else SGMCMC_LAUNCH(hyper_link_kernel<double>, dim3(1), block, 0, s, *L, num_data);
  }
  if (calc_log_prob) {
    SGMCMC_LAUNCH(finalize_prior_kernel, dim3((unsigned)L->n_seg), dim3(kThreads), 0, s, *L);
    SGMCMC_LAUNCH(total_prior_kernel, dim3(1), dim3(64), 0, s, *L);
  }
  return (int)hipGetLastError();
}

int sgmcmc_grad_reduce_prior(const sgmcmc_layout* L, const float* gpart, int n_slices,
                             int64_t stride, const float* loss_part, const float* correct_part,
                             int batch, double num_data, uint32_t flags,
                             const sgmcmc_step_args* A_dev, void* stream) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!L || L->n_chunks <= 0 || !(num_data > 0) || !gpart || n_slices <= 0 || batch <= 0 ||
      L->dtype != SGMCMC_F32)
    return (int)hipErrorInvalidValue;
  if (L->prior_flags != 0) return (int)hipErrorInvalidValue;  // lean prior code only (prior_body<T, false>)
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)L->n_chunks), block(kThreads);
  const GradParts G = {gpart, n_slices, stride, loss_part, correct_part, batch, num_data};
  if (A_dev)
    SGMCMC_LAUNCH(prior_kernel_indirect<float>, grid, block, 0, s, *L, num_data, A_dev, G);
  else {
    const int calc = (int)((flags & SGMCMC_CALC_METRICS) != 0);
    SGMCMC_LAUNCH((prior_kernel<float, false>), grid, block, 0, s, *L, num_data, calc, G);
    if (calc) {  // no sampler launch follows to finish the log-prior: do it here
      SGMCMC_LAUNCH(finalize_prior_kernel, dim3((unsigned)L->n_seg), dim3(kThreads), 0, s, *L);
      SGMCMC_LAUNCH(total_prior_kernel, dim3(1), dim3(64), 0, s, *L);
    }
  }
  return (int)hipGetLastError();
}

int sgmcmc_debug_normals(float* out, int64_t start, int64_t n, uint64_t seed, uint32_t stream,
                         uint64_t draw, uint32_t purpose, void* stream_) {
  SGMCMC_FRESH_ERROR_STATE();
  if (!out || n <= 0) return (int)hipErrorInvalidValue;
  SGMCMC_LAUNCH(debug_normals_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream_, out,
                     start, n, seed, stream, draw, purpose);
  return (int)hipGetLastError();
}

}  // extern "C"

// the fused dense-net kernels share the finalize body above (deferred finalize rides in their launch)
#include "mlp_hip.inc"
// the convolutional trunk's fp32-MFMA kernels (googleresnet gradient)
#include "conv_hip.inc"
#include "conv2_hip.inc"
#include "conv_down_hip.inc"
#include "conv50_hip.inc"
#include "bn_hip.inc"
// Measured alternatives that LOST to the default kernels (the BatchNorm folded into the next
// convolution's staging, the BatchNorm backward formed inside the convolution-gradient launch, the weight-gradient half
// on a side stream, round 6's uniform backward convolution).  They are kept,
// tested and switchable -- but only in a library built with -DSGMCMC_ALTERNATIVES (include/sgmcmc_hip_alternatives.h;
// SGMCMC_ALTERNATIVES=1 in the environment of bnn_priors_amd._hip.build()): the shipped library and the header a
// maintainer reads describe the path that runs.
#ifdef SGMCMC_ALTERNATIVES
#include "sgmcmc_hip_alternatives.h"
#include "conv_fused_hip.inc"
#include "conv_fold_hip.inc"
#include "conv_uni_hip.inc"
#endif
#include "pool_hip.inc"
#include "augment_hip.inc"
