// mlp_hip.hip -- fused forward + backward of the BASELINE dense classifier
// (ClassificationDenseNet: Linear-ReLU-Linear-ReLU-Linear + categorical likelihood;
//  reference: bnn_priors/models/dense_nets.py:48-67, models/base.py:168-191) for gfx950.
//
// Why: one leapfrog step of this net is ~30 ATen/rocBLAS launches of a few microseconds
// each (three GEMMs forward, five backward, bias/ReLU/softmax/NLL kernels, reductions); at
// batch 128 x 42k parameters the step is pure launch latency.  This kernel does the whole
// stochastic-gradient evaluation of -(1/B) sum_i log p(y_i | x_i) in ONE launch:
//
//   grid  = ceil(B / 16) workgroups, each owns 16 batch rows (gathered through an index
//           array, so the minibatch is never materialised);
//   block = 1024 threads = 16 wavefronts; every contraction runs on the matrix cores with
//           v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, so
//           no precision is given up); the two 784-wide contractions are spread over all 16
//           waves (K quarters x column tiles), the small layers use 4;
//   LDS   = the 16 x IN input slice (50 KB for MNIST), activations and their gradients;
//   out   = per-workgroup PARTIAL parameter gradients gpart[s][D] (packed, 4-aligned per
//           tensor like the sampler's noise index), per-workgroup loss / #correct.
// The partials are summed over s in a fixed order by sgmcmc_grad_reduce_prior (no atomics:
// run-to-run bitwise reproducible), which also adds the prior's gradient.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "sgmcmc_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 1024;   // 16 wavefronts: the first layer's K and tiles are spread over them
constexpr int kWaves = kThreads / 64;
constexpr int KSPLIT = 4;         // K quarters of forward 1 / column-tile stride of backward 1
constexpr int F1_UNROLL = 13;     // >= ceil(49 / KSPLIT): a wave issues all its W1 loads at once for IN = 784
constexpr int STG = 20;           // row stride of the per-wave 16x16 store-staging patch
constexpr int ROWS = SGMCMC_MLP_ROWS;  // batch rows per workgroup (one MFMA M tile)
constexpr int HP = 64;                 // hidden widths are padded to 4 MFMA column tiles
constexpr int HS = HP + 4;             // LDS row stride of the [ROWS][HP] activation arrays
constexpr int OP = 16;                 // output width padded to one tile

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

struct IdxBlock { int32_t idx[SGMCMC_MLP_MAX_INLINE]; };

template <bool INLINE>
__device__ __forceinline__ void mlp_fwdbwd_body(const sgmcmc_mlp_args& P, const IdxBlock* IB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int IN = P.in_features, H1 = P.hidden1, H2 = P.hidden2, OUT = P.out_features;
  const int INp = (IN + 15) & ~15;  // K of the first layer, padded to 16
  const int XS = INp + 4;           // LDS row stride of the input slice (bank spread)
  float* xs = reinterpret_cast<float*>(smem_raw);  // [ROWS][XS]
  float* h1s = xs + ROWS * XS;                     // [ROWS][HS]  relu(layer 1)
  float* h2s = h1s + ROWS * HS;                    // [ROWS][HS]  relu(layer 2)
  float* d1s = h2s + ROWS * HS;                    // [ROWS][HS]  dL/d(pre-activation 1)
  float* d2s = d1s + ROWS * HS;                    // [ROWS][HS]  dL/d(pre-activation 2)
  float* lgs = d2s + ROWS * HS;                    // [ROWS][OP]  logits
  float* dfs = lgs + ROWS * OP;                    // [ROWS][OP]  dL/dlogits
  float* red = dfs + ROWS * OP;                    // [2][ROWS]   per-row loss / correct
  float* w2s = red + 2 * ROWS;                     // [HP][HS]    W2, zero padded
  float* w3s = w2s + HP * HS;                      // [OP][HS]    W3, zero padded
  float* bs = w3s + OP * HS;                       // [3][HP]     b1, b2, b3 (zero padded)
  float* f1p = bs + 3 * HP;                        // [KSPLIT][ROWS][HS] K-split partials of layer 1
  float* stg = f1p + KSPLIT * ROWS * HS;           // [kWaves][16][STG] per-wave store staging
  int64_t* rowp = reinterpret_cast<int64_t*>(stg + kWaves * 16 * STG);  // [ROWS] source row or -1
  int* ys = reinterpret_cast<int*>(rowp + ROWS);   // [ROWS] labels (-1 = padding row)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;  // MFMA fragment coordinates
  const int slice = blockIdx.x;
  const int row0 = slice * ROWS;
  float* __restrict__ gp = P.gpart + (int64_t)slice * P.gpart_stride;
  const float invB = P.grad_scale > 0.f ? P.grad_scale : 1.0f / (float)P.batch;
  int tp = 0;
#define MLP_TRACE() do { if (P.trace && blockIdx.x == 0 && tid == 0) P.trace[tp] = (int64_t)clock64(); ++tp; } while (0)
  MLP_TRACE();

  // ---- phase 0a: resolve the 16 row indices / labels; stage the small weights and biases
  if (P.args_src && blockIdx.x == 0 && tid < (P.args_bytes >> 2))
    reinterpret_cast<uint32_t*>(P.args_dst)[tid] = reinterpret_cast<const uint32_t*>(P.args_src)[tid];
  if (tid < ROWS) {
    const int b = row0 + tid;
    int64_t src = -1;
    if (b < P.batch) src = INLINE ? (int64_t)IB->idx[b] : (P.idx ? P.idx[b] : (int64_t)b);
    rowp[tid] = src;
    ys[tid] = src >= 0 ? (int)P.Y[src] : -1;
  }
  for (int e = tid; e < HP * HS; e += kThreads) {
    const int m = e / HS, k = e - m * HS;
    w2s[e] = (m < H2 && k < H1) ? P.W2[m * H1 + k] : 0.f;
  }
  for (int e = tid; e < OP * HS; e += kThreads) {
    const int m = e / HS, k = e - m * HS;
    w3s[e] = (m < OUT && k < H2) ? P.W3[m * H2 + k] : 0.f;
  }
  if (tid < HP) {
    bs[tid] = tid < H1 ? P.b1[tid] : 0.f;
    bs[HP + tid] = tid < H2 ? P.b2[tid] : 0.f;
    bs[2 * HP + tid] = tid < OUT ? P.b3[tid] : 0.f;
  }
  __syncthreads();
  // ---- phase 0b: gather the 16 input rows into LDS; every thread's loads are independent
  {
    const int n4 = INp >> 2, in4 = IN >> 2;  // IN % 4 == 0 is checked by the host
    const int total = ROWS * n4;
    for (int e = tid; e < total; e += kThreads) {
      const int rr = e / n4, c = e - rr * n4;
      const int64_t src = rowp[rr];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src >= 0 && c < in4) v = *reinterpret_cast<const float4*>(P.X + src * (int64_t)IN + 4 * c);
      *reinterpret_cast<float4*>(xs + rr * XS + 4 * c) = v;
    }
  }
  __syncthreads();
  MLP_TRACE();  // end of phase0

  // ---- forward 1: h1 = relu(x W1^T + b1).  Wave (nt, q): hidden units 16nt..16nt+15, K quarter q;
  //      all of a lane's W1 loads are issued before the first MFMA (W1 was just rewritten by the
  //      sampler kernel on other XCDs, so these are long-latency reads).
  {
    const int nt = wave & 3, q = wave >> 2;
    const int n = 16 * nt + r;
    const float* __restrict__ wrow = P.W1 + (int64_t)(n < H1 ? n : 0) * IN;
    const bool live = n < H1;
    const int steps = INp >> 4;
    const int j0 = (steps * q) / KSPLIT, j1 = (steps * (q + 1)) / KSPLIT;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int jb = j0; jb < j1; jb += F1_UNROLL) {
      float4 b[F1_UNROLL];
#pragma unroll
      for (int u = 0; u < F1_UNROLL; ++u) {
        const int k = 16 * (jb + u) + 4 * g;
        b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && jb + u < j1 && k < IN) b[u] = *reinterpret_cast<const float4*>(wrow + k);
      }
#pragma unroll
      for (int u = 0; u < F1_UNROLL; ++u) {
        if (jb + u < j1) {
          const float4 a = *reinterpret_cast<const float4*>(xs + r * XS + 16 * (jb + u) + 4 * g);
          acc0 = mfma4(a.x, b[u].x, acc0);
          acc1 = mfma4(a.y, b[u].y, acc1);
          acc0 = mfma4(a.z, b[u].z, acc0);
          acc1 = mfma4(a.w, b[u].w, acc1);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) f1p[(q * ROWS + 4 * g + i) * HS + n] = acc0[i] + acc1[i];
  }
  __syncthreads();
  for (int e = tid; e < ROWS * HP; e += kThreads) {
    const int row = e / HP, n = e - row * HP;
    float pre = bs[n];
#pragma unroll
    for (int q = 0; q < KSPLIT; ++q) pre += f1p[(q * ROWS + row) * HS + n];
    h1s[row * HS + n] = (n < H1 && pre > 0.f) ? pre : 0.f;
  }
  __syncthreads();
  MLP_TRACE();  // end of f1

  // ---- forward 2: h2 = relu(h1 W2^T + b2)   (waves 0-3, one column tile each)
  if (wave < 4) {
    const int n = 16 * wave + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 4; ++j) {
      const int k = 4 * j + g;
      acc = mfma4(h1s[r * HS + k], w2s[n * HS + k], acc);
    }
    const float bias = bs[HP + n];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float pre = acc[i] + bias;
      h2s[(4 * g + i) * HS + n] = (n < H2 && pre > 0.f) ? pre : 0.f;
    }
  }
  __syncthreads();
  MLP_TRACE();  // end of f2

  // ---- forward 3: logits = h2 W3^T + b3 (one tile: wave 0)
  if (wave == 0) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 4; ++j) {
      const int k = 4 * j + g;
      acc = mfma4(h2s[r * HS + k], w3s[r * HS + k], acc);
    }
    const float bias = bs[2 * HP + r];
#pragma unroll
    for (int i = 0; i < 4; ++i) lgs[(4 * g + i) * OP + r] = acc[i] + bias;
  }
  __syncthreads();
  MLP_TRACE();  // end of f3

  // ---- softmax cross-entropy, mean over the FULL batch (models/base.py:57-62,181-182):
  //      loss_i = logsumexp(f_i) - f_i[y_i];  dL/df = (softmax - onehot) / B.
  //      Thread (row, c) of the first 256 owns one logit; 16-lane groups reduce with shuffles.
  if (tid < ROWS * OP) {
    const int row = tid >> 4, c = tid & 15;
    const int y = ys[row];
    const float st = P.inv_softmax_temp;
    const float v = c < OUT ? lgs[row * OP + c] * st : -INFINITY;
    float m = v;
    int arg = c;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      const float om = __shfl_xor(m, off, 16);
      const int oa = __shfl_xor(arg, off, 16);
      if (om > m || (om == m && oa < arg)) { m = om; arg = oa; }  // first maximal index
    }
    float e = c < OUT ? expf(v - m) : 0.f;
    float se = e;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) se += __shfl_xor(se, off, 16);
    const float lse = m + logf(se);
    float d = 0.f;
    if (y >= 0 && c < OUT) d = (expf(v - lse) - (c == y ? 1.f : 0.f)) * invB * st;
    dfs[row * OP + c] = d;
    if (c == 0) {
      red[row] = y >= 0 ? lse - lgs[row * OP + y] * st : 0.f;
      red[ROWS + row] = (y >= 0 && arg == y) ? 1.f : 0.f;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f, c = 0.f;
    for (int i = 0; i < ROWS; ++i) { l += red[i]; c += red[ROWS + i]; }
    P.loss_part[slice] = l;
    P.correct_part[slice] = c;
  }
  MLP_TRACE();  // end of softmax

  // ---- backward 3: dW3 = df^T h2 (tile column w), db3, and d2 = (df W3) * [h2 > 0]
  if (wave < 4) {
    const int n = 16 * wave + r;  // h2 unit
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < ROWS / 4; ++j) {
      const int k = 4 * j + g;  // batch row
      acc = mfma4(dfs[k * OP + r], h2s[k * HS + n], acc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * g + i;  // output class
      if (m < OUT && n < H2) gp[P.off_W3 + m * H2 + n] = acc[i];
    }
    f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < OP / 4; ++j) {
      const int k = 4 * j + g;  // class
      dacc = mfma4(dfs[r * OP + k], w3s[k * HS + n], dacc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      d2s[row * HS + n] = h2s[row * HS + n] > 0.f ? dacc[i] : 0.f;
    }
  } else if (wave == 4 && lane < OUT) {
    float s = 0.f;
    for (int i = 0; i < ROWS; ++i) s += dfs[i * OP + lane];
    gp[P.off_b3 + lane] = s;
  }
  __syncthreads();
  MLP_TRACE();  // end of b3

  // ---- backward 2: dW2 = d2^T h1 (16 tiles: one per wave), db2, d1 = (d2 W2) * [h1 > 0]
  {
    const int m0 = 16 * (wave & 3), n = 16 * (wave >> 2) + r;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < ROWS / 4; ++j)
      acc = mfma4(d2s[(4 * j + g) * HS + m0 + r], h1s[(4 * j + g) * HS + n], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 4 * g + i;
      if (m < H2 && n < H1) gp[P.off_W2 + m * H1 + n] = acc[i];
    }
  }
  if (wave < 4) {
    const int n = 16 * wave + r;  // h1 unit
    f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 4; ++j) {
      const int k = 4 * j + g;  // h2 unit
      dacc = mfma4(d2s[r * HS + k], w2s[k * HS + n], dacc);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * g + i;
      d1s[row * HS + n] = h1s[row * HS + n] > 0.f ? dacc[i] : 0.f;
    }
  } else if (wave == 4 && lane < H2) {
    float s = 0.f;
    for (int i = 0; i < ROWS; ++i) s += d2s[i * HS + lane];
    gp[P.off_b2 + lane] = s;
  }
  __syncthreads();
  MLP_TRACE();  // end of b2

  // ---- backward 1: dW1 = d1^T x.  Wave (mt, c): hidden-unit tile mt, input-column tiles
  //      c, c+4, c+8, ...; each 16x16 result is transposed through a per-wave LDS patch so the
  //      store is one 16-byte access per lane (4 lanes cover a 64-byte row segment).
  {
    const int mt = wave & 3, c0 = wave >> 2;
    const int m0 = 16 * mt;
    float a[ROWS / 4];
#pragma unroll
    for (int j = 0; j < ROWS / 4; ++j) a[j] = d1s[(4 * j + g) * HS + m0 + r];
    float* __restrict__ patch = stg + wave * 16 * STG;
    const int tiles = INp >> 4;
    const int srow = lane >> 2, sc4 = (lane & 3) * 4;  // this lane's row / column group in the patch
    for (int t = c0; t < tiles; t += 2 * KSPLIT) {
      const int t1 = t + KSPLIT;
      const bool two = t1 < tiles;
      const int n0 = 16 * t, n1 = 16 * t1;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < ROWS / 4; ++j) {
        const float* xr = xs + (4 * j + g) * XS;
        acc0 = mfma4(a[j], xr[n0 + r], acc0);
        if (two) acc1 = mfma4(a[j], xr[n1 + r], acc1);
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && !two) break;
        const f32x4 acc = half ? acc1 : acc0;
        const int nb = half ? n1 : n0;
#pragma unroll
        for (int i = 0; i < 4; ++i) patch[(4 * g + i) * STG + r] = acc[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 v = *reinterpret_cast<const float4*>(patch + srow * STG + sc4);
        const int m = m0 + srow, col = nb + sc4;
        if (m < H1 && col < IN)
          *reinterpret_cast<float4*>(gp + P.off_W1 + (int64_t)m * IN + col) = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (wave == 4 && lane < H1) {
      float s = 0.f;
      for (int i = 0; i < ROWS; ++i) s += d1s[i * HS + lane];
      gp[P.off_b1 + lane] = s;
    }
  }
  if (P.trace) __syncthreads();
  MLP_TRACE();  // end of b1
#undef MLP_TRACE
}

__global__ __launch_bounds__(kThreads) void mlp_fwdbwd_kernel(sgmcmc_mlp_args P) {
  mlp_fwdbwd_body<false>(P, nullptr);
}
__global__ __launch_bounds__(kThreads) void mlp_fwdbwd_kernel_inline(sgmcmc_mlp_args P, IdxBlock IB) {
  mlp_fwdbwd_body<true>(P, &IB);
}

}  // namespace

extern "C" {

int64_t sgmcmc_mlp_lds_bytes(int in_features) {
  const int INp = (in_features + 15) & ~15;
  return (int64_t)sizeof(float) * (ROWS * (INp + 4) + 4 * ROWS * HS + 2 * ROWS * OP + 2 * ROWS +
                                   HP * HS + OP * HS + 3 * HP + KSPLIT * ROWS * HS +
                                   kWaves * 16 * STG) +
         (int64_t)sizeof(int64_t) * ROWS + (int64_t)sizeof(int) * ROWS;
}

int sgmcmc_mlp_fwdbwd(const sgmcmc_mlp_args* P, void* stream) {
  if (!P || P->batch <= 0 || P->in_features <= 0 || (P->in_features & 3)) return (int)hipErrorInvalidValue;
  if (P->hidden1 <= 0 || P->hidden1 > HP || P->hidden2 <= 0 || P->hidden2 > HP ||
      P->out_features <= 0 || P->out_features > OP)
    return (int)hipErrorInvalidValue;
  const int64_t lds = sgmcmc_mlp_lds_bytes(P->in_features);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwdbwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int slices = (P->batch + ROWS - 1) / ROWS;
  hipLaunchKernelGGL(mlp_fwdbwd_kernel, dim3((unsigned)slices), dim3(kThreads), (size_t)lds,
                     (hipStream_t)stream, *P);
  return (int)hipGetLastError();
}

int sgmcmc_step_parts_value(const sgmcmc_layout* L, const sgmcmc_step_args* A,
                            const sgmcmc_grad_parts* P, void* stream);  // sgmcmc_hip.hip

int sgmcmc_dense_step_direct(const sgmcmc_layout* L, const sgmcmc_mlp_args* mlp,
                             const sgmcmc_step_args* A, double num_data, const int64_t* idx_host,
                             void* stream) {
  if (!L || !mlp || !A || !idx_host || mlp->batch <= 0 || mlp->batch > SGMCMC_MLP_MAX_INLINE)
    return (int)hipErrorInvalidValue;
  if ((mlp->in_features & 3) || mlp->hidden1 > HP || mlp->hidden2 > HP || mlp->out_features > OP)
    return (int)hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwdbwd_kernel_inline),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  IdxBlock IB;
  for (int b = 0; b < mlp->batch; ++b) IB.idx[b] = (int32_t)idx_host[b];
  const int slices = (mlp->batch + ROWS - 1) / ROWS;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mlp_fwdbwd_kernel_inline, dim3((unsigned)slices), dim3(kThreads),
                     (size_t)sgmcmc_mlp_lds_bytes(mlp->in_features), s, *mlp, IB);
  sgmcmc_grad_parts G;
  G.gpart = mlp->gpart; G.loss_part = mlp->loss_part; G.correct_part = mlp->correct_part;
  G.stride = mlp->gpart_stride; G.num_data = num_data; G.n_slices = slices; G.batch = mlp->batch;
  return sgmcmc_step_parts_value(L, A, &G, s);
}

namespace {
__global__ __launch_bounds__(256) void accumulate_parts_kernel(const float* __restrict__ gpart,
                                                               int n_slices, int64_t stride,
                                                               double* __restrict__ acc,
                                                               float* __restrict__ out, int64_t n,
                                                               const float* __restrict__ loss_part,
                                                               const float* __restrict__ correct_part,
                                                               double* __restrict__ stats, int first) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) {
    double a = first ? 0.0 : acc[j];
    for (int s = 0; s < n_slices; ++s) a += (double)gpart[(int64_t)s * stride + j];
    acc[j] = a;
    if (out) out[j] = (float)a;
  }
  if (stats && j == 0) {
    double l = first ? 0.0 : stats[0], c = first ? 0.0 : stats[1];
    for (int s = 0; s < n_slices; ++s) { l += (double)loss_part[s]; c += (double)correct_part[s]; }
    stats[0] = l; stats[1] = c;
  }
}
}  // namespace

int sgmcmc_accumulate_parts(const float* gpart, int n_slices, int64_t stride, double* acc,
                            float* out_f32, int64_t n, const float* loss_part,
                            const float* correct_part, double* stats, int first, void* stream) {
  if (!gpart || !acc || n_slices <= 0 || n <= 0 || n > stride) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(accumulate_parts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, gpart, n_slices, stride, acc, out_f32, n, loss_part,
                     correct_part, stats, first);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- native stepper
struct sgmcmc_dense_stepper {
  hipGraph_t* graphs = nullptr;       // one replica per pinned slot (its host address is baked in)
  hipGraphExec_t* execs = nullptr;
  hipStream_t capture_stream = nullptr;
  hipEvent_t* events = nullptr;
  unsigned char* used = nullptr;
  unsigned char* pinned = nullptr;
  int n_ring = 0, batch = 0;
  int64_t slot_bytes = 0;
  uint64_t k = 0;
};

int sgmcmc_dense_stepper_create(const sgmcmc_layout* L, const sgmcmc_mlp_args* mlp,
                                const sgmcmc_step_args* A_geometry, double num_data,
                                void* dev_args, void* pinned, int n_ring, int64_t slot_bytes,
                                sgmcmc_dense_stepper** out) {
  if (!L || !mlp || !A_geometry || !dev_args || !pinned || n_ring <= 0 || !out ||
      slot_bytes < (int64_t)sizeof(sgmcmc_step_args) + 8 * (int64_t)mlp->batch || (slot_bytes & 7))
    return (int)hipErrorInvalidValue;
  sgmcmc_dense_stepper* S = new sgmcmc_dense_stepper();
  S->n_ring = n_ring; S->batch = mlp->batch; S->slot_bytes = slot_bytes;
  S->pinned = (unsigned char*)pinned;
  S->events = new hipEvent_t[n_ring];
  S->graphs = new hipGraph_t[n_ring]();
  S->execs = new hipGraphExec_t[n_ring]();
  S->used = new unsigned char[n_ring]();
  hipError_t err = hipStreamCreateWithFlags(&S->capture_stream, hipStreamNonBlocking);
  for (int i = 0; i < n_ring && err == hipSuccess; ++i)
    err = hipEventCreateWithFlags(&S->events[i], hipEventDisableTiming);
  if (err != hipSuccess) return (int)err;
  const int slices = (mlp->batch + SGMCMC_MLP_ROWS - 1) / SGMCMC_MLP_ROWS;
  sgmcmc_grad_parts G;
  G.gpart = mlp->gpart; G.loss_part = mlp->loss_part; G.correct_part = mlp->correct_part;
  G.stride = mlp->gpart_stride; G.num_data = num_data; G.n_slices = slices; G.batch = mlp->batch;
  // one eager pass first: loads the code objects and sets the LDS attribute outside capture
  // (writes scratch only: idx = identity, no argument forwarding)
  sgmcmc_mlp_args warm = *mlp;
  warm.idx = nullptr; warm.args_src = nullptr; warm.args_dst = nullptr; warm.args_bytes = 0;
  int rc = sgmcmc_mlp_fwdbwd(&warm, S->capture_stream);
  if (rc) return rc;
  err = hipStreamSynchronize(S->capture_stream);
  if (err != hipSuccess) return (int)err;
  for (int i = 0; i < n_ring; ++i) {
    unsigned char* slot = S->pinned + (int64_t)i * slot_bytes;
    sgmcmc_mlp_args m = *mlp;
    m.args_src = slot;
    m.args_dst = dev_args;
    m.args_bytes = (int32_t)sizeof(sgmcmc_step_args);
    m.idx = reinterpret_cast<const int64_t*>(slot + sizeof(sgmcmc_step_args));
    err = hipStreamBeginCapture(S->capture_stream, hipStreamCaptureModeThreadLocal);
    if (err != hipSuccess) return (int)err;
    rc = sgmcmc_mlp_fwdbwd(&m, S->capture_stream);
    if (!rc)
      rc = sgmcmc_step_indirect_parts(L, A_geometry, (const sgmcmc_step_args*)dev_args, &G,
                                      S->capture_stream);
    err = hipStreamEndCapture(S->capture_stream, &S->graphs[i]);
    if (rc) return rc;
    if (err != hipSuccess) return (int)err;
    err = hipGraphInstantiate(&S->execs[i], S->graphs[i], nullptr, nullptr, 0);
    if (err != hipSuccess) return (int)err;
  }
  *out = S;
  return 0;
}

int sgmcmc_dense_stepper_step(sgmcmc_dense_stepper* S, const sgmcmc_step_args* A,
                              const int64_t* idx_host, void* stream) {
  if (!S || !A || !idx_host) return (int)hipErrorInvalidValue;
  const int i = (int)(S->k % (uint64_t)S->n_ring);
  ++S->k;
  hipError_t err;
  if (S->used[i]) {  // the replay that last read this slot must have finished
    err = hipEventSynchronize(S->events[i]);
    if (err != hipSuccess) return (int)err;
  }
  unsigned char* slot = S->pinned + (int64_t)i * S->slot_bytes;
  memcpy(slot, A, sizeof(sgmcmc_step_args));
  memcpy(slot + sizeof(sgmcmc_step_args), idx_host, 8 * (size_t)S->batch);
  hipStream_t s = (hipStream_t)stream;
  err = hipGraphLaunch(S->execs[i], s);
  if (err != hipSuccess) return (int)err;
  S->used[i] = 1;
  return (int)hipEventRecord(S->events[i], s);
}

int sgmcmc_dense_stepper_destroy(sgmcmc_dense_stepper* S) {
  if (!S) return 0;
  for (int i = 0; i < S->n_ring; ++i) {
    if (S->execs && S->execs[i]) (void)hipGraphExecDestroy(S->execs[i]);
    if (S->graphs && S->graphs[i]) (void)hipGraphDestroy(S->graphs[i]);
    (void)hipEventDestroy(S->events[i]);
  }
  if (S->capture_stream) (void)hipStreamDestroy(S->capture_stream);
  delete[] S->events;
  delete[] S->graphs;
  delete[] S->execs;
  delete[] S->used;
  delete S;
  return 0;
}

}  // extern "C"
