/*
 * sgmcmc_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the SG-MCMC leapfrog transitions of
 * ratschlab/bnn_priors (bnn_priors/mcmc/{sgld,verlet_sgld,hmc}.py) on a flat
 * parameter arena, plus the noise specification (Philox4x32-10 -> 23-bit
 * uniforms -> Box-Muller with fixed fmaf polynomials) that the HIP kernels in
 * bnn_priors_amd/csrc implement independently.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path never does.
 *
 * Parity status: PINNED.  The per-tensor Python oracle (oracle/samplers.py)
 * is checked against golden vectors produced by importing the reference
 * (tests/golden/make_goldens.py); this file is checked against that oracle
 * and against the Random123 known-answer vectors for Philox4x32-10.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fmaf()/fma(); every other operation is individually rounded.  The device
 * code follows the same rule so element-wise results are bit-identical.
 *
 * Reference citations are path:line under /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 v1.x).  Not part of the
 * reference (which uses torch.randn_like, mcmc/verlet_sgld.py:163): this is
 * the build's own counter-based replacement so that noise is a pure function
 * of (seed, stream, draw, purpose, arena index). */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                                 uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0; k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
  philox4x32_10(ctr, key, out);
}

/* Counter layout of the noise spec (DESIGN.md "Noise"):
 *   key = (seed_lo, seed_hi)
 *   ctr = (quad_lo, quad_hi, draw_lo, purpose<<28 | stream<<16 | draw_hi16)
 * quad = arena_index >> 2; each counter yields the normals of elements
 * 4*quad .. 4*quad+3.  purpose: 0 step noise, 1 momentum refresh, 2 M-H uniform. */
static inline void noise_counter(uint64_t quad, uint64_t draw, uint32_t stream,
                                 uint32_t purpose, uint32_t ctr[4]) {
  ctr[0] = (uint32_t)quad;
  ctr[1] = (uint32_t)(quad >> 32);
  ctr[2] = (uint32_t)draw;
  ctr[3] = (purpose << 28) | ((stream & 0xFFFu) << 16) | (uint32_t)((draw >> 32) & 0xFFFFu);
}

static inline float u32_as_f32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* ln(u) for u = (2k+1) * 2^-24, k < 2^23 (so u in [2^-24, 1-2^-24], exact in
 * fp32).  Cephes-style polynomial, every fused op explicit. */
static inline float spec_logf(float u) {
  uint32_t bits = f32_as_u32(u);
  int e = (int)(bits >> 23) - 126;
  float m = u32_as_f32((bits & 0x007FFFFFu) | 0x3F000000u); /* [0.5, 1) */
  float x;
  if (m < 0.70710678f) { e -= 1; x = (m + m) - 1.0f; } else { x = m - 1.0f; }
  float z = x * x;
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
  float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(z, -0.5f, y);
  float r = x + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

/* (sin, cos)(2*pi*u) for u = (2k+1) * 2^-24: exact integer range reduction to
 * |phi| <= pi/4, cephes sinf/cosf kernels, quadrant rotation. */
static inline void spec_sincos2pi(uint32_t k23, float* s_out, float* c_out) {
  uint32_t odd = 2u * k23 + 1u;                 /* u * 2^24, < 2^24 */
  uint32_t q = (odd + (1u << 21)) >> 22;        /* rint(4u) in 0..4 */
  int32_t rint_ = (int32_t)odd - (int32_t)(q << 22); /* |.| <= 2^21 */
  float r = (float)rint_ * 5.9604644775390625e-08f;  /* * 2^-24, exact */
  float phi = r * 6.2831855f;
  float z = phi * phi;
  float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  float s = fmaf(ps * z, phi, phi);
  float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  float c = fmaf(pc, z * z, fmaf(z, -0.5f, 1.0f));
  switch (q & 3u) {
    case 0: *s_out = s;  *c_out = c;  break;
    case 1: *s_out = c;  *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
  }
}

static inline float spec_uniform(uint32_t x) {
  return (float)(2u * (x >> 9) + 1u) * 5.9604644775390625e-08f; /* exact */
}

/* 4 standard normals (fp32 values) for the quad. */
static inline void spec_normal4(uint64_t seed, uint32_t stream, uint64_t draw,
                                uint32_t purpose, uint64_t quad, float z[4]) {
  uint32_t ctr[4], key[2], x[4];
  key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
  noise_counter(quad, draw, stream, purpose, ctr);
  philox4x32_10(ctr, key, x);
  for (int h = 0; h < 2; ++h) {
    float u0 = spec_uniform(x[2 * h]);
    float t = -2.0f * spec_logf(u0);
    t = t < 0.0f ? 0.0f : t;
    float rad = sqrtf(t);
    float s, c;
    spec_sincos2pi(x[2 * h + 1] >> 9, &s, &c);
    z[2 * h] = rad * c;
    z[2 * h + 1] = rad * s;
  }
}

/* normals for arena elements [start, start+n) */
void oracle_normals_f32(uint64_t seed, uint32_t stream, uint64_t draw, uint32_t purpose,
                        int64_t start, int64_t n, float* out) {
  int64_t i = start, end = start + n;
  while (i < end) {
    float z[4];
    uint64_t quad = (uint64_t)i >> 2;
    spec_normal4(seed, stream, draw, purpose, quad, z);
    for (int l = (int)(i & 3); l < 4 && i < end; ++l, ++i) out[i - start] = z[l];
  }
}

/* the M-H uniform: lane 0 of quad 0 with purpose = 2 (DESIGN.md "Noise") */
double oracle_mh_uniform(uint64_t seed, uint32_t stream, uint64_t draw) {
  uint32_t ctr[4], key[2], x[4];
  key[0] = (uint32_t)seed; key[1] = (uint32_t)(seed >> 32);
  noise_counter(0, draw, stream, 2u, ctr);
  philox4x32_10(ctr, key, x);
  return (double)spec_uniform(x[0]);
}

/* --------------------------------------------------- flat-arena transitions */
/* Segment s covers arena elements [off[s], off[s]+numel[s]).  sums is
 * double[n_seg][6]: g.g, g.m_old, g.m_new, m_old.m_old, m_new.m_new, theta_old.g
 * accumulated in fp64 from exact fp64 products, in element order. */

#define FLAG_INITIAL 1u
#define FLAG_FINAL 2u
#define FLAG_SAVE 4u

typedef struct {
  double grad_v, bhn, bh, mom_decay, noise_std, alpha; /* group scalars */
  uint64_t seed, draw;
  uint32_t stream, flags;
} oracle_step_params;

/* VerletSGLD._step_fn, mcmc/verlet_sgld.py:149-197 (fp32 instantiation).
 *   n     = xi * noise_std                              :163
 *   m_new = fma(g, grad_lr, n)                          :164-165
 *   m_new = fma(m_old, mom_decay, m_new) if mom_decay>0 :166-167
 *   theta = fma(m_new, bh*M, theta)   (not final)       :192-193
 *   v     = v*alpha + ((1-alpha)*g)*g (not final)       :195-197
 * save_state copies theta,g,m first (:72-83). */
#define DEFINE_VERLET(NAME, T, FMA)                                                         \
  void NAME(T* theta, const T* g, T* m, T* v, T* prev_theta, T* prev_g, T* prev_m,          \
            int n_seg, const int64_t* off, const int64_t* numel, const double* M,           \
            const oracle_step_params* P, double* sums) {                                     \
    const T noise_std = (T)P->noise_std, mom_decay = (T)P->mom_decay;                        \
    const T alpha = (T)P->alpha, one_m_alpha = (T)(1 - P->alpha);                            \
    for (int s = 0; s < n_seg; ++s) {                                                        \
      const T grad_lr = (T)(-.5 * P->grad_v * P->bhn * M[s]);                                \
      const T step = (T)(P->bh * M[s]);                                                      \
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;                                 \
      for (int64_t j = 0; j < numel[s]; ++j) {                                               \
        const int64_t i = off[s] + j;                                                        \
        float z[4];                                                                          \
        spec_normal4(P->seed, P->stream, P->draw, 0u, (uint64_t)i >> 2, z);                  \
        const T xi = (T)z[i & 3];                                                            \
        const T gi = g[i], mo = m[i], th = theta[i];                                         \
        if (P->flags & FLAG_SAVE) { prev_theta[i] = th; prev_g[i] = gi;                      \
          if (prev_m) prev_m[i] = mo; }                                                      \
        T mn = xi * noise_std;                                                               \
        mn = FMA(gi, grad_lr, mn);                                                           \
        if (P->mom_decay > 0) mn = FMA(mo, mom_decay, mn);                                   \
        a0 = fma((double)gi, (double)gi, a0);                                                \
        a1 = fma((double)gi, (double)mo, a1);                                                \
        a2 = fma((double)gi, (double)mn, a2);                                                \
        a3 = fma((double)mo, (double)mo, a3);                                                \
        a4 = fma((double)mn, (double)mn, a4);                                                \
        a5 = fma((double)th, (double)gi, a5);                                                \
        m[i] = mn;                                                                           \
        if (!(P->flags & FLAG_FINAL)) {                                                      \
          theta[i] = FMA(mn, step, th);                                                      \
          v[i] = v[i] * alpha + (one_m_alpha * gi) * gi;                                     \
        }                                                                                    \
      }                                                                                      \
      double* o = sums + 6 * s;                                                              \
      o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;                      \
    }                                                                                        \
  }
DEFINE_VERLET(oracle_verlet_step_f32, float, fmaf)
DEFINE_VERLET(oracle_verlet_step_f64, double, fma)

/* HMC._step_fn, mcmc/hmc.py:41-79: no noise, in-place kick
 *   m = fma(g, grad_lr, m) :64-65 ; theta = fma(m, bh*M, theta) :74 ; v as above :77-79 */
#define DEFINE_HMC(NAME, T, FMA)                                                            \
  void NAME(T* theta, const T* g, T* m, T* v, T* prev_theta, T* prev_g, T* prev_m,          \
            int n_seg, const int64_t* off, const int64_t* numel, const double* M,           \
            const oracle_step_params* P, double* sums) {                                     \
    const T alpha = (T)P->alpha, one_m_alpha = (T)(1 - P->alpha);                            \
    for (int s = 0; s < n_seg; ++s) {                                                        \
      const T grad_lr = (T)(-.5 * P->grad_v * P->bhn * M[s]);                                \
      const T step = (T)(P->bh * M[s]);                                                      \
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;                                 \
      for (int64_t j = 0; j < numel[s]; ++j) {                                               \
        const int64_t i = off[s] + j;                                                        \
        const T gi = g[i], mo = m[i], th = theta[i];                                         \
        if (P->flags & FLAG_SAVE) { prev_theta[i] = th; prev_g[i] = gi; prev_m[i] = mo; }    \
        const T mn = FMA(gi, grad_lr, mo);                                                   \
        a0 = fma((double)gi, (double)gi, a0);                                                \
        a1 = fma((double)gi, (double)mo, a1);                                                \
        a2 = fma((double)gi, (double)mn, a2);                                                \
        a3 = fma((double)mo, (double)mo, a3);                                                \
        a4 = fma((double)mn, (double)mn, a4);                                                \
        a5 = fma((double)th, (double)gi, a5);                                                \
        m[i] = mn;                                                                           \
        if (!(P->flags & FLAG_FINAL)) {                                                      \
          theta[i] = FMA(mn, step, th);                                                      \
          v[i] = v[i] * alpha + (one_m_alpha * gi) * gi;                                     \
        }                                                                                    \
      }                                                                                      \
      double* o = sums + 6 * s;                                                              \
      o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;                      \
    }                                                                                        \
  }
DEFINE_HMC(oracle_hmc_step_f32, float, fmaf)
DEFINE_HMC(oracle_hmc_step_f64, double, fma)

/* SGLD._step_fn, mcmc/sgld.py:119-154.  P->mom_decay carries `momentum` (a),
 * P->bhn carries hn, P->bh carries h, P->noise_std = sqrt(2(1-a)T) (0 => no draw).
 *   a>0: m = m*a ; m = fma(g, -hn*M, m)   :131     a=0: m = g * (-hn*M)  :134
 *   T>0: m = fma(xi, noise_std, m)        :142
 *   theta = fma(m, h*M, theta)            :150 ; v as above :153-154
 * sums[3] = m_old.m_old (a>0) or m_det.m_det with m_det the momentum before the
 * noise is added (a=0; :135-137); sums[4] = m_new.m_new (after noise). */
#define DEFINE_SGLD(NAME, T, FMA)                                                           \
  void NAME(T* theta, const T* g, T* m, T* v, int n_seg, const int64_t* off,                \
            const int64_t* numel, const double* M, const oracle_step_params* P,              \
            double* sums) {                                                                  \
    const T a = (T)P->mom_decay, noise_std = (T)P->noise_std;                                \
    const T alpha = (T)P->alpha, one_m_alpha = (T)(1 - P->alpha);                            \
    for (int s = 0; s < n_seg; ++s) {                                                        \
      const T grad_lr = (T)(-P->bhn * M[s]);                                                 \
      const T step = (T)(P->bh * M[s]);                                                      \
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0;                                 \
      for (int64_t j = 0; j < numel[s]; ++j) {                                               \
        const int64_t i = off[s] + j;                                                        \
        const T gi = g[i], th = theta[i];                                                    \
        const T mo = (P->mom_decay > 0) ? m[i] : (T)0;                                       \
        T mn, mt;                                                                            \
        if (P->mom_decay > 0) { mn = mo * a; mn = FMA(gi, grad_lr, mn); mt = mo; }           \
        else { mn = gi * grad_lr; mt = mn; }                                                 \
        if (P->noise_std > 0 && !(P->flags & FLAG_FINAL)) {                                  \
          float z[4];                                                                        \
          spec_normal4(P->seed, P->stream, P->draw, 0u, (uint64_t)i >> 2, z);                \
          mn = FMA((T)z[i & 3], noise_std, mn);                                              \
        }                                                                                    \
        a0 = fma((double)gi, (double)gi, a0);                                                \
        a1 = fma((double)gi, (double)mo, a1);                                                \
        a2 = fma((double)gi, (double)mn, a2);                                                \
        a3 = fma((double)mt, (double)mt, a3);                                                \
        a4 = fma((double)mn, (double)mn, a4);                                                \
        a5 = fma((double)th, (double)gi, a5);                                                \
        if (!(P->flags & FLAG_FINAL)) {                                                      \
          if (P->mom_decay > 0) m[i] = mn;                                                   \
          theta[i] = FMA(mn, step, th);                                                      \
          v[i] = v[i] * alpha + (one_m_alpha * gi) * gi;                                     \
        }                                                                                    \
      }                                                                                      \
      double* o = sums + 6 * s;                                                              \
      o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5;                      \
    }                                                                                        \
  }
DEFINE_SGLD(oracle_sgld_step_f32, float, fmaf)
DEFINE_SGLD(oracle_sgld_step_f64, double, fma)

/* SGLD.sample_momentum, mcmc/sgld.py:57-69:
 *   keep==0: m = xi*std ; else m = fma(xi, std, m*sqrt(keep)), std = sqrt(T(1-keep)) */
#define DEFINE_SAMPLE_M(NAME, T, FMA)                                                       \
  void NAME(T* m, int n_seg, const int64_t* off, const int64_t* numel, double std_,         \
            double keep, uint64_t seed, uint32_t stream, uint64_t draw) {                    \
    const T sd = (T)std_, sk = (T)sqrt(keep);                                                \
    for (int s = 0; s < n_seg; ++s)                                                          \
      for (int64_t j = 0; j < numel[s]; ++j) {                                               \
        const int64_t i = off[s] + j;                                                        \
        float z[4];                                                                          \
        spec_normal4(seed, stream, draw, 1u, (uint64_t)i >> 2, z);                           \
        const T xi = (T)z[i & 3];                                                            \
        m[i] = (keep == 0.0) ? xi * sd : FMA(xi, sd, m[i] * sk);                             \
      }                                                                                      \
  }
DEFINE_SAMPLE_M(oracle_sample_momentum_f32, float, fmaf)
DEFINE_SAMPLE_M(oracle_sample_momentum_f64, double, fma)
