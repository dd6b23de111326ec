/* Measured alternatives of libsgmcmc_hip.so -- NOT part of the shipped C ABI.
 *
 * Everything declared here lost to the default kernels inside the captured googleresnet step (DESIGN.md section 3 has the
 * numbers) and is compiled only into a library built with -DSGMCMC_ALTERNATIVES (SGMCMC_ALTERNATIVES=1 in the
 * environment of bnn_priors_amd._hip.build(); the tests that cover these entry points run only against such a build):
 *
 *   - sgmcmc_conv3x3_fx / _bnin          a residual block's first BatchNorm + ReLU folded into the second convolution's
 *                                        operand staging (statistics through per-XCD integer atomics): 83 -> 77 launches,
 *                                        the same steps/s;
 *   - sgmcmc_conv3x3_bwd_part            the weight-gradient half of a convolution backward on a side stream: every
 *                                        fork + join edge of a replayed graph costs ~19 us, 1,137 -> 845 steps/s;
 *   - sgmcmc_conv3x3_bn_bwd              the BatchNorm backward formed inside the convolution-gradient launch: needs a
 *                                        sums launch of its own, 1,059 vs 1,122 steps/s;
 *   - sgmcmc_conv3x3_bwd_uniform         (round 6) data gradient + weight-gradient slab of an (image, band) by ONE kind of
 *                                        workgroup from one staging: 16.7-16.95 vs 17.2-17.55 us per launch in the step,
 *                                        twice the slabs at 32 channels (reduction +4.6 us): 1,244 vs 1,242.5 steps/s.
 *
 * Same conventions as sgmcmc_hip.h (device pointers, hipStream_t as void*, hipError_t returned as int). */
#ifndef SGMCMC_HIP_ALTERNATIVES_H
#define SGMCMC_HIP_ALTERNATIVES_H
#include "sgmcmc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A residual block's first BatchNorm + ReLU folded into its second convolution (models/google_resnet.py:34-43):
 *   sgmcmc_conv3x3_fx    y = conv3x3(x, w) and the batch statistics of y ADDED to `fx` -- per-XCD integer slots,
 *                        int64 [8][5][channels][16] (sgmcmc_fx_slot_int64(channels) elements), ZEROED by the caller:
 *                        integer atomics commute, so the totals carry the same bits whatever ran where;
 *   sgmcmc_conv3x3_bnin  y = conv3x3(relu(BatchNorm(x)), w): training-mode BatchNorm on the statistics in bn->fx and
 *                        the ReLU applied while the operands are staged, bn->h <- relu(BatchNorm(x)) as a side output
 *                        (the backward needs it), saved / running statistics (or the log slot) written as
 *                        sgmcmc_bn_train_fwd[_log] does; `stats`: y's own statistics per slice as sgmcmc_conv3x3.
 * The BatchNorm's apply launch and its read of x disappear. */
typedef struct {
  const int64_t* fx;
  const float *gamma, *beta;
  float *save_mean, *save_invstd, *running_mean, *running_var;
  double* stat_log;
  double momentum, eps;
  float* h;
} sgmcmc_bn_in;
int64_t sgmcmc_fx_slot_int64(int channels);
int sgmcmc_conv3x3_fx(const float* x, const float* w, float* y, int n_img, int channels, int hw, int64_t* fx,
                      void* stream);
int sgmcmc_conv3x3_bnin(const float* x, const float* w, float* y, int n_img, int channels, int hw, double* stats,
                        const sgmcmc_bn_in* bn, void* stream);


/* One half of sgmcmc_conv3x3_bwd_ex on its own: which = 1 the data gradient (dx, with the epilogues of `epi`, which may
 * be NULL), which = 2 the weight-gradient slabs (always deferred: *deferred_slabs receives their number).  The halves are
 * independent given dy, so a caller can run the weight gradient on a second stream, off the critical path of the
 * backward pass; same workgroups, same bits as the merged launch. */
int sgmcmc_conv3x3_bwd_part(const float* x, const float* w, const float* dy, float* dx,
                            const sgmcmc_conv_bwd_epilogue* epi, float* scratch, int n_img, int channels, int hw,
                            int which, int* deferred_slabs, void* stream);

/* ---- BatchNorm backward folded into the convolution's gradient launch (csrc/conv_fused_hip.inc) ----------
 * For a "conv3x3 -> BatchNorm(train) -> [+ shortcut] -> ReLU" pair of the ResNet trunk (google_resnet.py:34-43,
 * 77-90), given dout = the gradient w.r.t. the pair's (post-ReLU) output `out`, y = the convolution's output:
 *
 *   (sgmcmc_bn_bwd_sums of sgmcmc_hip.h supplies the per-slice partial sums it needs, as a launch of its own)
 *   sgmcmc_conv3x3_bn_bwd: BOTH gradients of the convolution with
 *       dy = gamma*invstd * (dz - sum dz / M - xhat * sum(dz*xhat) / M),  xhat = (y - mean)*invstd
 *     formed while the operands are staged (what sgmcmc_bn_train_bwd's second launch would have written), plus
 *     dgamma = sum dz*xhat, dbeta = sum dz.  With e_dout / e_out given, dx += e_dout*[e_out > 0]: the gradient
 *     that reaches the convolution's INPUT through a shortcut whose ReLU mask is e_out (replaces an add launch).
 *     dw is left as *n_slabs partial slabs in `scratch` (sgmcmc_conv3x3_wrw_scratch_floats) for
 *     sgmcmc_wrw_reduce_many, as sgmcmc_conv3x3_bwd does with deferred_slabs. */
typedef struct {
  const float *dout, *mask_out, *y, *mean, *invstd, *gamma;
  const double* sums;
  int32_t n_sums, reserved;
  float *dgamma, *dbeta;
  const float *e_dout, *e_out;
} sgmcmc_conv_bn_bwd_args;
int sgmcmc_conv3x3_bn_bwd(const float* x, const float* w, float* dx, float* scratch,
                          const sgmcmc_conv_bn_bwd_args* A, int n_img, int channels, int hw, int* n_slabs,
                          void* stream);


/* The UNIFORM backward convolution (csrc/conv_uni_hip.inc) at (channels, hw) = (16, 32) and (32, 16): both gradients of
 * sgmcmc_conv3x3_bwd_ex (same epilogues; no groups, no wrw_mult) by 8-wave workgroups that stage the dy band (with halo),
 * the x band and the transposed weight slab ONCE and run the data gradient's and the weight gradient's MFMAs one after the
 * other.  dx and the sums partials carry the bits of sgmcmc_conv3x3_bwd_ex; the weight gradient is left as
 * sgmcmc_conv3x3_bwd_uniform_slabs(...) tap-major slabs of channels^2 * 9 floats in `scratch` (one reduction job for
 * sgmcmc_wrw_reduce_many, taps = 9): one per workgroup = per two (image, band) items at 16 channels, per item at 32. */
int sgmcmc_conv3x3_bwd_uniform_slabs(int n_img, int channels, int hw);
int sgmcmc_conv3x3_bwd_uniform(const float* x, const float* w, const float* dy, float* dx,
                               const sgmcmc_conv_bwd_epilogue* epi, float* scratch, int n_img, int channels, int hw,
                               void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGMCMC_HIP_ALTERNATIVES_H */
