/*
 * sgmcmc_hip.h -- C ABI of the MI355X (gfx950) SG-MCMC leapfrog engine.
 *
 * This is the drop-in boundary underneath the Python optimizer classes
 * bnn_priors_amd.mcmc.{SGLD,VerletSGLD,HMC}: plain pointers, sizes and PODs,
 * no torch types.  Every entry point
 *   - takes device pointers that the caller owns (nothing is allocated here),
 *   - enqueues its kernels on the given hipStream_t (passed as void*) and
 *     returns immediately (no host synchronisation inside),
 *   - returns a hipError_t as int (0 = success).
 *
 * The reference has no FFI: its samplers are pure-Python torch.optim.Optimizer
 * subclasses.  Each entry point therefore cites the reference *method* whose
 * per-tensor Python loop it replaces (paths under /root/reference/bnn_priors/).
 *
 * Data model
 * ----------
 * A sampler instance owns an ARENA tiled into CHUNKs of SGMCMC_CHUNK elements.
 * Parameter tensor ("segment") s occupies ceil(numel_s / chunk_elems)
 * consecutive chunks starting at first_chunk_s (chunk_elems is a property of the layout:
 * 4096, or 1024 for small models so that they still fill many CUs); the optimizer-owned state
 * arrays m (momentum), v (RMSprop square_avg), prev_theta / prev_g / prev_m
 * (Metropolis-Hastings roll-back copies) live in the arena at element offset
 * chunk * chunk_elems.  theta (the nn.Parameter storage) and g (its .grad)
 * are NOT moved: kernels reach them through the per-segment base pointers, so
 * autograd can keep allocating gradients wherever it likes.
 */
#ifndef SGMCMC_HIP_H
#define SGMCMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGMCMC_ABI_VERSION 6
#define SGMCMC_CHUNK 4096 /* default elements per arena chunk = 256 threads x 4 items x 4 elements */
#define SGMCMC_CHUNK_SMALL 1024 /* small models: one item per thread, 4x more workgroups */
#define SGMCMC_NSUMS 6
#define SGMCMC_PSTRIDE 8   /* doubles per chunk in `partials`: the 6 sums, [6] log-prior, [7] spare */
#define SGMCMC_MLP_ROWS 16 /* batch rows per workgroup of the fused dense-net kernel */

enum { SGMCMC_F32 = 0, SGMCMC_F64 = 1 };
enum { SGMCMC_VERLET = 0, SGMCMC_HMC = 1, SGMCMC_SGLD = 2 };
/* flags of sgmcmc_step_args */
enum {
  SGMCMC_INITIAL = 1,      /* first transition after an M-H point (verlet_sgld.py:86, hmc.py:46) */
  SGMCMC_FINAL = 2,        /* last transition: theta and v are not modified (verlet_sgld.py:119) */
  SGMCMC_SAVE_STATE = 4,   /* copy theta,g,m to prev_* first (verlet_sgld.py:72-83) */
  SGMCMC_CALC_METRICS = 8, /* update est_temperature / est_config_temp */
  SGMCMC_UNALIGNED = 16,   /* some theta/g base pointer is not 16-byte aligned: scalar loads */
  SGMCMC_NO_MOMENTUM = 32, /* SGLD with momentum == 0: m is neither read nor written */
  SGMCMC_SMALL_FINALIZE = 64, /* few chunks: one workgroup finalizes all segments and leaves
                                 scalars[3] = sum_s(delta_energy_s + point_energy_s) for the
                                 gradient / momentum of THIS transition */
  SGMCMC_DEFER_FINALIZE = 256, /* launch only the update kernel; the per-segment bookkeeping is run
                                  later by sgmcmc_finalize or inside the next sgmcmc_dense_step_direct
                                  (it is not an input of the next gradient evaluation) */
  SGMCMC_WITH_LOG_PRIOR = 128, /* with SMALL_FINALIZE + CALC_METRICS: partials[.][6] holds the
                                 fused priors' log-density partials (sgmcmc_grad_reduce_prior):
                                 finish state[s].aux and scalars[2] in the same launch */
  SGMCMC_INLINE_PRIOR = 512 /* sgmcmc_step_indirect only (float32 arena; priors of kind <= CAUCHY without linked
                               scales): g holds the likelihood gradient alone and the update kernel adds the
                               closed-form prior gradient in flight (and, on a CALC_METRICS step, leaves the
                               log-density partials for WITH_LOG_PRIOR) -- no sgmcmc_prior_grad launch before it.
                               g holds the full gradient afterwards, as after sgmcmc_prior_grad. */
};
/* element-wise priors the kernels differentiate in closed form (prior/loc_scale.py, prior/distributions.py:75-79)
 * and the element-wise hyper-priors of a hierarchical scale (prior/transformed.py:55-87, hierarchical.py:17-104):
 *   GENNORM            log p = -log(2 scale) - lgamma(1/beta) + log beta - (|theta - loc| / scale)^beta; beta in prior_df
 *   GAMMA_SOFTPLUS     theta is the raw parameter s, the value x = softplus(s) ~ Gamma(concentration = prior_loc,
 *                      rate = prior_scale): log p = c log r - lgamma(c) + (c-1) log x - r x
 *   UNIFORM_CDF        value x = low + (high - low) Phi(s), low = prior_loc, high = prior_scale: log p = -log(high-low)
 *   HALFCAUCHY_SOFTPLUS value x = softplus(s) * multiplier (prior_loc) ~ HalfCauchy(scale = prior_scale), the density
 *                      evaluated AT x: log p = log(2 / (pi g)) - log(1 + (x/g)^2)
 *   IMPROPER_SOFTPLUS  value x = softplus(s), NO density (log p = 0, no gradient of its own): the learnable scale of the
 *                      empirical-Bayes priors (prior/loc_scale.py:100-103 under prior/empirical_bayes.py:24-38)
 * A weight segment whose scale_link > 0 takes its scale from the VALUE of hyper segment scale_link - 1 (one element)
 * at launch time instead of prior_scale; sgmcmc_prior_grad then also adds
 * -(1/N) (d/dscale sum_j log p(theta_j)) dx/ds to the hyper segment's gradient (flags & SGMCMC_PRIOR_HAS_LINKS). */
enum { SGMCMC_PRIOR_NONE = 0, SGMCMC_PRIOR_NORMAL = 1, SGMCMC_PRIOR_LAPLACE = 2,
       SGMCMC_PRIOR_STUDENT_T = 3, SGMCMC_PRIOR_CAUCHY = 4, SGMCMC_PRIOR_GENNORM = 5,
       SGMCMC_PRIOR_GAMMA_SOFTPLUS = 6, SGMCMC_PRIOR_UNIFORM_CDF = 7, SGMCMC_PRIOR_HALFCAUCHY_SOFTPLUS = 8,
       SGMCMC_PRIOR_IMPROPER_SOFTPLUS = 9 };
/* flags of sgmcmc_prior_grad: some segment is linked to a hyper segment / some segment's kind is beyond CAUCHY
 * (without either the lean kernel for the four constant-scale families is launched) */
enum { SGMCMC_PRIOR_HAS_LINKS = 1, SGMCMC_PRIOR_FULL = 2 };

/* One parameter tensor.  Device-resident array, written by the host. */
typedef struct {
  void* theta;          /* base of the parameter storage (dtype of the layout) */
  void* g;              /* base of its gradient */
  double M;             /* state['preconditioner'] (sgld.py:47-52) */
  int64_t numel;
  int64_t first_chunk;
  int64_t noise_base;   /* Philox element index of element 0 (multiple of 4) */
  int32_t prior_kind;   /* SGMCMC_PRIOR_*; NONE = g already holds the full gradient */
  int32_t scale_link;   /* > 0: the prior's scale is the value of hyper segment scale_link - 1 (see above) */
  double prior_loc, prior_scale, prior_df;
} sgmcmc_segment;

typedef struct {
  int32_t seg;     /* owning segment */
  int32_t n_valid; /* elements of this chunk inside the segment (1..SGMCMC_CHUNK) */
} sgmcmc_chunk;

/* Per-segment running scalars, device-resident, fp64.  The host reads this
 * array back only when a caller asks for a metric or an energy. */
typedef struct {
  double sums[SGMCMC_NSUMS]; /* g.g, g.m_old, g.m_new, m_old.m_old, m_new.m_new, theta_old.g */
  double delta_energy;       /* state['delta_energy']            (verlet_sgld.py:170-175) */
  double prev_delta;         /* state['prev_new_momentum_delta'] (verlet_sgld.py:176) */
  double est_temperature;    /* (verlet_sgld.py:178-187) */
  double est_config_temp;    /* (verlet_sgld.py:189) */
  double point_energy;       /* last value computed by sgmcmc_delta_energy */
  double aux;                /* sgmcmc_segment_sum result (e.g. sum of v) */
} sgmcmc_seg_state;

typedef struct {
  int32_t dtype; /* SGMCMC_F32 | SGMCMC_F64 */
  int32_t n_seg;
  int64_t n_chunks;
  int64_t chunk_elems; /* SGMCMC_CHUNK or SGMCMC_CHUNK_SMALL: elements per chunk of THIS layout */
  const sgmcmc_segment* segs; /* device [n_seg] */
  const sgmcmc_chunk* chunks; /* device [n_chunks] */
  void* m;                    /* device arenas, n_chunks * SGMCMC_CHUNK elements each */
  void* v;
  void* prev_theta;
  void* prev_g;
  void* prev_m;
  double* partials;        /* device [n_chunks][SGMCMC_PSTRIDE] scratch */
  sgmcmc_seg_state* state; /* device [n_seg] */
  double* scalars;         /* device [8] outputs: [0] sgmcmc_delta_energy total, [1] non-finite flag,
                              [2] fused log-prior total, [3] energy total of the last transition
                              (SGMCMC_SMALL_FINALIZE only), [4] minibatch loss, [5] minibatch
                              accuracy (sgmcmc_grad_reduce_prior) */
  uint32_t prior_flags;    /* SGMCMC_PRIOR_HAS_LINKS | SGMCMC_PRIOR_FULL as they hold for THIS segment table: the
                              entry points that only carry the lean prior code (sgmcmc_grad_reduce_prior, the
                              gradient-assembling step kernels, SGMCMC_INLINE_PRIOR) return hipErrorInvalidValue
                              when either bit is set instead of evaluating a family they do not implement */
  uint32_t reserved;
} sgmcmc_layout;

/* Scalars of one transition of one parameter group, computed by the host in
 * double exactly as the reference's _update_group_fn does. */
typedef struct {
  int32_t kind;  /* SGMCMC_VERLET | SGMCMC_HMC | SGMCMC_SGLD */
  uint32_t flags;
  int32_t seg_begin, seg_end;     /* the group's segments [begin, end) */
  int64_t chunk_begin, chunk_end; /* and their chunks */
  double num_data;   /* N */
  double b2h2;       /* lr / N           (verlet_sgld.py:139) */
  double bh;         /* sqrt(lr / N)     (verlet_sgld.py:140; sgld.py:116 'h') */
  double bhn;        /* sqrt(lr * N)     (verlet_sgld.py:141; sgld.py:115 'hn') */
  double mom_decay;  /* verlet_sgld.py:144,98,131 ; SGLD: momentum a */
  double grad_v;     /* verlet_sgld.py:145,99,132 */
  double noise_std;  /* verlet_sgld.py:146,100,133 ; sgld.py:117 ; 0 => no draw */
  double rmsprop_alpha;
  double grad_clamp; /* > 0: clamp g to +-grad_clamp in flight (inference.py:219-220); 0 = off */
  uint64_t seed;     /* Philox key */
  uint64_t draw;     /* sweep counter */
  uint32_t stream;   /* chain id */
  uint32_t reserved;
} sgmcmc_step_args;

int sgmcmc_abi_version(void);
/* hash of the sources the library was built from (csrc/ + include/), as the build handed it in; "unstamped" if it did not */
const char* sgmcmc_source_sha(void);
const char* sgmcmc_error_string(int err);

/* One transition of one parameter group: fused noise + momentum + position +
 * RMSprop update and the six per-segment dot products, then the per-segment
 * energy / temperature bookkeeping.
 * Replaces SGLD._step_internal + _step_fn (mcmc/sgld.py:88-154),
 * VerletSGLD.initial_step/step/final_step + _step_fn (mcmc/verlet_sgld.py:85-197),
 * HMC._step_fn (mcmc/hmc.py:41-79). */
int sgmcmc_step(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream);

/* As sgmcmc_step, with the fused update kernel (not the finalize kernel) launched so that ev_start /
 * ev_stop carry ITS begin / end timestamps (hipExtLaunchKernelGGL): hipEventElapsedTime(ev_start,
 * ev_stop) is that kernel's execution time, measured live (bench.py roofline).  Events are hipEvent_t
 * created by sgmcmc_event_create. */
int sgmcmc_step_timed(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream,
                      void* ev_start, void* ev_stop);
/* As sgmcmc_step, but the kernels read the transition's scalars from DEVICE memory
 * (*A_dev, same struct) when they run.  Launch geometry and kernel selection (kind, dtype,
 * seg/chunk ranges, SGMCMC_UNALIGNED, SGMCMC_NO_MOMENTUM) are taken from *A_host and must not
 * differ in *A_dev (the kernels take chunk_begin from *A_host and request their chunk's m and v before *A_dev has
 * arrived: L->m and L->v must be non-NULL and padded to whole chunks whatever the flags; NULL: hipErrorInvalidValue).
 * This is what makes the launch capturable in a hipGraph that is replayed with a different learning rate / draw
 * counter / flags every step. */
int sgmcmc_step_indirect(const sgmcmc_layout* L, const sgmcmc_step_args* A_host,
                         const sgmcmc_step_args* A_dev, void* stream);

/* Likelihood gradient given as per-slice partials (written by sgmcmc_mlp_fwdbwd): element j of
 * segment s lives at gpart[slice*stride + noise_base_s + j]. */
typedef struct {
  const float* gpart;
  const float* loss_part;    /* [n_slices] */
  const float* correct_part; /* [n_slices] */
  int64_t stride;
  double num_data;
  int32_t n_slices, batch;
} sgmcmc_grad_parts;

/* sgmcmc_step_indirect with the gradient assembled in flight: g <- sum over slices (fixed
 * order) - (1/N) dlog p/dtheta, used by the transition AND written back to each segment's g;
 * on metric steps also the log-prior partials; scalars[4], [5] <- minibatch loss / accuracy.
 * fp32 layouts only.  One launch instead of sgmcmc_grad_reduce_prior + sgmcmc_step_indirect. */
int sgmcmc_step_indirect_parts(const sgmcmc_layout* L, const sgmcmc_step_args* A_host,
                               const sgmcmc_step_args* A_dev, const sgmcmc_grad_parts* G,
                               void* stream);
int sgmcmc_event_create(void** ev);
int sgmcmc_event_destroy(void* ev);
/* hipEventRecord on `stream`: lets a caller bracket ANY entry point of this library with events on the
 * stream its kernels are launched on (bench.py's live roofline timing of the convolution kernels). */
int sgmcmc_event_record(void* ev, void* stream);
/* The NEXT kernel this library launches (from the calling thread's next entry-point call) carries the two
 * events in its dispatch packet, as sgmcmc_step_timed does for the update kernel: the elapsed time between
 * them is that kernel's own duration.  Used by bench.py to time the convolution kernels live. */
int sgmcmc_time_next_launch(void* ev_start, void* ev_stop);
int sgmcmc_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms); /* synchronises on ev_stop */

/* m <- sqrt(keep)*m + std*xi   (keep == 0: m <- std*xi).
 * Replaces SGLD.sample_momentum (mcmc/sgld.py:57-69). */
int sgmcmc_sample_momentum(const sgmcmc_layout* L, double std, double keep, uint64_t seed,
                           uint32_t stream, uint64_t draw, void* stream_);

/* theta,g,m <- prev_theta,prev_g,prev_m.
 * Replaces the roll-back loop of VerletSGLD.maybe_reject (mcmc/verlet_sgld.py:62-69). */
int sgmcmc_restore(const sgmcmc_layout* L, int restore_momentum, uint32_t flags, void* stream);

/* scalars[0] <- sum_s (state[s].delta_energy + point_energy_s), in segment order,
 * point_energy_s = (M^2 N^2 b2h2 / 8) g.g (Verlet) or 0.5 m.m (HMC), from the
 * CURRENT g / m.  Replaces VerletSGLD.delta_energy's loop + _point_energy
 * (mcmc/verlet_sgld.py:27-47, mcmc/hmc.py:32-33); the caller adds (U - U_prev) * N. */
int sgmcmc_delta_energy(const sgmcmc_layout* L, int kind, double num_data, double b2h2,
                        double grad_clamp, uint32_t flags, void* stream);

/* state[s].aux <- sum of v over segment s (which = 0), of m*m (1) or g*g (2).
 * which = 0 replaces the square_avg.mean() loop of SGLD.update_preconditioner
 * (mcmc/sgld.py:156-179); the host finishes (mean + eps, min, ^(-1/4)). */
int sgmcmc_segment_sum(const sgmcmc_layout* L, int which, uint32_t flags, void* stream);

/* g <- g - (1/N) dlog p(theta)/dtheta for every segment with prior_kind != NONE
 * (element-wise Normal / Laplace / Student-t / Cauchy / generalised normal with scalar loc, scale, df / beta,
 * and the hyper-priors of hierarchical scales; see SGMCMC_PRIOR_*), i.e. what
 * autograd adds for the "- log_prior / N" term of potential_avg (models/base.py:72-77,
 * prior/base.py:57-58, prior/loc_scale.py:34-35,66-67,74-77).  With calc_log_prob != 0 also
 * state[s].aux <- sum_j log p(theta_j) (fp64) and scalars[2] <- the total over segments.
 * One launch for all tensors instead of ~10 ATen launches per prior tensor per step. */
int sgmcmc_prior_grad(const sgmcmc_layout* L, double num_data, int calc_log_prob, uint32_t flags,
                      void* stream);

/* ---- fused dense classifier (ClassificationDenseNet, models/dense_nets.py:48-67) ---------- */
/* Forward + backward of -(1/B) sum_i log softmax(net(x_i) / T)[y_i] for
 * net = Linear(in,h1)-ReLU-Linear(h1,h2)-ReLU-Linear(h2,out), fp32, in ONE launch
 * (csrc/mlp_hip.inc).  Rows are gathered as X[idx[b]], Y[idx[b]] (idx == NULL: b itself).
 * Workgroup s writes its 16 rows' PARTIAL gradients to gpart + s*gpart_stride at the given
 * per-tensor offsets, and loss_part[s] = sum of its rows' losses, correct_part[s] = #correct.
 * Limits: in % 4 == 0, h1,h2 <= 64, out <= 16, LDS(in) <= 160 KiB. */
typedef struct {
  const float* X;      /* [n_data, in]  device */
  const int64_t* Y;    /* [n_data]      device */
  const int64_t* idx;  /* [batch] device, or NULL */
  const float *W1, *b1, *W2, *b2, *W3, *b3;
  float* gpart;        /* [ceil(batch/16)][gpart_stride] */
  float* loss_part;    /* [ceil(batch/16)] */
  float* correct_part; /* [ceil(batch/16)] */
  int64_t gpart_stride;
  int64_t off_W1, off_b1, off_W2, off_b2, off_W3, off_b3;
  int32_t batch, in_features, hidden1, hidden2, out_features;
  float inv_softmax_temp;
  int64_t* trace; /* optional (else NULL): 9 clock64() stamps at the phase boundaries of workgroup 0 */
  /* optional (else NULL): workgroup 0 copies args_bytes (multiple of 4, <= 1024) from args_src
   * (typically pinned HOST memory, read over PCIe) to args_dst (device) so that later kernels of
   * the same graph find the step's scalars in device memory without a separate copy node */
  const void* args_src;
  void* args_dst;
  int32_t args_bytes;
  /* > 0: dL/dlogits is scaled by this instead of 1/batch.  The exact full-data gradient
   * (inference_reject.py:18-33) is the sum over mega-batches of  -sum_i log p_i / N : scale = 1/N */
  float grad_scale;
  /* optional (else NULL): sgmcmc_mlp_split_scratch_floats(batch) floats of device scratch.  When
   * given, sgmcmc_dense_step_direct runs the forward/backward as TWO launches that spread the two
   * 784-wide contractions over 4x more workgroups (lower latency; same partial-gradient layout). */
  float* split_scratch;
} sgmcmc_mlp_args;

int64_t sgmcmc_mlp_split_scratch_floats(int batch);

int sgmcmc_mlp_fwdbwd(const sgmcmc_mlp_args* P, void* stream);
int64_t sgmcmc_mlp_lds_bytes(int in_features);

/* g <- sum_{s<n_slices} gpart[s*stride + noise_base_seg + j]  (fixed order)  - (1/N) dlog p/dtheta,
 * written to each segment's g; i.e. sgmcmc_prior_grad with the likelihood gradient taken from
 * per-slice partials (fp32 layouts only).  Also scalars[4] <- sum(loss_part)/batch,
 * scalars[5] <- sum(correct_part)/batch.  The log-density partials are produced when
 * (A_dev ? A_dev->flags : flags) has SGMCMC_CALC_METRICS and are finished by the following
 * sgmcmc_step* launch carrying SGMCMC_WITH_LOG_PRIOR. */
int sgmcmc_grad_reduce_prior(const sgmcmc_layout* L, const float* gpart, int n_slices,
                             int64_t stride, const float* loss_part, const float* correct_part,
                             int batch, double num_data, uint32_t flags,
                             const sgmcmc_step_args* A_dev, void* stream);

/* ---- native replay of the fused dense leapfrog step ------------------------------------- */
/* A "stepper" owns n_ring replicas of a hipGraph  mlp_fwdbwd -> step_indirect_parts (update +
 * finalize), captured on an internal stream at creation, one per pinned host slot.  One call to
 * sgmcmc_dense_stepper_step per leapfrog step: copies *A (per-step scalars) and idx[batch] (row
 * indices) into the next slot and launches that slot's replica on `stream`; the first kernel
 * reads the slot straight from host memory (row indices) and forwards the scalars to `dev_args`,
 * so there is no separate copy node.  A slot is reused only after the replay that read it has
 * completed (event wait): the host may run ahead of the GPU by up to n_ring steps.
 * Memory is the caller's: dev_args (device, >= sizeof(sgmcmc_step_args)), pinned (host-pinned,
 * n_ring*slot_bytes), slot_bytes >= sizeof(sgmcmc_step_args) + 8*batch and a multiple of 8;
 * mlp->idx and mlp->args_* are set per slot by the library. */
typedef struct sgmcmc_dense_stepper sgmcmc_dense_stepper;
int sgmcmc_dense_stepper_create(const sgmcmc_layout* L, const sgmcmc_mlp_args* mlp,
                                const sgmcmc_step_args* A_geometry, double num_data,
                                void* dev_args, void* pinned, int n_ring, int64_t slot_bytes,
                                sgmcmc_dense_stepper** out);
int sgmcmc_dense_stepper_step(sgmcmc_dense_stepper* S, const sgmcmc_step_args* A,
                              const int64_t* idx_host, void* stream);
int sgmcmc_dense_stepper_destroy(sgmcmc_dense_stepper* S);

/* The same leapfrog step as three DIRECT launches with every argument passed by value in
 * the kernel-argument segment -- including the minibatch's row indices (int32, batch <=
 * SGMCMC_MLP_MAX_INLINE) -- so there is no staging buffer, no copy and no graph: the runtime
 * snapshots the arguments at launch time.  mlp->idx / args_* are ignored. */
#define SGMCMC_MLP_MAX_INLINE 256
int sgmcmc_dense_step_direct(const sgmcmc_layout* L, const sgmcmc_mlp_args* mlp,
                             const sgmcmc_step_args* A, double num_data, const int64_t* idx_host,
                             const sgmcmc_step_args* A_pending, void* stream);
/* A_pending != NULL: the bookkeeping of an EARLIER transition launched with
 * SGMCMC_DEFER_FINALIZE (small-finalize layouts only) is executed by one extra workgroup of this
 * call's gradient kernel, i.e. concurrently with the forward/backward and before this call's
 * update kernel.  If A->flags has SGMCMC_DEFER_FINALIZE this call's own bookkeeping is left
 * pending in turn; otherwise it is launched as usual. */

/* The same three launches for SEVERAL independent chains at once (grid dimension y = chain): chains of one
 * architecture and one schedule, stepped in lock-step, each with its own weights, data order, sampler arena and
 * Philox stream (chain_id).  `chains_dev`: device array [n_chains], written once by the caller; `chain0_host`: a
 * host copy of element 0 (launch geometry); A: the transition's scalars, common to all chains (A->stream is
 * ignored: every chain draws from ITS stream; A must carry SMALL_FINALIZE and DEFER_FINALIZE); idx_host:
 * [n_chains][batch] data-set rows (16-bit: data sets of up to 65,536 rows); A_pending as sgmcmc_dense_step_direct.
 * Chain c's results are bit-identical to the same chain stepped alone by sgmcmc_dense_step_direct. */
#define SGMCMC_MAX_CHAINS 8
#define SGMCMC_MLP_BATCH_MULTI 128
typedef struct {
  sgmcmc_mlp_args mlp;
  sgmcmc_layout layout;
  double num_data;
  uint32_t chain_id, reserved;
} sgmcmc_dense_chain;
int sgmcmc_dense_step_multi(const sgmcmc_dense_chain* chains_dev, const sgmcmc_dense_chain* chain0_host, int n_chains,
                            const sgmcmc_step_args* A, const uint16_t* idx_host, const sgmcmc_step_args* A_pending,
                            void* stream);

/* The per-segment bookkeeping of a transition that was launched with SGMCMC_DEFER_FINALIZE. */
int sgmcmc_finalize(const sgmcmc_layout* L, const sgmcmc_step_args* A, void* stream);

/* acc[j] <- (first ? 0 : acc[j]) + sum_{s<n_slices} gpart[s*stride + j]   for j < n  (fp64
 * accumulator, fixed order); out_f32[j] <- (float)acc[j] when out_f32 != NULL;
 * stats[0] += sum loss_part, stats[1] += sum correct_part (zeroed first when `first`).
 * Accumulates the mega-batches of the exact full-data gradient pass. */
int sgmcmc_accumulate_parts(const float* gpart, int n_slices, int64_t stride, double* acc,
                            float* out_f32, int64_t n, const float* loss_part,
                            const float* correct_part, double* stats, int first, void* stream);

/* 3x3, stride 1, zero-pad 1 convolution of the ResNet trunk on the fp32 matrix pipe, NCHW fp32,
 * `channels` -> `channels` (16 @ 32x32, 32 @ 16x16, 64 @ 8x8; anything else: hipErrorInvalidValue
 * and the caller keeps the shape on its library path).  Replaces the convolution inside
 * autograd's forward / backward of R1 (inference.py:215-223; models/google_resnet.py:34-43):
 *   transpose_w = 0:  y[n,co,p]  = sum_{ci,r,s} x[n,ci,p+(r-1,s-1)] w[co,ci,r,s]      (forward)
 *   transpose_w = 1:  y[n,ci,p]  = sum_{co,r,s} x[n,co,p-(r-1,s-1)] w[co,ci,r,s]      (data gradient,
 *                     x = the gradient w.r.t. the forward output)
 *   stats (forward only, may be NULL): [channels][sgmcmc_conv3x3_stat_slices(...)][2] doubles, the
 *   per-band (sum y, sum (y - band mean)^2) of every output channel -- the batch statistics of the BatchNorm that
 *   follows, taken from the accumulators (pass them to sgmcmc_bn_train_fwd as stats_in). */
int sgmcmc_conv3x3_stat_slices(int n_img, int channels, int hw);
int sgmcmc_conv3x3(const float* x, const float* w, float* y, int n_img, int channels, int hw,
                   int transpose_w, double* stats, void* stream);

/* Weight gradient of the same convolution: dw[co,ci,r,s] = sum_{n,p} dy[n,co,p] x[n,ci,p+(r-1,s-1)].
 * `scratch` holds sgmcmc_conv3x3_wrw_scratch_floats(...) floats of per-workgroup partial slabs that a
 * second launch sums in a fixed order (deterministic; no atomics).  -1 / hipErrorInvalidValue for
 * shapes outside the table above. */
int64_t sgmcmc_conv3x3_wrw_scratch_floats(int n_img, int channels, int hw);
int sgmcmc_conv3x3_wrw(const float* x, const float* dy, float* dw, float* scratch, int n_img,
                       int channels, int hw, void* stream);
/* Evaluation mode: y = relu?(BatchNorm_eval(conv3x3(x, w)) [+ residual]) in ONE launch -- with running statistics the
 * BatchNorm that follows a convolution is a per-channel affine map, applied to the accumulator tile in the epilogue
 * (invstd = 1 / sqrtf(var + eps); ((v - mean) * invstd) * gamma + beta; + residual; ReLU: sgmcmc_bn_eval_fwd's
 * arithmetic on sgmcmc_conv3x3's output, the same bits as the two launches).  Replaces conv -> BatchNorm2d.eval() ->
 * (+ shortcut) -> ReLU of models/google_resnet.py:34-56 inside the test-set passes (inference.py:199-213,
 * exp_utils.py:250-340).  residual may be NULL; the three trunk shapes. */
int sgmcmc_conv3x3_bn_eval(const float* x, const float* w, const float* gamma, const float* beta,
                           const float* running_mean, const float* running_var, double eps, const float* residual,
                           int relu, float* y, int n_img, int channels, int hw, void* stream);
/* Both gradients in one launch (they are independent and share the GPU): dx as sgmcmc_conv3x3 with
 * transpose_w = 1 on dy, dw and scratch as sgmcmc_conv3x3_wrw.  Results are bit-identical to the two
 * separate calls.
 * With `deferred_slabs` != NULL the slab reduction is NOT launched: *deferred_slabs receives the number of
 * slabs in `scratch` and the caller sums them later -- typically all weight gradients of a backward pass
 * in ONE launch of sgmcmc_wrw_reduce_many (same summation order: same bits). */
#define SGMCMC_REDUCE_JOBS 32
typedef struct sgmcmc_reduce_job {
  const float* part; /* [n_slabs][numel] */
  float* out;        /* [numel] */
  int32_t n_slabs, numel;
  int32_t taps;      /* <= 1: out[j] = sum_p part[p][j].  > 1: the slabs are TAP-MAJOR ([taps][numel / taps], what the 3x3
                      * weight-gradient kernels write: coalesced) and out is [numel / taps][taps] = [co][ci][r,s] */
  int32_t reserved;
} sgmcmc_reduce_job;
int sgmcmc_conv3x3_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                       float* scratch, int n_img, int channels, int hw, int* deferred_slabs, void* stream);
/* ... with dx += e_dout * [e_out > 0] in the data gradient's epilogue: the gradient a residual block's identity
 * shortcut carries back to the block input (models/google_resnet.py:52-56: out = relu(bn2(conv2(h)) + x)), added where
 * the first convolution's data gradient is produced instead of by a separate element-wise pass. */
int sgmcmc_conv3x3_bwd_add(const float* x, const float* w, const float* dy, float* dx, const float* e_dout,
                           const float* e_out, float* dw, float* scratch, int n_img, int channels, int hw,
                           int* deferred_slabs, void* stream);
/* The general form: the data gradient's epilogue may add the shortcut's gradient (e_dout masked by [e_out > 0]; e_out NULL: e_dout as it is -- see mask_dx)
 * and / or leave the partial sums of the BatchNorm backward that CONSUMES dx as its incoming gradient (s_*: that
 * BatchNorm's input y, its post-ReLU output, its saved mean / invstd): s_partial[(c * S + slice) * 2 + {0,1}] =
 * sum dz, sum dz * xhat over the slice, dz = dx * [s_out > 0], xhat = (s_y - mean_c) * invstd_c, S =
 * sgmcmc_conv3x3_stat_slices(n_img, channels, hw) slices per channel -- what sgmcmc_bn_bwd_sums would compute in a
 * launch of its own (same quantities, another summation grouping); feed them to sgmcmc_bn_bwd_dx.
 * Replaces BatchNorm2d's backward reductions (models/google_resnet.py:34-43 inside inference.py:215-223). */
typedef struct {
  const float *e_dout, *e_out;
  const float *s_y, *s_out, *s_mean, *s_invstd;
  double* s_partial;
  int32_t group_imgs; /* > 0: the launch carries groups of this many images (sgmcmc_bn_train_fwd, GROUPS) and s_mean /
                       * s_invstd are [groups][channels]; 0: one batch */
  int32_t wrw_mult;   /* sgmcmc_conv3x3_bwd_ex only; > 1: every weight-gradient workgroup sums over wrw_mult times as many
                       * (image, band) items, i.e. *deferred_slabs shrinks by that factor -- for launches that carry
                       * several minibatches (the slab count of ONE minibatch, each slab the sum over more images;
                       * another summation grouping than wrw_mult = 1: equal to rounding); 0 / 1: the default */
  int32_t mask_dx;    /* with s_partial: dx is STORED as dz = dx * [s_out > 0].  Everything that consumes the gradient of a
                       * BatchNorm + ReLU output forms exactly that from it (the BatchNorm's own backward; the shortcut add
                       * of the residual block before), so neither has to read s_out again: feed such a dx to
                       * sgmcmc_bn_bwd_dx with relu = 0 (y unused) and as e_dout with e_out = NULL (added as it is).
                       * Same bits as masking at the consumers. */
  int32_t reserved;
} sgmcmc_conv_bwd_epilogue;
int sgmcmc_conv3x3_bwd_ex(const float* x, const float* w, const float* dy, float* dx,
                          const sgmcmc_conv_bwd_epilogue* epi, float* dw, float* scratch, int n_img, int channels,
                          int hw, int* deferred_slabs, void* stream);
int sgmcmc_wrw_reduce_many(const sgmcmc_reduce_job* jobs, int n_jobs, void* stream);

#define SGMCMC_FRAG_JOBS 24   /* convolutions per launch of sgmcmc_conv3x3_prepare_weights */
/* ---- the same three contractions, PERSISTENT kernels on prepared weight fragments (csrc/conv2_hip.inc; round 3) ----
 * Replaces the convolutions of models/google_resnet.py:11-43 inside a gradient evaluation, as sgmcmc_conv3x3 /
 * sgmcmc_conv3x3_bwd_ex do, for the same three shapes.  WHERE THEY RUN: launches that carry several minibatches -- the
 * grouped exact full-data pass (inference_reject.py:18-33; graphed.py), 512 and more images per launch with the SAME
 * weights for a whole pass -- where persistent, double-buffered workgroups are 10-20 % faster (tools/conv_lab at 512
 * images: backward 62 -> 51 us at 32 channels, 65 -> 51 us at 64; forward 35 -> 28 us at 64; the pass 190 -> 172 ms).
 * The 128-image leapfrog step keeps sgmcmc_conv3x3[_bwd_ex] (there these are slower: 1,131 vs 1,154 steps/s,
 * DESIGN.md section 3).  Differences:
 *   - the weights are read as MFMA fragments that sgmcmc_conv3x3_prepare_weights leaves in caller-owned buffers of
 *     channels^2 * 9 floats each (forward order and transposed + flipped for the data gradient): ONE launch for all
 *     convolutions of a gradient evaluation (up to SGMCMC_FRAG_JOBS per launch; more are split over launches);
 *   - items of 4 image rows; workgroups are persistent over a stream of items, XCD-aware (an image's items, its
 *     channel tiles and both of its gradients are processed on XCD = image mod 8);
 *   - statistics / backward-sum partials: [channels][sgmcmc_conv3x3_frag_stat_slices(...)][2] doubles, slice =
 *     image * (hw / 4) + band -- equal parts, as sgmcmc_bn_train_fwd / sgmcmc_bn_bwd_dx expect;
 *   - weight-gradient slabs: `scratch` = [channels / 16][P][9][16][channels] floats, P = *deferred_slabs: one reduction
 *     job PER 16-output-channel tile t (part = scratch + t * P * 144 * channels, out = dw + t * 144 * channels,
 *     numel = 144 * channels, taps = 9).  With dw != NULL and deferred_slabs == NULL the reductions are launched here.
 * Results agree with the round-2 kernels up to fp32 summation order; runs are bitwise reproducible. */
typedef struct sgmcmc_frag_job {
  const float* w;  /* [channels][channels][3][3] */
  float* fwd;      /* channels^2 * 9 floats, or NULL */
  float* dgrad;    /* channels^2 * 9 floats, or NULL */
  int32_t channels, reserved;
} sgmcmc_frag_job;
int sgmcmc_conv3x3_prepare_weights(const sgmcmc_frag_job* jobs, int n_jobs, void* stream);
int sgmcmc_conv3x3_frag_stat_slices(int n_img, int channels, int hw);
int64_t sgmcmc_conv3x3_frag_scratch_floats(int n_img, int channels, int hw);
int sgmcmc_conv3x3_frag_fwd(const float* x, const float* frag_fwd, float* y, int n_img, int channels, int hw,
                            double* stats, void* stream);
int sgmcmc_conv3x3_frag_bwd(const float* x, const float* frag_dgrad, const float* dy, float* dx,
                            const sgmcmc_conv_bwd_epilogue* epi, float* dw, float* scratch, int n_img, int channels,
                            int hw, int* deferred_slabs, void* stream);


/* The two convolutions that open a down-sampling ResNet block, as one operator (they read the same input;
 * the 1x1 shortcut's operand is the 3x3's centre tap): models/google_resnet.py:77-90.
 *   y_main[n,co,oy,ox]  = sum_{ci,r,s} x[n,ci,2oy+r-1,2ox+s-1] w_main[co,ci,r,s]     (3x3, stride 2, pad 1)
 *   y_short[n,co,oy,ox] = sum_ci x[n,ci,2oy,2ox] w_short[co,ci]                        (1x1, stride 2)
 * cin -> 2 cin channels, hwi -> hwi/2 pixels, for (cin, hwi) = (16, 32) and (32, 16).  stats_main /
 * stats_short (both or neither): [2 cin][sgmcmc_conv_down_stat_slices(...)][2] per-band (sum, centred sum
 * of squares) of the two outputs, as for sgmcmc_conv3x3. */
int sgmcmc_conv_down_stat_slices(int n_img, int cin, int hwi);
int sgmcmc_conv_down_fwd(const float* x, const float* w_main, const float* w_short, float* y_main,
                         float* y_short, double* stats_main, double* stats_short, int n_img, int cin,
                         int hwi, void* stream);
/* All three gradients in one launch: dx (both paths summed: the transposed convolutions, evaluated per
 * parity class of the input pixel so that no multiply is spent on zeros), dw_main, dw_short.  `scratch`:
 * sgmcmc_conv_down_scratch_floats(...) floats, laid out [n_slabs][18 cin^2] then [n_slabs][2 cin^2];
 * `deferred_slabs` as for sgmcmc_conv3x3_bwd (two jobs for sgmcmc_wrw_reduce_many). */
int64_t sgmcmc_conv_down_scratch_floats(int n_img, int cin, int hwi);
int sgmcmc_conv_down_bwd(const float* x, const float* w_main, const float* w_short, const float* dy_main,
                         const float* dy_short, float* dx, float* dw_main, float* dw_short, float* scratch,
                         int n_img, int cin, int hwi, int* deferred_slabs, void* stream);
/* ... with the SUMS half of sgmcmc_conv_bwd_epilogue (e_dout / e_out must be NULL): the partial sums of the BatchNorm
 * backward whose incoming gradient dx is, [cin][sgmcmc_conv_down_bwd_sum_slices(...)][2] doubles. */
int sgmcmc_conv_down_bwd_sum_slices(int n_img, int cin, int hwi);
int sgmcmc_conv_down_bwd_ex(const float* x, const float* w_main, const float* w_short, const float* dy_main,
                            const float* dy_short, float* dx, const sgmcmc_conv_bwd_epilogue* epi, float* dw_main,
                            float* dw_short, float* scratch, int n_img, int cin, int hwi, int* deferred_slabs,
                            void* stream);

/* The stem convolution, 3 -> 16 channels, 3x3 / stride 1 / pad 1 on 32x32 images (google_resnet.py:96-100):
 * forward (+ optional per-band statistics [16][4 n_img][2]) and weight gradient (the images take no
 * gradient); scratch / deferred_slabs as for sgmcmc_conv3x3_bwd. */
int sgmcmc_conv_stem_fwd(const float* x, const float* w, float* y, double* stats, int n_img, void* stream);
int64_t sgmcmc_conv_stem_scratch_floats(int n_img);
int sgmcmc_conv_stem_wrw(const float* x, const float* dy, float* dw, float* scratch, int n_img,
                         int* deferred_slabs, void* stream);

/* The convolutional classifier's first layer (models/conv_nets.py:44-51): 1 -> 50 channels, 3x3 / stride 1 /
 * pad 1 on 28x28 images, without its bias (see sgmcmc_bias_relu_pool_*): forward and weight gradient
 * (n_img slabs of 450 floats; scratch / deferred_slabs as for sgmcmc_conv3x3_bwd). */
int sgmcmc_conv_first_fwd(const float* x, const float* w, float* y, int n_img, void* stream);
int64_t sgmcmc_conv_first_scratch_floats(int n_img);
int sgmcmc_conv_first_wrw(const float* x, const float* dy, float* dw, float* scratch, int n_img,
                          int* deferred_slabs, void* stream);
/* ... and the layer WITH its tail, Conv2d(1, 50, 3, padding=1) -> + bias -> ReLU -> MaxPool2d(2)
 * (models/conv_nets.py:46-57), one launch each way; the 50 x 28 x 28 map is never written.
 *   fwd: pooled [n][50][14][14] and code [n][50][14][14] bytes -- bits 0-1 the window position of the first maximum
 *        in scan order (NaNs win, as ATen's max_pool2d), bit 2 set when max + bias > 0.
 *   bwd: from the gradient w.r.t. pooled and code: weight-gradient slabs [P][450] at scratch and, with want_bias,
 *        bias-gradient slabs [P][50] behind them (P = 7 n_img bands); reduced into dw / dbias here, or --
 *        deferred_slabs != NULL -- *deferred_slabs = P and the caller runs sgmcmc_wrw_reduce_many over both. */
int sgmcmc_conv_first_pool_fwd(const float* x, const float* w, const float* bias, float* pooled, uint8_t* code,
                               int n_img, void* stream);
int64_t sgmcmc_conv_first_pool_scratch_floats(int n_img);
int sgmcmc_conv_first_pool_bwd(const float* x, const float* dpooled, const uint8_t* code, float* dw, float* dbias,
                               float* scratch, int n_img, int want_bias, int* deferred_slabs, void* stream);

/* The convolutional classifier's SECOND convolution (models/conv_nets.py:46-70): 50 -> 50 channels, 3x3 / stride 1 /
 * pad 1 on 14x14 maps, without its bias (see sgmcmc_bias_relu_pool_*), on the fp32 matrix pipe (csrc/conv50_hip.inc).
 *   sgmcmc_conv50: transpose_w = 0 forward, 1 data gradient (x = the gradient w.r.t. the forward output);
 *   sgmcmc_conv50_bwd: data gradient (dx may be NULL: weight gradient only) and weight gradient in one launch;
 *     scratch: sgmcmc_conv50_scratch_floats(n_img) floats of partial slabs; deferred_slabs as sgmcmc_conv3x3_bwd. */
int sgmcmc_conv50(const float* x, const float* w, float* y, int n_img, int transpose_w, void* stream);
int64_t sgmcmc_conv50_scratch_floats(int n_img);
int sgmcmc_conv50_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* scratch, int n_img,
                      int* deferred_slabs, void* stream);
/* Round 3: the forward also leaves wT[ci][co][rs] = w[co][ci][8 - rs] (2500 * 9 floats; NULL: not wanted) -- the weights
 * as the data gradient's rows -- so that the backward stages them with the forward's straight copy instead of strided
 * gathers: sgmcmc_conv50_bwd_t(x, wT, dy, dx (required), dw, scratch, ...) = sgmcmc_conv50_bwd with w replaced by the
 * wT of a forward on the SAME weights; a data gradient alone is sgmcmc_conv50_fwd(dy, wT, dx, NULL, ...). */
int sgmcmc_conv50_fwd(const float* x, const float* w, float* y, float* wT, int n_img, void* stream);
int sgmcmc_conv50_bwd_t(const float* x, const float* wT, const float* dy, float* dx, float* dw, float* scratch,
                        int n_img, int* deferred_slabs, void* stream);
/* ... and the layer WITH its tail, Conv2d(50, 50, 3, padding=1) -> + bias -> ReLU -> MaxPool2d(2)
 * (models/conv_nets.py:58-66), one launch each way; the 50 x 14 x 14 map and its gradient exist in LDS only.
 *   fwd: pooled [n][50][7][7], code [n][50][7][7] bytes (encoding of sgmcmc_conv_first_pool_fwd), wT as
 *        sgmcmc_conv50_fwd (may be NULL).
 *   bwd: from the gradient w.r.t. pooled and code: dx [n][50][14][14] (required), weight-gradient slabs [P][22500]
 *        (tap-major) at scratch and, with want_bias, bias-gradient slabs [P][50] behind them (P = ceil(n_img / 2));
 *        reduced into dw / dbias here, or -- deferred_slabs != NULL -- left to sgmcmc_wrw_reduce_many. */
int sgmcmc_conv50_pool_fwd(const float* x, const float* w, const float* bias, float* pooled, uint8_t* code, float* wT,
                           int n_img, void* stream);
int64_t sgmcmc_conv50_pool_scratch_floats(int n_img);
int sgmcmc_conv50_pool_bwd(const float* x, const float* wT, const float* dpooled, const uint8_t* code, float* dx,
                           float* dw, float* dbias, float* scratch, int n_img, int want_bias, int* deferred_slabs,
                           void* stream);

/* Training-mode batch normalisation over (N, H*W) per channel of an NCHW fp32 tensor, fused with the
 * optional residual add and ReLU that follow it in the ResNet trunk (models/google_resnet.py:34-43,
 * 77-90; replaces nn.BatchNorm2d + `+ shortcut` + ReLU inside R1's autograd graph):
 *   y = relu?(gamma * (x - mean) * invstd + beta [+ residual]);  save_mean / save_invstd are
 *   written for the backward; running_mean / running_var (both or neither) get nn.BatchNorm2d's momentum
 *   update with the unbiased batch variance.  `plane` = H*W must be a multiple of 4.
 * Backward: dz = dy * (y > 0) when relu;  dresidual (optional) = dz;  dbeta = sum dz;
 *   dgamma = sum dz * xhat;  dx = gamma * invstd * (dz - dbeta/M - xhat * dgamma/M).
 * `scratch`: sgmcmc_bn_scratch_doubles(n, channels, plane, groups) doubles of per-slice partial sums, combined in
 * a fixed order (deterministic).  Forward: when `stats_in` ([channels][stats_slices][2] partial (sum, sum of squared
 * deviations from the partial's own mean) of x over EQUAL parts, e.g. from sgmcmc_conv3x3) is given, the statistics
 * pass over x is skipped.
 *
 * GROUPS (round 4).  `groups` = G >= 1 independent minibatches of n / G images each, stored one after the other, in ONE
 * launch: the exact full-data gradient (inference_reject.py:18-33) evaluates its minibatches several per launch chain,
 * and training-mode statistics are per minibatch (models/google_resnet.py:14-27).  Every group is normalised with its
 * own batch statistics and gives the bits of a launch on that group alone:
 *   - `n`, `stats_slices`, `n_partials` count the WHOLE launch (multiples of G); a producer's image-major partials
 *     ([channels][slices][2] from the convolutions' epilogues) are read as [channels][G][slices / G][2];
 *   - save_mean / save_invstd: [G][channels];  dgb: [G][2][channels] = every group's (dgamma, dbeta) -- a caller whose
 *     parameters are shared by the groups sums the G rows (sgmcmc_wrw_reduce_many: n_slabs = G, numel = 2 channels);
 *   - stat_log: group g's row at stat_log + g * log_stride doubles;
 *   - running statistics cannot be advanced in place when G > 1 (the groups' updates are ordered): pass NULL or use
 *     the logging entry point; hipErrorInvalidValue otherwise.
 * gamma / beta are shared by the groups. */
/* Evaluation mode (model.eval(): the per-epoch posterior-predictive evaluation of inference.py:199-213): one pass with
 * the running statistics, y = relu?(gamma (x - running_mean) / sqrt(running_var + eps) + beta [+ residual]). */
int sgmcmc_bn_eval_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                       const float* running_mean, const float* running_var, double eps, int relu, int n,
                       int channels, int plane, float* y, void* stream);
int64_t sgmcmc_bn_scratch_doubles(int n, int channels, int plane, int groups);
int sgmcmc_bn_train_fwd(const float* x, const float* residual, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, double momentum, double eps, int relu,
                        int n, int channels, int plane, float* y, float* save_mean, float* save_invstd,
                        double* scratch, const double* stats_in, int stats_slices, int groups, void* stream);
/* The same forward with the running statistics left ALONE: the batch mean and unbiased variance of every channel go to
 * stat_log[2 c + {0,1}] (doubles; group g: + g * log_stride) instead, and sgmcmc_bn_running_replay advances
 * running_mean / running_var by a sequence of such entries (log + j * entry_stride doubles, j = 0 .. n_entries - 1) in
 * order -- the same bits as n_entries forwards in that order.  For gradient passes whose minibatches are evaluated
 * concurrently -- on several streams and / or several per launch (the exact full-data pass, inference_reject.py:18-33)
 * -- while nn.BatchNorm2d's running statistics must still advance batch by batch. */
int sgmcmc_bn_train_fwd_log(const float* x, const float* residual, const float* gamma, const float* beta,
                            double* stat_log, int64_t log_stride, double eps, int relu, int n, int channels, int plane,
                            float* y, float* save_mean, float* save_invstd, double* scratch, const double* stats_in,
                            int stats_slices, int groups, void* stream);
int sgmcmc_bn_running_replay(const double* log, int64_t entry_stride, int n_entries, double momentum,
                             float* running_mean, float* running_var, int channels, void* stream);
/* ... for all BatchNorm layers of a network in ONE launch (any number: SGMCMC_BN_REPLAY_LAYERS per launch): layer i's
 * entries are layers[i].log + j * entry_stride (the layers' columns of one log array). */
#define SGMCMC_BN_REPLAY_LAYERS 32
typedef struct sgmcmc_bn_replay_layer {
  const double* log;
  float *running_mean, *running_var;
  double momentum;
  int32_t channels, reserved;
} sgmcmc_bn_replay_layer;
int sgmcmc_bn_running_replay_many(const sgmcmc_bn_replay_layer* layers, int n_layers, int64_t entry_stride,
                                  int n_entries, void* stream);
int sgmcmc_bn_train_bwd(const float* dy, const float* y, const float* x, const float* gamma,
                        const float* save_mean, const float* save_invstd, int relu, int n, int channels,
                        int plane, float* dx, float* dresidual, float* dgb, double* scratch, int groups, void* stream);
/* The first launch of sgmcmc_bn_train_bwd (relu = 1) on its own: per-slice partial sums of dz = dout * [out > 0] and
 * dz * xhat, sums [channels][*n_sums][2] doubles (sgmcmc_bn_scratch_doubles(...) of them; *n_sums counts all groups). */
int sgmcmc_bn_bwd_sums(const float* dout, const float* out, const float* y, const float* save_mean,
                       const float* save_invstd, double* sums, int* n_sums, int n, int channels, int plane,
                       int groups, void* stream);
/* The second launch of sgmcmc_bn_train_bwd alone, for partial sums that already exist: `partial` =
 * [channels][n_partials][2] doubles (sum dz, sum dz * xhat per slice) from sgmcmc_bn_bwd_sums or from the epilogue of
 * the convolution gradient that produced dy (sgmcmc_conv3x3_bwd_ex).
 * rs (may be NULL): the residual is itself the output of a BatchNorm WITHOUT ReLU (the down-sampling block's shortcut,
 * models/google_resnet.py:77-90) whose incoming gradient is dresidual = dz -- with relu: formed here and stored
 * (dresidual required); with relu = 0: dy arrives masked already (sgmcmc_conv_bwd_epilogue.mask_dx), the residual's
 * gradient IS dy and dresidual must be NULL; the launch
 * also leaves that BatchNorm's backward sums in rs->partial, [channels][sgmcmc_bn_scratch_doubles(...) / (2
 * channels)][2] doubles (the slices of this launch's own geometry); rs->mean / rs->invstd: [G][channels]. */
typedef struct {
  const float *y, *mean, *invstd; /* the residual BatchNorm's input and saved statistics */
  double* partial;
} sgmcmc_bn_residual_sums;
int sgmcmc_bn_bwd_dx(const float* dy, const float* y, const float* x, const float* gamma, const float* save_mean,
                     const float* save_invstd, int relu, int n, int channels, int plane, const double* partial,
                     int n_partials, float* dx, float* dresidual, float* dgb, const sgmcmc_bn_residual_sums* rs,
                     int groups, void* stream);
/* out = relu(BN(x) + BN_s(r)): a down-sampling block's last BatchNorm (models/google_resnet.py:77-90) with the 1x1
 * shortcut's BatchNorm -- no ReLU, this sum its only consumer -- applied on the fly: the shortcut BatchNorm's own
 * launch and its output tensor disappear; the same bits as sgmcmc_bn_train_fwd twice.  Training mode; both layers'
 * batch statistics as equal-part partial pairs [channels][slices][2] (a convolution epilogue's); save_* / running_* /
 * stat_log (+ log_stride) per layer as for sgmcmc_bn_train_fwd[_log]. */
typedef struct {
  const float* r;     /* the shortcut BatchNorm's input */
  const float* gamma;
  const float* beta;
  const double* partial;
  int n_partials, reserved;
  double eps, momentum;
  float *save_mean, *save_invstd, *running_mean, *running_var;
  double* stat_log;
  int64_t log_stride;
} sgmcmc_bn_dual;
int sgmcmc_bn_train_fwd_dual(const float* x, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, double momentum, double eps, int n, int channels, int plane, float* y,
                             float* save_mean, float* save_invstd, const double* stats_in, int stats_slices,
                             double* stat_log, int64_t log_stride, const sgmcmc_bn_dual* rs, int groups, void* stream);

/* y = maxpool2x2(relu(x + bias_c)), NCHW fp32, h and w even: the Conv2d(+bias) -> ReLU -> MaxPool2d(2) tail
 * of models/conv_nets.py:44-56 in one pass (the convolution itself is then run without its bias).
 * Backward: dx in full (the gradient goes to the first maximal element of each window, as ATen, if its
 * pre-activation is positive); dbias_part (may be NULL): [sgmcmc_pool_slices(...)][channels] partial sums
 * for sgmcmc_wrw_reduce_many.  `bias` may be NULL (= 0). */
int sgmcmc_pool_slices(int n, int channels, int h, int w);
int sgmcmc_bias_relu_pool_fwd(const float* x, const float* bias, float* y, int n, int channels, int h, int w,
                              void* stream);
int sgmcmc_bias_relu_pool_bwd(const float* x, const float* bias, const float* dy, float* dx,
                              float* dbias_part, int n, int channels, int h, int w, void* stream);

/* The ResNet's classifier head (google_resnet.py:103-110: AvgPool2d over the whole map -> Flatten -> Linear):
 *   pooled[n,c] = mean_p h[n,c,p];  logits[n,k] = bias[k] + sum_c weight[k,c] pooled[n,c]
 * and its gradients: dh[n,c,p] = (sum_k dlogits[n,k] weight[k,c]) / plane; slab_w [n][classes*channels] and
 * slab_b [n][classes] (either may be NULL) are per-row terms of dweight / dbias for sgmcmc_wrw_reduce_many
 * (n slabs).  channels <= 64, plane a multiple of 16, classes <= 16; bias may be NULL. */
int sgmcmc_pool_linear_fwd(const float* h, const float* weight, const float* bias, float* pooled,
                           float* logits, int n, int channels, int plane, int classes, void* stream);
int sgmcmc_pool_linear_bwd(const float* dlogits, const float* pooled, const float* weight, float* dh,
                           float* slab_w, float* slab_b, int n, int channels, int plane, int classes,
                           void* stream);
/* The head, the softmax cross-entropy on its logits and both backward passes in ONE launch (round 3): everything
 * between the trunk's last activation h and its gradient dh is per image.  Outputs: pooled / logits as
 * sgmcmc_pool_linear_fwd, loss_rows[n] = -log softmax(logits_n)[y_n], dlogits = grad_scale * (softmax - onehot) -- the
 * bits of sgmcmc_softmax_xent_fwd_grad -- and dh / slab_w / slab_b (/ partial, with the four bn_* pointers: plane == 64)
 * as sgmcmc_pool_linear_bwd[_sums] would produce from that dlogits.  counters (may be NULL): n_counters <= 256 int64
 * values that get + 1 (the BatchNorm layers' num_batches_tracked, kept in one array).  Replaces the head's Linear,
 * Categorical(logits).log_prob and their autograd backward inside R1 (inference.py:215-223, models/base.py:168-191). */
int sgmcmc_pool_linear_loss(const float* h, const float* weight, const float* bias, const int64_t* y, float* pooled,
                            float* logits, float* dlogits, float* loss_rows, float* dh, float* slab_w, float* slab_b,
                            const float* bn_y, const float* bn_out, const float* bn_mean, const float* bn_invstd,
                            double* partial, int group_rows, int64_t* counters, int n_counters, int n, int channels,
                            int plane, int classes, float grad_scale, void* stream);
/* ... when h is the output of a BatchNorm + ReLU (bn_y its input, bn_out = h, saved mean / invstd) on 8x8 maps
 * (plane == 64): the launch also leaves that BatchNorm's backward sums, partial[(c * n + image) * 2 + {0,1}] doubles
 * (n slices per channel) for sgmcmc_bn_bwd_dx -- the last BatchNorm of models/google_resnet.py:103-110's trunk.
 * group_rows > 0: that BatchNorm ran on groups of group_rows rows (sgmcmc_bn_train_fwd, GROUPS): bn_mean / bn_invstd are
 * [n / group_rows][channels]; 0: one batch. */
int sgmcmc_pool_linear_bwd_sums(const float* dlogits, const float* pooled, const float* weight, float* dh,
                                float* slab_w, float* slab_b, const float* bn_y, const float* bn_out,
                                const float* bn_mean, const float* bn_invstd, double* partial, int group_rows, int n,
                                int channels, int plane, int classes, void* stream);

/* A narrow linear layer, y = x W^T + b with out_features <= 16 (the convolutional classifier's head,
 * Flatten -> Linear(2450, 10), models/conv_nets.py:57-70): one launch each way, fixed-order reductions.
 * Backward, one launch: dx (may be NULL), dbias (may be NULL) and the weight gradient as
 * sgmcmc_linear_row_groups(n) partial slabs [group][out_features][in_features] (may be NULL), one per group of 16
 * batch rows, for sgmcmc_wrw_reduce_many (fixed order). */
int sgmcmc_linear_fwd(const float* x, const float* weight, const float* bias, float* y, int n, int in_features,
                      int out_features, void* stream);
/* sgmcmc_linear_fwd and, in the same launch, the rows' softmax cross-entropy on the logits it produced: dlogits =
 * grad_scale * (softmax(y_n) - onehot(y_labels[n])) and loss_rows[n] = -log softmax(y_n)[y_labels[n]] -- the bits of
 * sgmcmc_softmax_xent_fwd_grad (models/conv_nets.py:57-70 + models/base.py:168-191 in one launch). */
int sgmcmc_linear_fwd_loss(const float* x, const float* weight, const float* bias, const int64_t* y_labels, float* y,
                           float* dlogits, float* loss_rows, int n, int in_features, int out_features,
                           float grad_scale, void* stream);
int sgmcmc_linear_row_groups(int n);
int sgmcmc_linear_bwd(const float* x, const float* weight, const float* dy, float* dx, float* dweight_slabs,
                      float* dbias, int n, int in_features, int out_features, void* stream);

/* Minibatch gather from an HBM-resident image set with random crop (zero padding `pad`) and horizontal flip
 * applied on the way -- the `cifar10_augmented` pipeline (data/CIFAR/cifar.py:136-172: RandomCrop(32,
 * padding=4), RandomHorizontalFlip) without leaving the device:
 *   out[b,c,y,x] = data[idx[b], c, y + dy - pad, x' + dx - pad] (fill[c] outside; fill == NULL: 0),
 *   x' = flip_b ? W-1-x : x,
 *   (dx, dy, flip_b) = Philox4x32-10(seed; counter (idx[b], draw, purpose 3, stream)) -> r0 % (2 pad + 1),
 *   r1 % (2 pad + 1), r2 & 1 (only if `flip`).  `idx`: int64 device array of data-set rows; `draw`: the
 *   caller's per-pass counter.  pad = 0 and flip = 0 is a plain gather.  `fill` (device, [channels], or NULL):
 *   the value of the padding -- the reference pads the RAW image with black before normalising, so on
 *   normalised data the padding of channel c is (0 - mean_c) / std_c. */
int sgmcmc_augment_gather(const float* data, const int64_t* idx, float* out, const float* fill, int batch,
                          int channels, int height, int width, int pad, int flip, uint64_t seed,
                          uint32_t stream, uint64_t draw, void* stream_);

/* Round 4: everything between two replays of a captured step in ONE launch -- sgmcmc_augment_gather into `out` (the
 * graph's static input; pad = 0, flip = 0, channels = height = 1: a plain row gather), the labels gathered beside it
 * (labels_out[b] = labels[idx[b]]; both NULL: none), up to three plain copies as sgmcmc_stage_batch (the argument block
 * from its pinned slot; n_copies may be 0) and the previous transition's deferred bookkeeping (A_pending, as
 * sgmcmc_stage_batch).  Replaces the DataLoader hand-over of inference.py:197-205 + data/CIFAR/cifar.py:136-172 for a
 * captured step: 1 launch instead of 3 (gather, label index_select, staging copy). */
typedef struct sgmcmc_gather {
  const float* data;          /* [rows][channels][height][width] */
  const int64_t* labels;      /* [rows] or NULL */
  const int64_t* idx;         /* [batch] data-set rows */
  float* out;                 /* [batch][channels][height][width] */
  int64_t* labels_out;        /* [batch] or NULL */
  const float* fill;          /* [channels] or NULL */
  int32_t batch, channels, height, width, pad, flip;
  uint64_t seed, draw;
  uint32_t stream, reserved;
} sgmcmc_gather;
int sgmcmc_gather_stage(const sgmcmc_gather* G, const void* const* src, void* const* dst, const int64_t* bytes,
                        int n_copies, const sgmcmc_layout* L, const sgmcmc_step_args* A_pending, void* stream);

/* Up to three plain device copies in ONE launch: dst[j][0 .. bytes[j]) = src[j][...].  For the per-step staging of a
 * captured step -- the minibatch (x, y) into the graph's static inputs and the argument block from PINNED host
 * memory (the kernel reads it over the bus) -- which otherwise costs three copy dispatches between two graph
 * launches.  Pointers 16-byte aligned, sizes multiples of 4; else hipErrorInvalidValue.  No reference counterpart:
 * the reference's batches arrive from its DataLoader (experiments/train_bnn.py:172-180).
 * A_pending (may be NULL; else with L): the previous transition was launched with SGMCMC_DEFER_FINALIZE
 * (SMALL_FINALIZE layouts) and its per-segment bookkeeping is executed by extra workgroups of THIS launch -- the
 * same results as sgmcmc_finalize(L, A_pending) at this point of the stream. */
int sgmcmc_stage_batch(const void* const* src, void* const* dst, const int64_t* bytes, int n_copies,
                       const sgmcmc_layout* L, const sgmcmc_step_args* A_pending, void* stream);

/* loss = scale * sum_b -log softmax(logits_b)[y_b] for up to 1024 rows of up to 16 classes (the likelihood term of
 * models/base.py:168-191), one launch each way: forward keeps probs [rows][classes] for the backward,
 * dlogits = *grad_out * scale * (probs - onehot(y)).  scale = 1/rows for the minibatch mean, 1/N in the exact
 * full-data pass. */
int sgmcmc_softmax_xent_fwd(const float* logits, const int64_t* y, float* probs, float* loss, int rows,
                            int classes, double scale, void* stream);
int sgmcmc_softmax_xent_bwd(const float* probs, const int64_t* y, const float* grad_out, float* dlogits,
                            int rows, int classes, double scale, void* stream);
/* The same loss AND its gradient in one launch, for a caller that seeds autograd itself (logits.backward(dlogits)):
 * dlogits[b,k] = grad_scale * (softmax(logits_b)[k] - [k == y_b]) -- bit-identical to sgmcmc_softmax_xent_bwd on the
 * saved probabilities with grad_out[0] * scale == grad_scale. */
int sgmcmc_softmax_xent_fwd_grad(const float* logits, const int64_t* y, float* loss, float* dlogits, int rows,
                                 int classes, double scale, float grad_scale, void* stream);

/* Test hook: out[i] = spec normal (fp32) of noise index start+i. */
int sgmcmc_debug_normals(float* out, int64_t start, int64_t n, uint64_t seed, uint32_t stream,
                         uint64_t draw, uint32_t purpose, void* stream_);

#ifdef __cplusplus
}
#endif
#endif /* SGMCMC_HIP_H */
