"""ctypes binding of the C ABI in include/sgmcmc_hip.h (libsgmcmc_hip.so).

There is deliberately NO fallback: if the HIP library is missing or cannot be
loaded, every sampler constructor raises.  ``import torch`` happens first so the
library's ``libamdhip64.so.7`` dependency resolves to the HIP runtime torch has
already mapped (same SONAME) -- one runtime per process, streams and device
pointers are shared with torch.
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SGMCMC_ALTERNATIVES=1: the library that ALSO carries the measured alternatives (include/sgmcmc_hip_alternatives.h:
# folded BatchNorm, fused BatchNorm backward, side-stream weight gradients) -- a separate file,
# so the shipped library never contains them; the switches that select them (resblock.FOLD_BN, conv.SIDE_STREAM, ...)
# refuse to turn on without it.
ALTERNATIVES = os.environ.get("SGMCMC_ALTERNATIVES", "0") == "1"
LIB_PATH = os.path.join(_HERE, "_build", "libsgmcmc_hip_alt.so" if ALTERNATIVES else "libsgmcmc_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")
# one translation unit: sgmcmc_hip.hip #includes mlp_hip.inc (they share the finalize body)
SOURCES = [os.path.join(_HERE, "csrc", "sgmcmc_hip.hip")]
SOURCE = SOURCES[0]

ABI_VERSION = 6
CHUNK = 4096
CHUNK_SMALL = 1024
NSUMS = 6
PSTRIDE = 8
MLP_ROWS = 16
F32, F64 = 0, 1
VERLET, HMC, SGLD = 0, 1, 2
INITIAL, FINAL, SAVE_STATE, CALC_METRICS, UNALIGNED, NO_MOMENTUM, SMALL_FINALIZE = 1, 2, 4, 8, 16, 32, 64
WITH_LOG_PRIOR = 128
DEFER_FINALIZE = 256
INLINE_PRIOR = 512
PRIOR_NONE, PRIOR_NORMAL, PRIOR_LAPLACE, PRIOR_STUDENT_T, PRIOR_CAUCHY, PRIOR_GENNORM = 0, 1, 2, 3, 4, 5
PRIOR_GAMMA_SOFTPLUS, PRIOR_UNIFORM_CDF, PRIOR_HALFCAUCHY_SOFTPLUS, PRIOR_IMPROPER_SOFTPLUS = 6, 7, 8, 9
PRIOR_HAS_LINKS, PRIOR_FULL = 1, 2

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-shared"]

# numpy mirrors of the device-resident tables
SEGMENT_DTYPE = np.dtype([("theta", "<u8"), ("g", "<u8"), ("M", "<f8"), ("numel", "<i8"),
                          ("first_chunk", "<i8"), ("noise_base", "<i8"), ("prior_kind", "<i4"),
                          ("scale_link", "<i4"), ("prior_loc", "<f8"), ("prior_scale", "<f8"),
                          ("prior_df", "<f8")], align=True)
CHUNK_DTYPE = np.dtype([("seg", "<i4"), ("n_valid", "<i4")], align=True)
SEG_STATE_FIELDS = ("sum_gg", "sum_gmo", "sum_gmn", "sum_momo", "sum_mnmn", "sum_thg",
                    "delta_energy", "prev_delta", "est_temperature", "est_config_temp",
                    "point_energy", "aux")
assert SEGMENT_DTYPE.itemsize == 80 and CHUNK_DTYPE.itemsize == 8


class Layout(ctypes.Structure):
    _fields_ = [("dtype", ctypes.c_int32), ("n_seg", ctypes.c_int32), ("n_chunks", ctypes.c_int64),
                ("chunk_elems", ctypes.c_int64), ("segs", ctypes.c_void_p), ("chunks", ctypes.c_void_p),
                ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("prev_theta", ctypes.c_void_p),
                ("prev_g", ctypes.c_void_p), ("prev_m", ctypes.c_void_p),
                ("partials", ctypes.c_void_p), ("state", ctypes.c_void_p),
                ("scalars", ctypes.c_void_p), ("prior_flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class StepArgs(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("flags", ctypes.c_uint32),
                ("seg_begin", ctypes.c_int32), ("seg_end", ctypes.c_int32),
                ("chunk_begin", ctypes.c_int64), ("chunk_end", ctypes.c_int64),
                ("num_data", ctypes.c_double), ("b2h2", ctypes.c_double), ("bh", ctypes.c_double),
                ("bhn", ctypes.c_double), ("mom_decay", ctypes.c_double),
                ("grad_v", ctypes.c_double), ("noise_std", ctypes.c_double),
                ("rmsprop_alpha", ctypes.c_double), ("grad_clamp", ctypes.c_double),
                ("seed", ctypes.c_uint64), ("draw", ctypes.c_uint64),
                ("stream", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class FragJob(ctypes.Structure):
    "sgmcmc_frag_job"
    _fields_ = [("w", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("dgrad", ctypes.c_void_p),
                ("channels", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ReduceJob(ctypes.Structure):
    "sgmcmc_reduce_job"
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n_slabs", ctypes.c_int32),
                ("numel", ctypes.c_int32), ("taps", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class ConvBwdEpilogue(ctypes.Structure):
    "sgmcmc_conv_bwd_epilogue"
    _fields_ = ([(n, ctypes.c_void_p) for n in ("e_dout", "e_out", "s_y", "s_out", "s_mean", "s_invstd", "s_partial")]
                + [("group_imgs", ctypes.c_int32), ("wrw_mult", ctypes.c_int32), ("mask_dx", ctypes.c_int32),
                   ("reserved", ctypes.c_int32)])


class BnResidualSums(ctypes.Structure):
    "sgmcmc_bn_residual_sums"
    _fields_ = [(n, ctypes.c_void_p) for n in ("y", "mean", "invstd", "partial")]


class BnDual(ctypes.Structure):
    "sgmcmc_bn_dual"
    _fields_ = ([(n, ctypes.c_void_p) for n in ("r", "gamma", "beta", "partial")]
                + [("n_partials", ctypes.c_int32), ("reserved", ctypes.c_int32), ("eps", ctypes.c_double),
                   ("momentum", ctypes.c_double)]
                + [(n, ctypes.c_void_p) for n in ("save_mean", "save_invstd", "running_mean", "running_var", "stat_log")]
                + [("log_stride", ctypes.c_int64)])


class Gather(ctypes.Structure):
    "sgmcmc_gather"
    _fields_ = ([(n, ctypes.c_void_p) for n in ("data", "labels", "idx", "out", "labels_out", "fill")]
                + [(n, ctypes.c_int32) for n in ("batch", "channels", "height", "width", "pad", "flip")]
                + [("seed", ctypes.c_uint64), ("draw", ctypes.c_uint64), ("stream", ctypes.c_uint32),
                   ("reserved", ctypes.c_uint32)])


class BnReplayLayer(ctypes.Structure):
    "sgmcmc_bn_replay_layer"
    _fields_ = [("log", ctypes.c_void_p), ("running_mean", ctypes.c_void_p), ("running_var", ctypes.c_void_p),
                ("momentum", ctypes.c_double), ("channels", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class BnIn(ctypes.Structure):
    "sgmcmc_bn_in"
    _fields_ = ([(n, ctypes.c_void_p) for n in ("fx", "gamma", "beta", "save_mean", "save_invstd", "running_mean",
                                                "running_var", "stat_log")]
                + [("momentum", ctypes.c_double), ("eps", ctypes.c_double), ("h", ctypes.c_void_p)])


class ConvBnBwdArgs(ctypes.Structure):
    "sgmcmc_conv_bn_bwd_args"
    _fields_ = ([(n, ctypes.c_void_p) for n in ("dout", "mask_out", "y", "mean", "invstd", "gamma", "sums")]
                + [("n_sums", ctypes.c_int32), ("reserved", ctypes.c_int32)]
                + [(n, ctypes.c_void_p) for n in ("dgamma", "dbeta", "e_dout", "e_out")])


class GradParts(ctypes.Structure):
    _fields_ = [("gpart", ctypes.c_void_p), ("loss_part", ctypes.c_void_p),
                ("correct_part", ctypes.c_void_p), ("stride", ctypes.c_int64),
                ("num_data", ctypes.c_double), ("n_slices", ctypes.c_int32),
                ("batch", ctypes.c_int32)]


class MlpArgs(ctypes.Structure):
    _fields_ = [("X", ctypes.c_void_p), ("Y", ctypes.c_void_p), ("idx", ctypes.c_void_p),
                ("W1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("W2", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("W3", ctypes.c_void_p), ("b3", ctypes.c_void_p),
                ("gpart", ctypes.c_void_p), ("loss_part", ctypes.c_void_p),
                ("correct_part", ctypes.c_void_p), ("gpart_stride", ctypes.c_int64),
                ("off_W1", ctypes.c_int64), ("off_b1", ctypes.c_int64), ("off_W2", ctypes.c_int64),
                ("off_b2", ctypes.c_int64), ("off_W3", ctypes.c_int64), ("off_b3", ctypes.c_int64),
                ("batch", ctypes.c_int32), ("in_features", ctypes.c_int32),
                ("hidden1", ctypes.c_int32), ("hidden2", ctypes.c_int32),
                ("out_features", ctypes.c_int32), ("inv_softmax_temp", ctypes.c_float),
                ("trace", ctypes.c_void_p), ("args_src", ctypes.c_void_p),
                ("args_dst", ctypes.c_void_p), ("args_bytes", ctypes.c_int32),
                ("grad_scale", ctypes.c_float), ("split_scratch", ctypes.c_void_p)]


class DenseChain(ctypes.Structure):
    "sgmcmc_dense_chain"
    _fields_ = [("mlp", MlpArgs), ("layout", Layout), ("num_data", ctypes.c_double),
                ("chain_id", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


MAX_CHAINS, MLP_BATCH_MULTI = 8, 128

EXPORTS = {
    "sgmcmc_abi_version": (ctypes.c_int, []),
    "sgmcmc_source_sha": (ctypes.c_char_p, []),
    "sgmcmc_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "sgmcmc_step": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(StepArgs), ctypes.c_void_p]),
    "sgmcmc_step_timed": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(StepArgs),
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_step_indirect": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(StepArgs),
                                            ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_step_indirect_parts": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(StepArgs),
                                                  ctypes.c_void_p, ctypes.POINTER(GradParts),
                                                  ctypes.c_void_p]),
    "sgmcmc_event_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p)]),
    "sgmcmc_event_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "sgmcmc_event_record": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_time_next_launch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_event_elapsed_ms": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.POINTER(ctypes.c_float)]),
    "sgmcmc_sample_momentum": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_double, ctypes.c_double,
                                              ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64,
                                              ctypes.c_void_p]),
    "sgmcmc_restore": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_int, ctypes.c_uint32,
                                      ctypes.c_void_p]),
    "sgmcmc_delta_energy": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_int, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_uint32,
                                           ctypes.c_void_p]),
    "sgmcmc_segment_sum": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_int, ctypes.c_uint32,
                                          ctypes.c_void_p]),
    "sgmcmc_prior_grad": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_double, ctypes.c_int,
                                         ctypes.c_uint32, ctypes.c_void_p]),
    "sgmcmc_mlp_fwdbwd": (ctypes.c_int, [ctypes.POINTER(MlpArgs), ctypes.c_void_p]),
    "sgmcmc_mlp_lds_bytes": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_mlp_split_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_grad_reduce_prior": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                                ctypes.c_int, ctypes.c_double, ctypes.c_uint32,
                                                ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_dense_stepper_create": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(MlpArgs),
                                                   ctypes.POINTER(StepArgs), ctypes.c_double,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                   ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]),
    "sgmcmc_dense_stepper_step": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(StepArgs),
                                                 ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_dense_stepper_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "sgmcmc_dense_step_direct": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(MlpArgs),
                                                ctypes.POINTER(StepArgs), ctypes.c_double,
                                                ctypes.c_void_p, ctypes.POINTER(StepArgs),
                                                ctypes.c_void_p]),
    "sgmcmc_dense_step_multi": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(DenseChain), ctypes.c_int,
                                               ctypes.POINTER(StepArgs), ctypes.c_void_p, ctypes.POINTER(StepArgs),
                                               ctypes.c_void_p]),
    "sgmcmc_finalize": (ctypes.c_int, [ctypes.POINTER(Layout), ctypes.POINTER(StepArgs),
                                       ctypes.c_void_p]),
    "sgmcmc_accumulate_parts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv3x3_stat_slices": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sgmcmc_conv3x3": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.c_void_p]),
    "sgmcmc_conv3x3_wrw_scratch_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sgmcmc_conv3x3_wrw": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv3x3_bn_eval": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_void_p, ctypes.c_int,
                                              ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "sgmcmc_bn_scratch_doubles": (ctypes.c_int64, [ctypes.c_int] * 4),
    "sgmcmc_bn_train_fwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_double, ctypes.c_double]
                            + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_bn_train_fwd_dual": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_double, ctypes.c_double]
                                 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
                                 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(BnDual), ctypes.c_int,
                                    ctypes.c_void_p]),
    "sgmcmc_bn_train_fwd_log": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_double] + [ctypes.c_int] * 4
                                + [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_bn_running_replay": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_bn_running_replay_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                                     ctypes.c_void_p]),
    "sgmcmc_bn_train_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4
                            + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv3x3_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3
                           + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_wrw_reduce_many": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv_down_stat_slices": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sgmcmc_conv_down_fwd": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "sgmcmc_conv_down_scratch_floats": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sgmcmc_conv_down_bwd": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_int] * 3
                             + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv_stem_fwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv_stem_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv_stem_wrw": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                                     ctypes.c_void_p]),
    "sgmcmc_pool_slices": (ctypes.c_int, [ctypes.c_int] * 4),
    "sgmcmc_bias_relu_pool_fwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "sgmcmc_bias_relu_pool_bwd": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "sgmcmc_linear_fwd_loss": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_void_p]),
    "sgmcmc_pool_linear_loss": (ctypes.c_int, [ctypes.c_void_p] * 16 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5
                                + [ctypes.c_float, ctypes.c_void_p]),
    "sgmcmc_pool_linear_fwd": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "sgmcmc_pool_linear_bwd_sums": (ctypes.c_int, [ctypes.c_void_p] * 11 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "sgmcmc_pool_linear_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "sgmcmc_linear_fwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "sgmcmc_linear_row_groups": (ctypes.c_int, [ctypes.c_int]),
    "sgmcmc_linear_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
    "sgmcmc_augment_gather": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
                              + [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p]),
    "sgmcmc_softmax_xent_fwd_grad": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                    ctypes.c_float, ctypes.c_void_p]),
    "sgmcmc_conv3x3_bwd_add": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int] * 3
                               + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv3x3_bwd_ex": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.POINTER(ConvBwdEpilogue)]
                              + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv_down_bwd_sum_slices": (ctypes.c_int, [ctypes.c_int] * 3),
    "sgmcmc_bn_eval_fwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_double] + [ctypes.c_int] * 4
                           + [ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_conv_down_bwd_ex": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.POINTER(ConvBwdEpilogue)]
                                + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_bn_bwd_dx": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
                         + [ctypes.c_void_p] * 3 + [ctypes.POINTER(BnResidualSums), ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_gather_stage": (ctypes.c_int, [ctypes.POINTER(Gather)] + [ctypes.c_void_p] * 3
                            + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_stage_batch": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p]),
    "sgmcmc_bn_bwd_sums": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.POINTER(ctypes.c_int)] + [ctypes.c_int] * 4
                           + [ctypes.c_void_p]),
    "sgmcmc_conv50": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv50_fwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv50_bwd_t": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                                   ctypes.c_void_p]),
    "sgmcmc_conv50_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv50_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                                 ctypes.c_void_p]),
    "sgmcmc_conv_first_fwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv_first_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv_first_wrw": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                                                      ctypes.c_void_p]),
    "sgmcmc_conv50_pool_fwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv50_pool_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv50_pool_bwd": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int, ctypes.c_int,
                                                                       ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv_first_pool_fwd": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv_first_pool_scratch_floats": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv_first_pool_bwd": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int, ctypes.c_int,
                                                                           ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_softmax_xent_fwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                                        ctypes.c_void_p]),
    "sgmcmc_softmax_xent_bwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                                                        ctypes.c_void_p]),
    "sgmcmc_debug_normals": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64,
                                            ctypes.c_uint32, ctypes.c_void_p]),
    "sgmcmc_conv3x3_prepare_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "sgmcmc_conv3x3_frag_stat_slices": (ctypes.c_int, [ctypes.c_int] * 3),
    "sgmcmc_conv3x3_frag_scratch_floats": (ctypes.c_int64, [ctypes.c_int] * 3),
    "sgmcmc_conv3x3_frag_fwd": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2),
    "sgmcmc_conv3x3_frag_bwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.POINTER(ConvBwdEpilogue)]
                                + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
}

# include/sgmcmc_hip_alternatives.h (libsgmcmc_hip_alt.so only)
ALT_EXPORTS = {
    "sgmcmc_fx_slot_int64": (ctypes.c_int64, [ctypes.c_int]),
    "sgmcmc_conv3x3_fx": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]),
    "sgmcmc_conv3x3_bnin": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3
                            + [ctypes.c_void_p, ctypes.POINTER(BnIn), ctypes.c_void_p]),
    "sgmcmc_conv3x3_bwd_part": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.POINTER(ConvBwdEpilogue), ctypes.c_void_p]
                                + [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv3x3_bn_bwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.POINTER(ConvBnBwdArgs)]
                              + [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]),
    "sgmcmc_conv3x3_bwd_uniform_slabs": (ctypes.c_int, [ctypes.c_int] * 3),
    "sgmcmc_conv3x3_bwd_uniform": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.POINTER(ConvBwdEpilogue), ctypes.c_void_p]
                                   + [ctypes.c_int] * 3 + [ctypes.c_void_p]),
}
if ALTERNATIVES:
    EXPORTS.update(ALT_EXPORTS)

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def lib():
    """Load libsgmcmc_hip.so (built by ``__graft_entry__.build()``); never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipExtensionMissing(
                f"{LIB_PATH} not found: build the HIP extension first "
                f"(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise HipExtensionMissing(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in EXPORTS.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.sgmcmc_abi_version() != ABI_VERSION:
            raise HipExtensionMissing(f"ABI mismatch: library {L.sgmcmc_abi_version()} != {ABI_VERSION}")
        # the binary must be the one these sources compile to: a library left over from before an edit of csrc/ would
        # otherwise run -- and be measured -- under the tree's name.  A variant build (tools/build_variant.sh) carries
        # "<sha>+<flags>"; SGMCMC_ALLOW_STALE_LIB=1 turns the refusal into a warning (bisecting with an old binary).
        built_from, tree = L.sgmcmc_source_sha().decode(), source_sha()
        if built_from.split("+")[0] != tree:
            msg = (f"{LIB_PATH} was built from sources {built_from!r}, the tree is {tree!r}: rebuild it "
                   f"(python -c 'import __graft_entry__ as g; g.build()')")
            if os.environ.get("SGMCMC_ALLOW_STALE_LIB", "0") != "1":
                raise HipExtensionMissing(msg)
            import warnings
            warnings.warn(msg)
        _lib = L
    return _lib


def library_sha():
    "what the LOADED library says it was built from: the stamp of every measurement (bench.py, tools/step_summary.py)"
    return lib().sgmcmc_source_sha().decode()


def check(err, what):
    if err != 0:
        raise RuntimeError(f"{what} failed: hipError {err} ({lib().sgmcmc_error_string(err).decode()})")


def source_sha():
    """Hash of every kernel / ABI source (csrc/*, include/*): what a committed measurement of the kernels
    (profiles/in_step_us.json, profiles/pmc_traffic.json) records as ``source_sha`` and bench.py compares with,
    so that figures measured on other kernels than the tree's are marked stale instead of being mixed in."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(_HERE, "csrc"), INCLUDE_DIR):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".inc", ".h", ".hpp")):
                h.update(name.encode())
                with open(os.path.join(d, name), "rb") as f:
                    h.update(f.read())
    return h.hexdigest()[:16]


def build(verbose=False):
    """hipcc cross-compile for gfx950 (works without a GPU)."""
    import subprocess
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    cmd = ["hipcc", *HIPCC_FLAGS, *(["-DSGMCMC_ALTERNATIVES"] if ALTERNATIVES else []),
           f'-DSGMCMC_SOURCE_SHA="{source_sha()}"', "-I", INCLUDE_DIR,
           "-I", os.path.join(_HERE, "csrc"), *SOURCES, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH
