"""How a BatchNorm backward gets its channel sums from the convolution gradient that produces its incoming gradient.

Forward: an operator whose output is ``out = relu(bn(y) [+ r])`` tags the tensor it returns with where it came
from (``tag_output``: the BatchNorm's input y and its saved mean / invstd).  An operator whose data gradient has the
SUMS epilogue (``sgmcmc_conv3x3_bwd_ex``: the trunk's 3x3 convolutions) looks the tag up on ITS input
(``source_of``) and keeps it for its backward.

Backward: that operator's data-gradient launch also leaves, per channel and band, the partial sums of
``dz = dx * [out > 0]`` and ``dz * xhat`` -- exactly what the producing BatchNorm's backward needs -- and tags the
gradient tensor it returns (``tag_gradient``).  The BatchNorm's backward finds them on its incoming gradient
(``sums_of``) and skips its own sums launch (5 us, 21 per googleresnet step before this).

The tag is only trusted while the gradient tensor is untouched: autograd adds gradients of a tensor with several
consumers either out of place (a new tensor: no tag) or in place (the version counter moves: ``sums_of`` rejects it).
Nothing here launches anything; reference semantics: BatchNorm2d's backward reductions inside autograd's pass over
models/google_resnet.py:34-43 (inference.py:215-223)."""
import os

ENABLED = True          # (module attribute, not an environment switch)
# Round 4: a launch that leaves the sums also STORES the gradient masked, dz = dx * [out > 0] (``mask_dx`` of
# sgmcmc_conv_bwd_epilogue: it holds the mask for the sums anyway).  The gradient of a BatchNorm + ReLU output has two
# kinds of consumers -- the BatchNorm's dx launch and the shortcut add of the residual block before -- and both form
# exactly dz from it, so with the tag's ``masked`` flag neither reads ``out`` again (a quarter of the dx launch's
# traffic; same bits).  Where the tag is lost (a gradient autograd accumulated) the consumers mask as before, which is
# correct for a masked operand too: (dz + other) * [out > 0] = (dx + other) * [out > 0].  What changes: the gradient
# autograd holds for such an activation is dz, not dx -- they differ only where the activation is exactly 0, where
# the ReLU's own backward zeroes it.  SGMCMC_PREMASK=0 stores dx as before.
PREMASK = ENABLED and os.environ.get("SGMCMC_PREMASK", "1") != "0"
STATS = {"upstream": 0, "own": 0}        # BatchNorm backward passes that found their sums / launched their own


def tag_output(out, bn_input, saved):
    "out = relu(bn(bn_input) [+ r]), saved = [2][C] (mean, invstd)"
    if ENABLED:
        out._sgmcmc_bn_src = (bn_input.detach(), saved)
    return out


def source_of(x):
    "(bn_input, saved) if x is a tagged BatchNorm + ReLU output of x's own shape, else (None, None)"
    src = getattr(x, "_sgmcmc_bn_src", None) if ENABLED else None
    if src is None or src[0].shape != x.shape or not src[0].is_contiguous():
        return None, None
    return src


def tag_linear_output(out, bn_input, saved):
    "out = bn(bn_input) without ReLU (a down-sampling block's shortcut): a BatchNorm that takes it as residual can sum for it"
    if ENABLED:
        out._sgmcmc_bn_src_lin = (bn_input.detach(), saved)
    return out


def linear_source_of(r):
    src = getattr(r, "_sgmcmc_bn_src_lin", None) if (ENABLED and r is not None) else None
    if src is None or src[0].shape != r.shape or not src[0].is_contiguous():
        return None, None
    return src


def tag_gradient(dx, partial, n_partials, masked=False):
    "``masked``: dx was stored as dz = dx * [out > 0] (PREMASK)"
    dx._sgmcmc_bn_sums = (partial, n_partials, dx._version, dx.data_ptr(), bool(masked))
    return dx


def sums_of(dout):
    "(partial, n_partials, masked) left by the launch that produced dout, or None"
    tok = getattr(dout, "_sgmcmc_bn_sums", None) if ENABLED else None
    if tok is None or tok[2] != dout._version or tok[3] != dout.data_ptr() or not dout.is_contiguous():
        STATS["own"] += 1
        return None
    STATS["upstream"] += 1
    return tok[0], tok[1], tok[4]
