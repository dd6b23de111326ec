"""Random crop + horizontal flip, oracle side (TEST INFRASTRUCTURE): a numpy restatement of the
specification in ``bnn_priors_amd/csrc/augment_hip.inc`` -- the on-device counterpart of the reference's
``cifar10_augmented`` transforms (bnn_priors/data/CIFAR/cifar.py:136-172: ``RandomCrop(32, padding=4)``,
``RandomHorizontalFlip``).  torchvision is not part of this image, so the reference's own random
stream cannot be replayed: parity with the reference is distributional (uniform 9 x 9 offsets, fair
flips -- checked in tests/test_augment.py); parity between this restatement and the HIP kernel is bit for
bit.
"""
import numpy as np

from .noise import philox4x32_10_np

PURPOSE_AUGMENT = 3


def decisions(rows, seed, stream, draw, pad, flip):
    "(dx, dy, flipped) per data-set row, int arrays"
    rows = np.asarray(rows, dtype=np.uint64)
    ctr = np.zeros(rows.shape + (4,), dtype=np.uint32)
    ctr[..., 0] = (rows & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    ctr[..., 1] = (rows >> np.uint64(32)).astype(np.uint32)
    ctr[..., 2] = np.uint32(draw & 0xFFFFFFFF)
    ctr[..., 3] = np.uint32((PURPOSE_AUGMENT << 28) | ((stream & 0xFFF) << 16) | ((draw >> 32) & 0xFFFF))
    r = philox4x32_10_np(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32))
    span = np.uint32(2 * pad + 1)
    dx = (r[..., 0] % span).astype(np.int64) - pad
    dy = (r[..., 1] % span).astype(np.int64) - pad
    fl = ((r[..., 2] & np.uint32(1)) == 1) if flip else np.zeros(rows.shape, dtype=bool)
    return dx, dy, fl


def gather(data, rows, seed, stream, draw, pad, flip, fill=None):
    "data [N, C, H, W] -> augmented batch [len(rows), C, H, W]; ``fill``: per-channel padding value (default 0)"
    data = np.asarray(data)
    n, c, h, w = data.shape
    dx, dy, fl = decisions(rows, seed, stream, draw, pad, flip)
    out = np.zeros((len(rows), c, h, w), dtype=data.dtype)
    ys, xs = np.arange(h)[:, None], np.arange(w)[None, :]
    for b, row in enumerate(rows):
        sx = (w - 1 - xs if fl[b] else xs) + dx[b]
        sy = ys + dy[b]
        ok = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
        src = data[int(row)][:, np.clip(sy, 0, h - 1), np.clip(sx, 0, w - 1)]
        out[b] = np.where(ok[None], src, 0 if fill is None else np.asarray(fill, dtype=data.dtype)[:, None, None])
    return out
