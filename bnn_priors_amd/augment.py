"""Random crop + horizontal flip applied ON THE DEVICE while a minibatch is gathered from the HBM-resident
training set (``csrc/augment_hip.inc``; C ABI ``sgmcmc_augment_gather``) -- the data side of the
reference's ``cifar10_augmented`` (bnn_priors/data/CIFAR/cifar.py:136-172: ``RandomCrop(32, padding=4)``,
``RandomHorizontalFlip`` on DataLoader workers).

    ds = AugmentedTensorDataset(x_train, y_train, RandomCropFlip(pad=4, flip=True, seed=1234, stream=rank))
    loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True)
    runner = VerletSGLDRunnerReject(model, loader, ...)       # gathers + augments on the GPU

A sample's crop offset and flip are a function of (seed, stream, its row in the data set, pass counter):
every traversal of the set -- leapfrog epochs and exact-gradient passes alike, as in the reference, whose
loader re-draws on every access -- sees new augmentations, reproducibly and independently of batch
composition.  The random stream is this package's Philox specification, not torchvision's.
"""
import torch

from . import _hip

__all__ = ("RandomCropFlip", "AugmentedTensorDataset")


class RandomCropFlip:
    def __init__(self, pad=4, flip=True, seed=0, stream=0, fill=None):
        """``fill``: per-channel value of the padding (sequence of C floats), default 0.  The reference pads
        the raw image with black BEFORE normalising (cifar.py:158-163), so for a normalised data set pass
        ``fill = -mean / std`` to reproduce its border exactly."""
        if pad < 0:
            raise ValueError("pad must be >= 0")
        self.pad, self.flip, self.seed, self.stream = int(pad), bool(flip), int(seed), int(stream)
        self.fill = None if fill is None else torch.as_tensor(fill, dtype=torch.float32).reshape(-1)

    def gather(self, data, idx, draw, out=None):
        """data [N, C, H, W] float32 on the GPU, idx int64 [B] on the same device -> augmented [B, C, H, W]
        (``out``: a contiguous float32 [B, C, H, W] tensor to write instead of a new one -- e.g. a slice of a captured
        graph's static input: no staging copy)"""
        if not data.is_cuda or data.dtype != torch.float32 or data.dim() != 4 or not data.is_contiguous():
            raise ValueError("augmentation needs a contiguous float32 [N, C, H, W] tensor on the GPU")
        idx = idx.to(device=data.device, dtype=torch.int64).contiguous()
        shape = (idx.numel(),) + tuple(data.shape[1:])
        if out is None:
            out = torch.empty(shape, dtype=torch.float32, device=data.device)
        elif (tuple(out.shape) != shape or out.dtype != torch.float32 or out.device != data.device
              or not out.is_contiguous()):
            raise ValueError("out must be a contiguous float32 tensor of the gathered batch's shape on the data's device")
        if idx.numel() == 0:
            return out
        fill = 0
        if self.fill is not None:
            if self.fill.numel() != data.shape[1]:
                raise ValueError("fill needs one value per channel")
            if self.fill.device != data.device:
                self.fill = self.fill.to(data.device)
            fill = self.fill.data_ptr()
        err = _hip.lib().sgmcmc_augment_gather(data.data_ptr(), idx.data_ptr(), out.data_ptr(), fill, idx.numel(),
                                               data.shape[1], data.shape[2], data.shape[3], self.pad,
                                               int(self.flip), self.seed & (2 ** 64 - 1), self.stream, int(draw),
                                               torch.cuda.current_stream().cuda_stream)
        if err:
            _hip.check(err, "sgmcmc_augment_gather")
        return out


class AugmentedTensorDataset(torch.utils.data.Dataset):
    """``TensorDataset(x, y)`` whose images are augmented when read.  The runners' batch source
    recognises ``.tensors`` / ``.augment`` and gathers whole minibatches with one kernel launch, bumping
    ``draw`` once per traversal; reading single items (a plain DataLoader) uses the current ``draw`` --
    call ``next_draw()`` between epochs then."""

    def __init__(self, x, y, augment):
        if len(x) != len(y):
            raise ValueError("x and y differ in length")
        self.tensors, self.augment, self.draw = (x, y), augment, 0

    def __len__(self):
        return len(self.tensors[0])

    def next_draw(self):
        self.draw += 1
        return self.draw

    def __getitem__(self, i):
        x, y = self.tensors
        idx = torch.tensor([i], dtype=torch.int64, device=x.device)
        return self.augment.gather(x, idx, self.draw)[0], y[i]
