"""Input side of the hot path: the reference's CPU workers normalise, rescale, crop and flip every
decoded clip in fp32 and ship 154 MB per 32-clip batch over PCIe (lib/datasets/howto100m.py:437-452,
lib/datasets/utils.py:110-160,309-326, lib/datasets/transform.py:8-191).  Here the host only DRAWS the random numbers
-- same numpy calls in the same order as the reference, so a seeded run picks the same crops -- and the decoded
uint8 frames (38 MB per batch) go to the GPU, where `pvrl_frames_u8_patchify` does normalise + bilinear short-side
rescale + crop + flip fused into the patch-embed im2col.
"""
import math

import numpy as np
import torch


def spatial_sampling_params(height, width, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224,
                            random_horizontal_flip=True, inverse_uniform_sampling=False):
    """The draws of utils.spatial_sampling (utils.py:110-160) for one clip of `height` x `width` frames.
    -> (new_h, new_w, y_off, x_off, flip).  Uses np.random exactly like transform.py:31-37,103-108,142."""
    assert spatial_idx in [-1, 0, 1, 2]
    # random_short_side_scale_jitter (transform.py:8-61)
    if inverse_uniform_sampling and spatial_idx == -1:
        size = int(round(1.0 / np.random.uniform(1.0 / max_scale, 1.0 / min_scale)))
    else:
        size = int(round(np.random.uniform(min_scale, max_scale)))
    new_h, new_w = height, width
    if not ((width <= height and width == size) or (height <= width and height == size)):
        new_w = new_h = size
        if width < height:
            new_h = int(math.floor((float(height) / width) * size))
        else:
            new_w = int(math.floor((float(width) / height) * size))
    if spatial_idx == -1:
        # random_crop (transform.py:84-118)
        y_off = x_off = 0
        if not (new_h == crop_size and new_w == crop_size):
            if new_h > crop_size:
                y_off = int(np.random.randint(0, new_h - crop_size))
            if new_w > crop_size:
                x_off = int(np.random.randint(0, new_w - crop_size))
        flip = 0
        if random_horizontal_flip:
            flip = int(np.random.uniform() < 0.5)      # horizontal_flip(0.5, ...) (transform.py:121-147)
    else:
        # uniform_crop (transform.py:150-191)
        y_off = int(math.ceil((new_h - crop_size) / 2))
        x_off = int(math.ceil((new_w - crop_size) / 2))
        if new_h > new_w:
            if spatial_idx == 0:
                y_off = 0
            elif spatial_idx == 2:
                y_off = new_h - crop_size
        else:
            if spatial_idx == 0:
                x_off = 0
            elif spatial_idx == 2:
                x_off = new_w - crop_size
        flip = 0
    if new_h < crop_size or new_w < crop_size:
        raise ValueError(f"rescaled frame {new_h}x{new_w} is smaller than the crop {crop_size}")
    return new_h, new_w, y_off, x_off, flip


class DecodedClips:
    """A batch of decoded clips waiting for the fused GPU input kernel: `frames` uint8 [B, T, H0, W0, 3] (decoder
    order, on the GPU), `params` int32 [B, 5] = (new_h, new_w, y_off, x_off, flip) per clip.  Quacks like the fp32
    tensor [B, 3, T, crop, crop] the reference's loader would have produced (`.shape`, `.device`, `.is_cuda`)."""

    def __init__(self, frames, params, mean, std, crop_size):
        assert frames.dtype == torch.uint8 and frames.dim() == 5 and frames.shape[-1] == 3
        self.frames = frames.contiguous()
        p = torch.as_tensor(params, dtype=torch.int32).reshape(-1, 5)
        assert p.shape[0] == frames.shape[0]
        self.params_host = p.cpu()
        self.params = p.to(frames.device)
        self.mean = [float(v) for v in mean]
        self.std = [float(v) for v in std]
        self.crop = int(crop_size)

    @property
    def shape(self):
        B, T = self.frames.shape[:2]
        return torch.Size((B, 3, T, self.crop, self.crop))

    @property
    def device(self):
        return self.frames.device

    @property
    def is_cuda(self):
        return self.frames.is_cuda

    def contiguous(self):
        return self

    def float(self):
        return self
