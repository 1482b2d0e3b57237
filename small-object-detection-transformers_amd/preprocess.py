"""Input pre-processing of the reference's training / evaluation loop on the device.

Train.py:364-374 moves the uint8 batch to the GPU, converts it with ``.float() / 255.0`` and, when the images were loaded at
``train_img_size = down_factor x test_img_size`` (defaults 1024 / 512, Train.py:94,612-613), shrinks RGB and IR with
``F.interpolate(..., mode='bilinear', align_corners=True)``: four full-tensor ATen kernels and two f32 intermediates at the
loaded resolution.  ``preprocess_batch`` is that in ONE launch (csrc/preprocess.hip, ``sodt_preprocess_u8``): uint8 planes
in, f32 planes out, the layout ``Model.forward`` / the front-end kernel read.  test.py:124-129 is the ``down_factor=1`` case.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops


def preprocess_batch(imgs: torch.Tensor, irs: torch.Tensor, down_factor: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """imgs, irs: uint8 (B, C, H, W) on the GPU -> f32 (B, C, H // down_factor, W // down_factor) in [0, 1]."""
    if imgs.dtype != torch.uint8 or irs.dtype != torch.uint8:
        raise TypeError("preprocess_batch takes the uint8 batch of the data loader (Train.py:362)")
    if not imgs.is_cuda or not irs.is_cuda:
        raise RuntimeError("preprocess_batch needs the batch on the GPU (Train.py:364: imgs.to(device)); there is no CPU fallback")
    if imgs.dim() != 4 or irs.dim() != 4 or imgs.shape[0] != irs.shape[0] or imgs.shape[2:] != irs.shape[2:]:
        raise ValueError("imgs and irs must be (B, C, H, W) with the same batch and size")
    if down_factor < 1:
        raise ValueError("down_factor >= 1 (Train.py:94: int(train_img_size / test_img_size))")
    imgs, irs = imgs.contiguous(), irs.contiguous()
    B, c1, H, W = imgs.shape
    c2 = irs.shape[1]
    Ho, Wo = H // down_factor, W // down_factor
    out1 = torch.empty(B, c1, Ho, Wo, device=imgs.device, dtype=torch.float32)
    out2 = torch.empty(B, c2, Ho, Wo, device=imgs.device, dtype=torch.float32)
    ops._launch("sodt_preprocess_u8", imgs.data_ptr(), irs.data_ptr(), out1.data_ptr(), out2.data_ptr(), B, c1, c2, H, W, Ho, Wo)
    return out1, out2
