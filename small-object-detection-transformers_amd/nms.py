"""non_max_suppression on the GPU - the drop-in for `utils/general.py:425` (called from
test.py:145 on the eval output of `Model.forward`).

Same signature and return value as the reference: a list with one (n, 6) f32 tensor
[x1, y1, x2, y2, conf, cls] per image, on the prediction's device.  Everything after the
Detect decode - candidate selection, class-offset NMS, the 300-detection cut, merge-NMS and
the redundancy filter - runs in HIP kernels (csrc/nms.hip) behind the C ABI
(`sodt_nms_candidates`, `sodt_nms_select`); the host only reads two counters per image.

Differences from the reference, all documented in DESIGN.md:
  * `labels` (autolabelling rows, general.py:451-458): each image's (nl, 5) [cls, x, y, w, h] rows are appended to its
    prediction rows with obj = 1 and a one-hot class before the device kernels run (candidate row index N + k).
  * ties between equal scores are broken by candidate order (a stable sort), where
    torchvision.ops.nms / argsort leave the order unspecified.
  * the 10 s time limit (general.py:509-511) does not exist.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops

MAX_DET = 300


def non_max_suppression(prediction: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        classes: Optional[Sequence[int]] = None, agnostic: bool = False, multi_label: bool = False,
                        labels=(), return_index: bool = False) -> List[torch.Tensor]:
    if not prediction.is_cuda:
        raise RuntimeError("non_max_suppression: the prediction must live on the GPU (there is no CPU fallback)")
    if prediction.dim() != 3 or prediction.shape[2] < 6:
        raise ValueError(f"prediction must be (B, N, 5 + nc), got {tuple(prediction.shape)}")
    pred = prediction.detach()
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        pred = pred.float().contiguous()
    B, N, no = pred.shape
    nc = no - 5
    dev = pred.device
    ml = bool(multi_label) and nc > 1
    allow = None
    if classes is not None:
        allow = torch.zeros(nc, dtype=torch.uint8, device=dev)
        idx = torch.as_tensor([int(c) for c in classes if 0 <= int(c) < nc], dtype=torch.long, device=dev)
        allow[idx] = 1
    nlab = [len(l) for l in labels] if labels else []
    if nlab and len(nlab) != B:
        raise ValueError("labels: one (nl, 5) tensor per image")
    Nx = N + (max(nlab) if nlab else 0)
    cap = Nx * nc if ml else Nx
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    counters = torch.zeros(2, dtype=torch.int32, device=dev)
    out_rows = torch.empty(MAX_DET, 6, dtype=torch.float32, device=dev)
    out_idx = torch.empty(MAX_DET, dtype=torch.int32, device=dev)
    output = [torch.zeros((0, 6), device=dev)] * B
    index = [torch.zeros((0,), dtype=torch.long, device=dev)] * B
    with torch.cuda.device(dev):
        for b in range(B):
            pb = pred[b]
            if nlab and nlab[b]:                          # general.py:451-458
                l = labels[b].to(device=dev, dtype=torch.float32)
                v = torch.zeros((nlab[b], no), device=dev)
                v[:, :4] = l[:, 1:5]
                v[:, 4] = 1.0
                v[torch.arange(nlab[b], device=dev), l[:, 0].long() + 5] = 1.0
                pb = torch.cat((pb, v), 0).contiguous()
            ops.nms_candidates(pb, conf_thres, ml, allow, keys, counters[0:1])
            n = int(counters[0].item())
            if n == 0:
                continue
            ws = torch.empty(ops.nms_workspace_bytes(n), dtype=torch.uint8, device=dev)
            ops.nms_select(pb, keys, n, iou_thres, agnostic, ws, out_rows, out_idx, counters[1:2])
            m = int(counters[1].item())
            output[b] = out_rows[:m].clone()
            index[b] = out_idx[:m].long()
    return (output, index) if return_index else output
