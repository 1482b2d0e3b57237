"""ComputeLoss of the reference's training step (basics/utils/loss.py:90-224) on the device: build_targets, CIoU box loss,
objectness and class BCE and their gradient with respect to the head output in four small launches (csrc/loss.hip,
``sodt_yolo_loss``) - no autograd graph over ~60 ATen kernels, no host synchronisation.  Same constructor, call
signature and return tuple as the reference class, so ``compute_loss = ComputeLoss(model)`` /
``loss, lbox, lobj, lcls = compute_loss(pred, targets)`` (Train.py:281,418) port unchanged.

Not carried over: focal loss (``fl_gamma > 0``) and ``autobalance`` (both off in models/hyp.scratch.yaml and
Train.py:281) and label smoothing other than the reference's hard-coded ``smooth_BCE(eps=0.0)`` (loss.py:104); asking
for them raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import ops


DEFAULT_HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0)   # models/hyp.scratch.yaml


def synthetic_targets(B: int, per_image: int = 32, nc: int = 8, seed: int = 0) -> torch.Tensor:
    """Benchmark targets: per image `per_image` boxes, class ~U{0..nc-1}, centre ~U(0.05, 0.95),
    size ~U(0.01, 0.05); rows (image, class, x, y, w, h) as Train.py:362 delivers them."""
    g = torch.Generator().manual_seed(seed)
    n = B * per_image
    img = torch.arange(B).repeat_interleave(per_image).float()
    cls = torch.randint(0, nc, (n,), generator=g).float()
    xy = 0.05 + 0.9 * torch.rand(n, 2, generator=g)
    wh = 0.01 + 0.04 * torch.rand(n, 2, generator=g)
    return torch.cat((img[:, None], cls[:, None], xy, wh), 1)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, targets, anchors, hyp, gr, nc):
        B, na, ny, nx, no = pred.shape
        nt = int(targets.shape[0])
        dpred = torch.empty_like(pred)
        out = torch.empty(4, device=pred.device, dtype=torch.float32)
        nbytes = C.c_size_t(0)
        lib = L.load()
        if lib.sodt_yolo_loss_workspace_bytes(B * na * ny * nx, nt, nc, C.byref(nbytes)) != 0:
            raise RuntimeError("sodt_yolo_loss_workspace_bytes: unsupported shape (nc <= 32)")
        ws = torch.empty(nbytes.value, device=pred.device, dtype=torch.uint8)
        ops._launch("sodt_yolo_loss", pred.data_ptr(), targets.data_ptr() if nt else None, nt, anchors.data_ptr(), B, na, ny, nx, nc,
                    C.c_float(hyp["box"]), C.c_float(hyp["cls"]), C.c_float(hyp["cls_pw"]), C.c_float(hyp["obj"]),
                    C.c_float(hyp["obj_pw"]), C.c_float(hyp["anchor_t"]), C.c_float(gr), ws.data_ptr(), nbytes.value,
                    dpred.data_ptr(), out.data_ptr())
        ctx.save_for_backward(dpred)
        # independent tensors, not slices of `out`: views of one buffer returned by a multi-output Function are
        # MULTI_OUTPUT_NODE views on which the reference loop's in-place `loss *= opt.world_size` (Train.py:440),
        # `loss *= 4.` and `loss += sr_loss` raise
        loss, lbox, lobj, lcls = (out[i:i + 1].clone() for i in range(4))
        ctx.mark_non_differentiable(lbox, lobj, lcls)
        return loss, lbox, lobj, lcls

    @staticmethod
    def backward(ctx, g_loss, g_box, g_obj, g_cls):
        (dpred,) = ctx.saved_tensors
        # only the total is differentiated by the training loop (Train.py:445); the three components are reporting values
        return dpred * g_loss.to(dpred.dtype).reshape(()), None, None, None, None, None


class ComputeLoss:
    def __init__(self, model, autobalance: bool = False):
        if autobalance:
            raise NotImplementedError("autobalance is off in the reference's training loop (Train.py:281)")
        h = model.hyp
        if h.get("fl_gamma", 0.0) > 0:
            raise NotImplementedError("focal loss (fl_gamma > 0) is not built; models/hyp.scratch.yaml uses 0")
        det = model.module.detect[-1] if hasattr(model, "module") else model.detect[-1]
        if det.nl != 1:
            raise NotImplementedError("one detection layer (models/model.yaml)")
        self.hyp, self.gr, self.autobalance = h, model.gr, False
        self.cp, self.cn = 1.0, 0.0                 # smooth_BCE(eps=0.0), loss.py:104
        self.balance = [4.0, 1.0, 0.25, 0.06, .02]  # loss.py:110 for nl == 1
        for k in ("na", "nc", "nl", "anchors"):
            setattr(self, k, getattr(det, k))

    def __call__(self, p, targets):
        pred = p[0] if isinstance(p, (list, tuple)) else p
        if not pred.is_cuda:
            raise RuntimeError("ComputeLoss needs the head output on the GPU: there is no CPU fallback")
        if pred.dtype != torch.float32 or not pred.is_contiguous():
            pred = pred.float().contiguous()
        targets = targets.to(device=pred.device, dtype=torch.float32).contiguous()
        anchors = self.anchors[0].to(device=pred.device, dtype=torch.float32).contiguous()
        hyp = {k: float(self.hyp[k]) for k in ("box", "cls", "cls_pw", "obj", "obj_pw", "anchor_t")}
        return _LossFn.apply(pred, targets, anchors, hyp, float(self.gr), int(self.nc))
