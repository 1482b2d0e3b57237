"""Fused optimizer step + EMA for the training loop around the hot path.

The reference steps with ``optim.SGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)`` over the two
weight-decay groups of ``basics/optimizer.py:35-49`` (Train.py:139-150), calls ``scaler.step(optimizer)`` every
``accumulate`` batches (Train.py:448-450) and then ``ema.update(model)`` - a Python loop over the 273 state_dict tensors
(basics/utils/torch_utils.py:291-301).  Here the engine keeps parameters, gradients, momentum and the EMA in flat f32
buffers of one layout, so all of it - plus the cast of the updated masters to the bf16 copy the GEMM kernels read - is
ONE streaming kernel (csrc/optim.hip, ``sodt_sgd_ema_step``).

``FusedSGD`` is a ``torch.optim.Optimizer``: param groups, ``lr`` / ``momentum`` / ``weight_decay`` per group (the
warm-up of Train.py:375-385 writes them every iteration), LR schedulers and ``zero_grad`` behave as with torch's SGD.
``ModelEMA`` mirrors the reference class (``.ema``, ``.updates``, ``.decay``, ``update``, ``update_attr``); attached to
the optimizer (``FusedSGD(..., ema=ema)``) its parameter average rides in the fused kernel and ``ema.update(model)``
only handles the few non-parameter buffers.
"""
from __future__ import annotations

import math
from copy import deepcopy
from typing import Optional

import torch

from . import ops


def set_weight_decay(model, skip_list=(), skip_keywords=(), weight_decay: float = 0.00048):
    """basics/optimizer.py:35-49: 1-D parameters and biases are not decayed."""
    has_decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.dim() == 1 or name.endswith(".bias") or name in skip_list or any(k in name for k in skip_keywords):
            no_decay.append(p)
        else:
            has_decay.append(p)
    return [{"params": has_decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]


class ModelEMA:
    """basics/utils/torch_utils.py:271-301 with the parameter average kept in one flat f32 buffer (the EMA module's
    parameters are views of it, so ``ema.ema`` is an ordinary Model for evaluation and checkpoints)."""

    def __init__(self, model, decay: float = 0.9999, updates: int = 0):
        model._get_engine()                 # (parameters of `model` live in its engine's flat buffer from here on)
        self.ema = deepcopy(model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        # the copy's own engine re-homes ITS parameters into a flat buffer of the same layout: that buffer is the average
        self.flat = self.ema._get_engine().flat_param
        self._fused_pending = False        # set by FusedSGD.step when the parameter average was done in the fused kernel

    def next_decay(self) -> float:
        return self.decay(self.updates + 1)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            msd = model.state_dict()
            if self._fused_pending:          # parameters already averaged with this d by the optimizer's kernel
                self._fused_pending = False
                names = [k for k, _ in self.ema.named_buffers()]
            else:
                names = list(self.ema.state_dict().keys())
            esd = self.ema.state_dict()
            dst = [esd[k] for k in names if esd[k].dtype.is_floating_point]
            src = [msd[k].detach() for k in names if esd[k].dtype.is_floating_point]
            if dst:
                torch._foreach_mul_(dst, d)
                torch._foreach_add_(dst, src, alpha=1.0 - d)
            if self.ema._engine is not None:
                self.ema._engine.invalidate_params()

    def update_attr(self, model, include=(), exclude=("process_group", "reducer")):
        for k, v in model.__dict__.items():
            if (len(include) and k not in include) or k.startswith("_") or k in exclude:
                continue
            setattr(self.ema, k, v)


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum, nesterov, weight_decay) semantics (dampening 0) over the engine's flat buffers."""

    def __init__(self, params, model, lr: float = 0.01, momentum: float = 0.937, weight_decay: float = 0.0,
                 nesterov: bool = True, ema: Optional[ModelEMA] = None):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov))
        if len(self.param_groups) > 4:
            raise ValueError("at most 4 parameter groups (sodt_sgd_ema_step)")
        self.model, self.ema = model, ema
        self._eng = None
        self._mom = None
        self._groups = None
        self._sig = None

    def _bind(self):
        eng = self.model._get_engine()
        if eng is not self._eng:
            self._eng, self._mom, self._sig = eng, torch.zeros_like(eng.flat_param), None
        sig = tuple(tuple(id(p) for p in g["params"]) for g in self.param_groups)
        if sig != self._sig:       # chunk -> group map (255: padding / parameters this optimizer does not own)
            gmap = torch.full((eng.flat_param.numel() // 4,), 255, dtype=torch.uint8)
            off_of = {id(eng.params[n]): (eng.grad_offsets[n], eng.params[n].numel()) for n in eng.grad_order}
            for gi, g in enumerate(self.param_groups):
                for p in g["params"]:
                    if id(p) not in off_of:
                        raise ValueError("FusedSGD: a parameter does not belong to the model's engine")
                    o, n = off_of[id(p)]
                    gmap[o // 4: (o + n + 3) // 4] = gi
            self._groups, self._sig = gmap.to(eng.dev), sig
        return eng

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        eng = self._bind()
        eng._check_param_views()
        if eng._claim_grads():              # no backward since zero_grad(set_to_none=True): torch skips parameters without a gradient
            eng.flat_grad.zero_()
            return loss
        gs = self.param_groups
        nest = {bool(g["nesterov"]) for g in gs}
        if len(nest) != 1:
            raise ValueError("FusedSGD: nesterov must be the same for every group")
        ema_flat, d = None, 0.0
        if self.ema is not None:
            ema_eng = self.ema.ema._get_engine()
            if self.ema.flat is not ema_eng.flat_param:       # the EMA model's engine was rebuilt (fuse(), unpickled / assigned model)
                self.ema.flat = ema_eng.flat_param
            ema_eng._check_param_views()
            ema_flat, d = self.ema.flat, self.ema.next_decay()
            self.ema._fused_pending = True
        cast = next(iter(eng.flat_cast.values())) if eng.flat_cast else None
        ops.sgd_ema_step(eng.flat_param, eng.flat_grad, self._mom, ema_flat, cast, self._groups,
                         [g["lr"] for g in gs], [g["momentum"] for g in gs], [g["weight_decay"] for g in gs],
                         nest.pop(), grad_scale, d)
        if cast is not None:
            eng.mark_cast_fresh()
        else:
            eng.invalidate_params()         # (no mirror to keep fresh; the engine still has to know the masters moved: _param_epoch)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd["momentum_flat"] = None if self._mom is None else self._mom.clone()
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)                       # the caller's dict keeps its "momentum_flat" entry
        mom = sd.pop("momentum_flat", None)
        super().load_state_dict(sd)
        if mom is not None:
            self._bind()
            self._mom.copy_(mom.to(self._mom.device))
