"""Execution engine: the whole forward AND backward of Model(model.yaml) as a fixed
sequence of HIP kernels (C ABI, include/sodt_hip.h), token-major activations.

* One ``Plan`` per (batch, resolution, dtype, mode): every activation / gradient buffer
  is allocated once; the first run records every launch, later runs replay the list.
* Backward is hand-written (no torch autograd inside): ``_EngineFn`` is a single
  autograd node whose backward runs the recorded reverse sequence and deposits parameter
  gradients into one flat f32 buffer (``param.grad`` are views of it), the layout a
  bucketed RCCL all-reduce wants (ddp.py).
* Reference lines each phase replaces are cited in the kernel sources and
  include/sodt_hip.h; the graph is backbone_vit.py:190-272 + models/model.yaml:65-74.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Dict, List, Optional, Tuple

import warnings

import torch

from . import _lib as L
from . import ops
from .ops import SegSpec

HEADS = 12
E = "image_encoder."
TAPS2 = ((0, 0), (0, 1), (1, 0), (1, 1))                       # (dy, dx) of the 2x2 conv, kernel index kh*2+kw
TAPS3 = tuple((dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1))
MERGE = ((0, 0), (1, 0), (0, 1), (1, 1))                        # PatchMerging gather order (backbone_vit.py:850-853)
USE_HIPGRAPH = os.environ.get("SODT_HIPGRAPH", "0") == "1"     # opt-in: see Engine._replay


class Plan:
    def __init__(self, B, S, dt, training, dev):
        self.B, self.S, self.dt, self.training, self.dev = B, S, dt, training, dev
        self.bufs: Dict[str, torch.Tensor] = {}
        self.bwd_marks: List[Tuple[int, int, int]] = []   # (index in bwd_main, first, last+1 element of flat_grad) of the
        #                                                   gradient buckets that are final at that launch (ddp overlap)
        self.gen = 0                              # bumped by every forward that overwrites the saved activations
        self.fwd_pre: Optional[list] = None
        self.fwd_main: Optional[list] = None
        self.bwd_main: Optional[list] = None
        self.saved: dict = {}
        self.graphs: dict = {}                    # hipGraph of a recorded list (captured on its first replay)

    def zbuf(self, pool, name, shape, dtype):
        """Small accumulator that must be zero at the start of every forward (pool "f") or backward (pool "b"): carved out of
        one 2 MiB pool per direction that a single memset clears (24 tiny memset launches per step otherwise)."""
        t = self.bufs.get(name)
        if t is None:
            pbuf = self.zpool(pool)
            n = 1
            for d in shape:
                n *= d
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            off = self.zused[pool]
            if off + nbytes > pbuf.numel():
                raise RuntimeError("zero pool exhausted")
            t = pbuf[off: off + nbytes].view(dtype).view(shape)
            self.zused[pool] = (off + nbytes + 255) // 256 * 256
            self.bufs[name] = t
        return t

    def zpool(self, pool):
        pbuf = self.bufs.get("zpool." + pool)
        if pbuf is None:
            pbuf = self.bufs["zpool." + pool] = torch.zeros(1 << 21, device=self.dev, dtype=torch.uint8)
            self.zused = getattr(self, "zused", {})
            self.zused[pool] = 0
        return pbuf

    def buf(self, name, shape, dtype=None, zero=False):
        t = self.bufs.get(name)
        if t is None:
            dtype = self.dt if dtype is None else dtype
            t = (torch.zeros if zero else torch.empty)(shape, device=self.dev, dtype=dtype)
            self.bufs[name] = t
        return t


class _LazyFeatures(list):
    """forward_once's feature list `y` (model.py:268-281).  Entries produced by nn.Upsample(x2, nearest) and Concat
    (models/model.yaml:66-67,71-72: y[4], y[5], y[8], y[9]) are index arithmetic inside the GEMM loaders and never exist in
    the workspace; they are computed on first access.  rules: {index: ("up", src) | ("cat", [srcs])} from the head plan."""

    def __init__(self, items=(), rules=None):
        super().__init__(items)
        self._rules = dict(rules or {})

    def _fill(self, i):
        if list.__getitem__(self, i) is None and i in self._rules:
            r = self._rules[i]
            if r[0] == "up":
                v = torch.nn.functional.interpolate(self[r[1]].float(), scale_factor=2, mode="nearest")
            else:
                v = torch.cat([self[j].float() for j in r[1]], 1)
            list.__setitem__(self, i, v)

    def __getitem__(self, i):
        if isinstance(i, int):
            j = i + len(self) if i < 0 else i
            self._fill(j)
        else:
            for j in sorted(self._rules):
                self._fill(j)
        return list.__getitem__(self, i)

    def __iter__(self):
        for j in sorted(self._rules):
            self._fill(j)
        return list.__iter__(self)

    def __add__(self, other):       # Model.forward appends the head output: stay lazy
        out = _LazyFeatures(list.__iter__(self), self._rules)
        list.extend(out, other)
        return out


class _EngineFn(torch.autograd.Function):
    """One autograd node for the whole model.  Parameter gradients are written straight into
    ``param.grad`` (views of the engine's flat buffer), so the node returns None for them."""

    @staticmethod
    def forward(ctx, engine, plan, x_rgb, x_ir, anchor):
        ctx.engine, ctx.plan = engine, plan
        ctx.inputs = (x_rgb, x_ir)
        ctx.set_materialize_grads(False)         # an unused output (detection-only or SR-only loss) arrives as None, not zeros
        pred = engine._forward(plan, x_rgb, x_ir)
        ctx.gen = plan.gen
        if engine.sr:
            return pred, engine._sr_forward(plan)
        return pred

    @staticmethod
    def backward(ctx, dpred, dsr=None):
        if ctx.gen != ctx.plan.gen:
            raise RuntimeError("the activations this backward needs were overwritten by a later forward of the same "
                               "(batch, resolution, dtype, mode): call backward() before the next forward, or run the "
                               "extra forward under model.eval() / torch.no_grad() with a different mode")
        ctx.engine._backward(ctx.plan, ctx.inputs[0], ctx.inputs[1], None if dpred is None else dpred.contiguous().float(),
                             None if dsr is None else dsr.contiguous().float())
        return None, None, None, None, None


class Engine:
    def __init__(self, model):
        self.model = model
        self.params: Dict[str, torch.nn.Parameter] = dict(model.named_parameters())
        self.buffers: Dict[str, torch.Tensor] = dict(model.named_buffers())
        p0 = next(iter(self.params.values()))
        if not p0.is_cuda:
            raise RuntimeError("move the model to the GPU first: the engine has no CPU path")
        for n, p in self.params.items():
            if p.dtype != torch.float32:
                raise RuntimeError(f"master parameters must be float32 (got {p.dtype} for {n}); call model.float()")
        self.dev = p0.device
        enc = model.image_encoder
        self.img_t = enc.pos_embed.shape[1]
        det = model.detect[-1]
        self.na, self.no, self.nc = det.na, det.no, det.nc
        self.det_np = (det.na * det.no + 15) // 16 * 16      # Detect GEMM width, zero-padded to a multiple of 16 (39 -> 48 at nc = 8)
        self.fused = not any(type(m).__name__ == "BatchNorm2d" for m in model.detect.modules())
        self.sr = bool(getattr(model, "sr", False))
        self._check_head()
        self.plans: Dict[Tuple, Plan] = {}
        self.probes_fwd: Optional[dict] = None    # bench.py: {call index: (start event, end event)}
        self.probes_bwd: Optional[dict] = None
        self.prep: Dict[torch.dtype, dict] = {}
        self._build_grad_buffer()
        self._build_param_buffer()
        # anchor that makes the autograd node require grad even if the caller froze everything else
        self._anchor = torch.zeros(1, device=self.dev, requires_grad=True)
        self.ddp = None     # set by ddp.attach()
        self.use_fused_wmsa = True     # tests / tools may switch the fused block kernel off to compare with the four launches it replaces
        self.use_fused_mlp = True      # the same for the fused linear MLP (csrc/mlp.hip) against its two GEMM launches
        # 2x2-conv MLPs (bf16): fc1 folded into the convolution's weights (csrc/convmlp.hip) - no fc1 GEMM, and in the backward no
        # d(x) = du W1 and no dW1 GEMM; widths above this run the three-GEMM form (the composition kernels are plain f32 loops)
        self.convmlp_fold_maxc = 384
        self.use_direct_conv3 = True   # the head's 3x3 Conv at 64 -> 64 channels on the direct kernels (csrc/conv3.hip) against the nine-segment GEMM
        # 64 MiB of f32 for the per-slice partial tiles of the bf16 weight-gradient GEMMs (largest need: 12.5 M floats)
        self._tn_scratch = torch.empty(16 << 20, dtype=torch.float32, device=self.dev)
        ops.set_tn_scratch(self._tn_scratch)

    # ------------------------------------------------------------------ head graph
    def _check_head(self):
        """Parse ``model.detect`` (parse_model's nn.Sequential with the reference's ``.f`` wiring, model.py:268-281) into a
        launch plan.  Accepted: any CHAIN of compute units - Conv (1x1 or 3x3, stride 1), C3 (n = 1), SPP - in which every
        unit reads ONE feature-list entry, possibly through nn.Upsample(x2, nearest) and Concat rows (their inputs: the
        previous rows or encoder outputs y[0..2]), ending in a one-layer Detect; every unit output and every encoder output
        is consumed exactly once.  That covers models/model.yaml:65-74, the identical head of SRyolo_MF.yaml:52-71, and
        variants with extra Conv / SPP (common.py:129-140) rows.  Upsample and Concat never move data: they become the
        K-segments (source + spatial map) of the consuming unit's first GEMM."""
        from . import model as M
        d = self.model.detect
        enc_c = (256, 256, 512)                                   # neck widths (backbone_vit.py:167-187), levels 0, 1, 2
        # virtual tensor: (parts, level) with parts = [(ref, C, shr)], ref = ("enc", j) | ("unit", k); resolution t >> level
        vals = {j: ([(("enc", j), enc_c[j], 0)], j) for j in range(3)}
        units, rules = [], {}
        uses: Dict[tuple, int] = {}

        def resolve(f, yi):
            j = yi - 1 if f == -1 else f
            if not isinstance(j, int) or j not in vals or j >= yi:
                raise NotImplementedError(f"head row {yi - 3}: input {f!r} does not name an earlier feature-list entry")
            return j
        if not isinstance(d[-1], M.Detect) or d[-1].nl != 1:
            raise NotImplementedError("the head must end in a one-layer Detect (models/model.yaml:74)")
        for k, m in enumerate(d):
            yi, kind = 3 + k, type(m).__name__
            if kind in ("Conv", "C3", "SPP"):
                if not isinstance(m.f, int):
                    raise NotImplementedError(f"head row {k}: {kind} takes one input")
                parts, level = vals[resolve(m.f, yi)]
                c1 = sum(c for _, c, _ in parts)
                if kind == "Conv":
                    kk, c2 = m.conv.kernel_size[0], m.conv.out_channels
                    if m.conv.in_channels != c1:
                        raise NotImplementedError(f"head row {k}: Conv expects {m.conv.in_channels} channels, graph gives {c1}")
                    if kk == 3 and len(parts) * 9 > L.MAX_SEG:
                        raise NotImplementedError(f"head row {k}: a 3x3 Conv on a concatenation needs {len(parts) * 9} K-segments (max {L.MAX_SEG})")
                elif kind == "C3":
                    kk, c2 = 1, m.cv3.conv.out_channels
                    if len(m.m) != 1 or m.m[0].add:
                        raise NotImplementedError("C3 with n=1, shortcut=False only (models/model.yaml:68,73)")
                    if m.cv1.conv.in_channels != c1:
                        raise NotImplementedError(f"head row {k}: C3 expects {m.cv1.conv.in_channels} channels, graph gives {c1}")
                else:
                    kk, c2 = 1, m.cv2.conv.out_channels
                    if m.cv1.conv.in_channels != c1:
                        raise NotImplementedError(f"head row {k}: SPP expects {m.cv1.conv.in_channels} channels, graph gives {c1}")
                for ref, _, _ in parts:
                    uses[ref] = uses.get(ref, 0) + 1
                units.append(dict(k=k, kind=kind, parts=parts, level=level, c1=c1, c2=c2, ksize=kk))
                vals[yi] = ([(("unit", k), c2, 0)], level)
            elif kind == "Upsample":
                parts, level = vals[resolve(m.f, yi)]
                if level == 0:
                    raise NotImplementedError(f"head row {k}: Upsample above the stride-4 grid of Detect (model.py:130)")
                vals[yi] = ([(ref, c, shr + 1) for ref, c, shr in parts], level - 1)
                rules[yi] = ("up", resolve(m.f, yi))
            elif kind == "Concat":
                srcs = [resolve(f, yi) for f in m.f]
                if len({vals[j][1] for j in srcs}) != 1:
                    raise NotImplementedError(f"head row {k}: Concat of different resolutions")
                vals[yi] = (sum((vals[j][0] for j in srcs), []), vals[srcs[0]][1])
                rules[yi] = ("cat", srcs)
            elif kind == "Detect":
                if k != len(d) - 1 or len(m.f) != 1:
                    raise NotImplementedError("Detect must be the last row with one input")
                parts, level = vals[resolve(m.f[0], yi)]
                if len(parts) != 1 or parts[0][0][0] != "unit" or parts[0][2] != 0 or level != 0:
                    raise NotImplementedError("Detect reads one unit output on the stride-4 grid (model.py:130: stride = [4.])")
                uses[parts[0][0]] = uses.get(parts[0][0], 0) + 1
                self.head_out = (parts[0][0][1], parts[0][1])          # (unit row, channels)
            else:
                raise NotImplementedError(f"head row {k}: module {kind} is outside the hot path (SURVEY.md section 8)")
        for ref in [("enc", j) for j in range(3)] + [("unit", u["k"]) for u in units]:
            if uses.get(ref, 0) != 1:
                raise NotImplementedError(f"head: {ref} is consumed {uses.get(ref, 0)} times; the hand-written backward routes every "
                                          "feature to exactly one consumer")
        self.head_units, self.head_rules, self.nd = units, rules, len(d) - 1
        self.det_name = f"detect.{self.nd}."
        self.sr_taps = None
        if self.sr:
            # model_up(y[l1], y[l2]) (model.py:286) cannot run with the yaml's l1 / l2 = 4 / 8 (256 channels into the 128-channel
            # conv1); the taps are the first feature-list entries with the channel counts and grids DeepLab(ch, c1, c2) needs:
            # c1 on the stride-4 grid (low-level) and c2 on the stride-8 grid - y[8] and y[5] in models/model.yaml
            mu = self.model.model_up
            def first(cn, level):
                for yi in sorted(vals):
                    if vals[yi][1] == level and sum(c for _, c, _ in vals[yi][0]) == cn:
                        return yi
                raise NotImplementedError(f"sr=True: no feature-list entry has {cn} channels on the stride-{4 << level} grid")
            def fits(yi, cn, level):
                return yi in vals and vals[yi][1] == level and sum(c for _, c, _ in vals[yi][0]) == cn
            l1, l2 = getattr(self.model, "l1", None), getattr(self.model, "l2", None)
            if fits(l1, mu.c1, 0) and fits(l2, mu.c2, 1):
                self.sr_taps = (l1, l2)                      # the yaml's own taps (model.py:286: model_up(y[l1], y[l2])) are usable
            else:
                self.sr_taps = (first(mu.c1, 0), first(mu.c2, 1))
                warnings.warn(f"sr=True: y[l1={l1}] / y[l2={l2}] of the yaml do not have {mu.c1} channels on the stride-4 grid / {mu.c2} on the "
                              f"stride-8 grid that DeepLab(c1, c2) takes; tapping y[{self.sr_taps[0]}] / y[{self.sr_taps[1]}] instead "
                              "(DESIGN.md section 4.3: graph parity unpinned)")
            self.sr_parts = (vals[self.sr_taps[0]][0], vals[self.sr_taps[1]][0])

    def _unit_out_name(self, u):
        return f"h{u['k']}" + {"Conv": ".y", "C3": ".cv3.y", "SPP": ".cv2.y"}[u["kind"]]

    # ------------------------------------------------------------------ gradients: one flat f32 buffer
    def _build_grad_buffer(self):
        order = [E + f"channel_embed_{c}.proj.weight" for c in "rgbi"] + [E + f"channel_embed_{c}.proj.bias" for c in "rgbi"]
        order += [E + f"chan_block.norm{i}.weight" for i in range(1, 5)] + [E + f"chan_block.norm{i}.bias" for i in range(1, 5)]
        # the rest in REVERSE order of completion in the backward, so that the buckets that become final first are
        # contiguous slices at the end of the buffer (ddp.py): [front end | pos/patch embed | stage 1 | PatchMerging 1,
        # neck 1 | stage 2 | PatchMerging 2, neck 2 | stage 3 | neck 3 | head]
        groups = [E + "pos_embed", E + "patch_embed.", E + "stage1.", E + "pmerging1.", E + "neck1.", E + "stage2.",
                  E + "pmerging2.", E + "neck2.", E + "stage3.", E + "neck3.", "detect.", "model_up."]
        seen = set(order)
        rest = []
        for gname in groups:
            rest += [n for n in self.params if n.startswith(gname) and n not in seen]
            seen.update(rest)
        rest += [n for n in self.params if n not in seen]      # nothing today; keeps unknown parameters reducible
        self.grad_order = order + rest
        offs, tot = {}, 0
        for n in self.grad_order:
            offs[n] = tot
            tot += (self.params[n].numel() + 7) // 8 * 8      # every view 16-byte aligned in f32 AND in the bf16 mirror
        self.flat_grad = torch.zeros(tot, device=self.dev, dtype=torch.float32)
        self.g: Dict[str, torch.Tensor] = {}
        for n in self.grad_order:
            p = self.params[n]
            self.g[n] = self.flat_grad[offs[n]: offs[n] + p.numel()].view(p.shape)
        self.grad_offsets = offs
        # first element of the buffer's tail that is complete before the stage-1 backward (see _backward_main)
        self.ddp_split = offs[next(n for n in self.grad_order if n.startswith(E + "pmerging1."))]
        self.ddp_split3 = offs[next(n for n in self.grad_order if n.startswith(E + "stage3."))]
        assert self.ddp_split < self.ddp_split3 and all(
            n.startswith((E + "stage3.", E + "neck3.", "detect.", "model_up.")) for n in self.grad_order if offs[n] >= self.ddp_split3)
        assert all(not n.startswith((E + "stage1.", E + "patch_embed.", E + "pos_embed", E + "channel_embed", E + "chan_block"))
                   for n in self.grad_order if offs[n] >= self.ddp_split)
        fe = offs[E + "channel_embed_r.proj.weight"]
        self.g_fe_w = self.flat_grad[fe: fe + 4 * 48 * 16]
        self.g_fe_b = self.flat_grad[offs[E + "channel_embed_r.proj.bias"]:][: 4 * 48]
        self.g_fe_g = self.flat_grad[offs[E + "chan_block.norm1.weight"]:][: 4 * 48]
        self.g_fe_be = self.flat_grad[offs[E + "chan_block.norm1.bias"]:][: 4 * 48]

    def _build_param_buffer(self):
        """Re-home every parameter into ONE flat f32 buffer with the layout of the gradient buffer (`p.data` become views),
        so that the optimizer step, the EMA and the cast to the run dtype are one streaming kernel over
        (flat_param, flat_grad, momentum, ema) - optim.FusedSGD, csrc/optim.hip - instead of 273 small launches."""
        self.flat_param = torch.zeros_like(self.flat_grad)
        with torch.no_grad():
            for n in self.grad_order:
                p = self.params[n]
                v = self.flat_param[self.grad_offsets[n]: self.grad_offsets[n] + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self.flat_cast: Dict[torch.dtype, torch.Tensor] = {}     # run-dtype mirror of flat_param (bf16 path)
        self.param_cast_fresh = False     # True: flat_cast == cast(flat_param) (set by optim.FusedSGD.step, cleared by invalidate_params)
        self._cast_version = -1           # params_version() at the moment the mirror was last made fresh
        self._param_epoch = 0             # bumped by every write params_version() cannot see (optim.FusedSGD.step, invalidate_params)

    def params_version(self) -> int:
        """Sum of the parameters' autograd version counters: every in-place write through the Parameter objects
        (load_state_dict's copy_, a torch optimizer's add_, p.mul_() under no_grad) bumps it.  Writes through `p.data`
        aliases and raw kernels do not: those callers use invalidate_params()."""
        return sum(p._version for p in self.params.values())

    def mark_cast_fresh(self):
        """optim.FusedSGD.step: the fused kernel has just written cast(flat_param) into the mirror."""
        self.param_cast_fresh = True
        self._cast_version = self.params_version()
        self._param_epoch += 1

    def invalidate_params(self):
        """Tell the engine the f32 masters were changed by something other than optim.FusedSGD (load_state_dict, a torch
        optimizer, manual edits): the run-dtype mirror is re-cast at the next forward.  In-place writes through the
        Parameter objects are also detected by their version counters (params_version)."""
        self.param_cast_fresh = False
        self._param_epoch += 1

    def _check_param_views(self):
        for n in self.grad_order:
            p = self.params[n]
            if p.data_ptr() != self.flat_param.data_ptr() + 4 * self.grad_offsets[n]:
                raise RuntimeError(f"{n} no longer lives in the engine's flat parameter buffer (model.to()/.float()/.half() after the "
                                   "first forward re-allocates parameters): rebuild the engine with model._engine = None")

    def _claim_grads(self):
        """Make param.grad the views of the flat buffer; zero it when the grads were reset."""
        fresh = False
        for n, p in self.params.items():
            v = self.g[n]
            if p.grad is None:
                p.grad = v
                fresh = True
            elif p.grad.data_ptr() != v.data_ptr():
                raise RuntimeError(f"{n}.grad is not the engine's flat-buffer view; use optimizer.zero_grad(set_to_none=True) "
                                   "or leave .grad untouched between steps")
        return fresh

    # ------------------------------------------------------------------ prepared parameters
    def _prep_for(self, dt):
        P = self.prep.get(dt)
        if P is not None:
            return P
        dev = self.dev
        w: Dict[str, torch.Tensor] = {}
        wT: Dict[str, torch.Tensor] = {}
        descs_t, descs_f = [], []

        def add(desc_list, src, dst, dims, perm, dst_ld):
            d = L.PrepDesc()
            d.src, d.dst = src.data_ptr(), dst.data_ptr()
            d.d0, d.d1, d.d2 = dims
            d.p0, d.p1, d.p2 = perm
            d.dst_ld = dst_ld
            desc_list.append(d)

        # run-dtype mirror of the whole parameter buffer: every weight whose GEMM layout [N][K] IS its storage layout (all
        # nn.Linear and 1x1 Conv2d weights, pos_embed) is a VIEW of it - no per-step permute / cast launch for them; f32 runs
        # view the masters themselves
        if dt == torch.float32:
            mirror = self.flat_param
        else:
            mirror = self.flat_cast.get(dt)
            if mirror is None:
                mirror = self.flat_cast[dt] = torch.zeros_like(self.flat_param, dtype=dt)

        def mview(n, shape):
            o = self.grad_offsets[n]
            return mirror[o: o + self.params[n].numel()].view(shape)
        for n, p in self.params.items():
            if p.dim() < 2 or "relative_position_bias_table" in n or "channel_embed" in n or n.startswith("model_up."):
                continue        # (model_up: sr.SRBranch lays its own weights out)
            if n == E + "pos_embed":
                t = p.shape[1]
                w[n] = mview(n, (t * t, 192))
                continue
            N, K = p.shape[0], p.shape[1]
            taps = p.numel() // (N * K)
            Np = (N + 15) // 16 * 16 if n.startswith(self.det_name) else N     # Detect: 39 -> 48 zero-padded
            if taps == 1 and Np == N:
                w[n] = mview(n, (N, K))
            else:
                w[n] = torch.zeros(Np, taps * K, device=dev, dtype=dt)       # [N][tap*K + k]
                add(descs_t, p, w[n], (N, K, taps), (0, 2, 1), taps * K)
            wT[n] = torch.zeros(K, taps * Np, device=dev, dtype=dt)          # [K][tap*N + n]
            add(descs_t, p, wT[n], (N, K, taps), (1, 2, 0), taps * Np)
        # Linear MLPs on the bf16 path keep only the activation: the backward GEMM recomputes the pre-activation from
        # [fc1.weight | fc2.weight^T] ([4C][2C]) in its first K half (SODT_EPI_DGELU_RC)
        wcat: Dict[str, torch.Tensor] = {}
        if dt == torch.bfloat16:
            for n, p in self.params.items():
                if n.endswith("mlp.fc1.weight") and p.dim() == 2 and p.shape[0] == 4 * p.shape[1]:
                    Cc = p.shape[1]
                    p2 = self.params[n.replace("fc1", "fc2")]
                    wc = torch.zeros(4 * Cc, 2 * Cc, device=dev, dtype=dt)
                    add(descs_t, p, wc, (4 * Cc, Cc, 1), (0, 2, 1), 2 * Cc)
                    add(descs_t, p2, wc[:, Cc:], (Cc, 4 * Cc, 1), (1, 2, 0), 2 * Cc)
                    wcat[n] = wc
        # 2x2-conv MLPs with fc1 folded into the convolution: composed weights (re-made every forward by sodt_convmlp_compose)
        cmlp: Dict[str, Dict[str, torch.Tensor]] = {}
        if dt == torch.bfloat16:
            for n, p in self.params.items():
                if n.endswith("mlp.conv1.weight") and p.dim() == 4 and tuple(p.shape[2:]) == (2, 2) and p.shape[0] == p.shape[1]:
                    Cc = p.shape[0]
                    if Cc <= self.convmlp_fold_maxc and Cc % 64 == 0:
                        cmlp[n[: -len("mlp.conv1.weight")]] = dict(
                            weff=torch.zeros(Cc, 4 * Cc, device=dev, dtype=dt), weffT=torch.zeros(Cc, 4 * Cc, device=dev, dtype=dt),
                            beff=torch.zeros(Cc, device=dev), vtap=torch.zeros(4, Cc, device=dev))
        # f32 side: transposed relative-position tables, packed front-end parameters
        bias_t: Dict[str, torch.Tensor] = {}
        for n, p in self.params.items():
            if "relative_position_bias_table" in n:
                bias_t[n] = torch.zeros(p.shape[1], p.shape[0], device=dev, dtype=torch.float32)
                add(descs_f, p, bias_t[n], (p.shape[0], p.shape[1], 1), (1, 0, 2), p.shape[0])
        fe = {k: torch.zeros(4 * 48 * (16 if k == "w" else 1), device=dev, dtype=torch.float32) for k in ("w", "b", "g", "be")}
        for ci, c in enumerate("rgbi"):
            add(descs_f, self.params[E + f"channel_embed_{c}.proj.weight"], fe["w"][ci * 768:], (48, 16, 1), (0, 1, 2), 16)
            add(descs_f, self.params[E + f"channel_embed_{c}.proj.bias"], fe["b"][ci * 48:], (48, 1, 1), (0, 1, 2), 1)
            add(descs_f, self.params[E + f"chan_block.norm{ci + 1}.weight"], fe["g"][ci * 48:], (48, 1, 1), (0, 1, 2), 1)
            add(descs_f, self.params[E + f"chan_block.norm{ci + 1}.bias"], fe["be"][ci * 48:], (48, 1, 1), (0, 1, 2), 1)

        # fused W-MSA block kernel (csrc/wmsa_block.hip): one stage-ordered parameter pack per eligible block
        wmsa: Dict[str, torch.Tensor] = {}
        code = L.BF16 if dt == torch.bfloat16 else L.F32
        enc = self.model.image_encoder
        for sname in ("stage1", "stage2", "stage3"):
            for i, blk in enumerate(getattr(enc, sname)):
                nb = ops.wmsa_pack_bytes(blk.dim, HEADS, blk.window_size, code) if self.use_fused_wmsa else 0
                if nb > 0:
                    wmsa[E + f"{sname}.{i}."] = torch.zeros(nb // (2 if dt == torch.bfloat16 else 4), device=dev, dtype=dt)

        def table(descs):
            arr = (L.PrepDesc * len(descs))(*descs)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            mx = max(d.d0 * d.d1 * d.d2 for d in descs)
            return host.to(dev), len(descs), mx
        P = dict(w=w, wT=wT, wcat=wcat, cmlp=cmlp, bias_t=bias_t, fe=fe, tab_t=table(descs_t), tab_f=table(descs_f), dt=dt,
                 ones={}, keep=(descs_t, descs_f), wmsa=wmsa)
        self.prep[dt] = P
        return P

    def _run_prep(self, P):
        tt, nt, mt = P["tab_t"]
        ops.prep_weights(tt, nt, mt, L.BF16 if P["dt"] == torch.bfloat16 else L.F32)
        tf, nf, mf = P["tab_f"]
        ops.prep_weights(tf, nf, mf, L.F32)
        p = self.params
        for pre, wpk in P["wmsa"].items():
            ops.wmsa_pack(p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"], p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"],
                          p[pre + "attn.relative_position_bias_table"], p[pre + "norm1.weight"], p[pre + "norm1.bias"],
                          p[pre + "norm2.weight"], p[pre + "norm2.bias"], wpk, p[pre + "attn.qkv.weight"].shape[1], HEADS, 8)

    # ------------------------------------------------------------------ public entry
    MAX_PLANS = 4      # each plan owns a full activation workspace (31 GB at B=8 @1024^2 bf16): least recently used goes

    def run(self, x_rgb, x_ir, dt, training):
        if x_rgb.dim() != 4 or x_rgb.shape[1] != 3 or x_ir.dim() != 4 or x_rgb.shape[-1] != x_rgb.shape[-2]:
            raise ValueError("expected x_rgb (B,3,S,S) and x_ir (B,>=1,S,S)")
        B, _, S, _ = x_rgb.shape
        if S % 32:
            raise ValueError("S must be a multiple of 32 (4x patch stride, 8x8 windows on t/2 .. even t/4)")
        if x_ir.shape[0] != B or x_ir.shape[1] < 1 or tuple(x_ir.shape[-2:]) != (S, S):
            raise ValueError(f"x_ir must be (B={B}, >=1, {S}, {S}) like x_rgb, got {tuple(x_ir.shape)}")
        if x_rgb.device != self.dev or x_ir.device != self.dev:
            raise ValueError(f"inputs must be on the model's device {self.dev} (got {x_rgb.device}, {x_ir.device}): "
                             "the kernels take raw device pointers")
        x_rgb = x_rgb.float().contiguous()
        x_ir = x_ir.float().contiguous()
        key = (B, S, dt, training)
        plan = self.plans.pop(key, None)
        if plan is None:
            while len(self.plans) >= self.MAX_PLANS:
                self.plans.pop(next(iter(self.plans)))          # dicts keep insertion order: the first key is the LRU one
            plan = Plan(B, S, dt, training, self.dev)
        self.plans[key] = plan                                   # (re)insert as most recently used
        out_sr = None
        if training and torch.is_grad_enabled():
            pred = _EngineFn.apply(self, plan, x_rgb, x_ir, self._anchor)
            if self.sr:
                pred, out_sr = pred
        else:
            pred = self._forward(plan, x_rgb, x_ir)
            if self.sr and training:                              # model.py:284-287: the branch runs in training mode only
                out_sr = self._sr_forward(plan)
        return pred, self._features(plan), out_sr

    def decode(self, pred):
        B, na, ny, nx, no = pred.shape
        z = torch.empty(B, na * ny * nx, no, device=pred.device, dtype=torch.float32)
        ag = self.buffers[self.det_name + "anchor_grid"].reshape(-1).contiguous().float()
        ops.detect_decode(pred, ag, z, B, na, ny, nx, no, 4.0)
        return z

    def _features(self, plan):
        """y[0..10] of forward_once (model.py:246,281) as NCHW *views* of the workspace (valid until the
        next forward).  y4, y5, y8, y9 (upsample / concat) are never materialised by the kernels: the list
        builds them with torch on first access (``_LazyFeatures``), so callers that index them get tensors
        as in the reference and nobody else pays for them."""
        B, t = plan.B, plan.S // 4
        b = plan.bufs

        def nchw(name, h, c):
            return b[name].view(B, h, h, c).permute(0, 3, 1, 2)
        items = [nchw("f0", t, 256), nchw("f1", t // 2, 256), nchw("f2", t // 4, 512)] + [None] * self.nd
        for u in self.head_units:
            items[3 + u["k"]] = nchw(self._unit_out_name(u), t >> u["level"], u["c2"])
        y = _LazyFeatures(items, self.head_rules)
        if self.model.materialize_features:
            for i in sorted(self.head_rules):
                y[i]
        return y

    # ================================================================== forward
    def _forward(self, plan: Plan, x_rgb, x_ir):
        P = self._prep_for(plan.dt)
        B, S = plan.B, plan.S
        t = S // 4
        plan.gen += 1          # the saved activations of an earlier forward on this plan are gone (see _EngineFn.backward)
        # (0) run-dtype mirror of the masters: written by the fused optimizer step; re-cast here when anything else may have
        #     changed them (live call: the decision is per step)
        if plan.dt != torch.float32:
            ver = self.params_version()
            if not self.param_cast_fresh or ver != self._cast_version:
                ops.cast(self.flat_param, self.flat_cast[plan.dt], self.flat_param.numel())
                # stays valid until the masters change: fresh only when an optimizer that maintains the mirror is stepping
                self._cast_version = ver
        # (1) parameter preparation (recorded)
        if plan.fwd_pre is None:
            with ops.Recorder() as rec:
                self._run_prep(P)
            plan.fwd_pre = rec.calls
        else:
            ops.replay(plan.fwd_pre)
        # (2) front end: live call (input pointers change per step)
        x0 = plan.buf("x0", (B * t * t, 192))
        fe = P["fe"]
        cb = self.model.image_encoder.chan_block
        ca_ws, ca_shift = int(cb.window_size), int(cb.shift_size)
        if ca_ws == 1:      # the shipped configuration (backbone_vit.py:438): fused kernel
            ops.frontend_fwd(x_rgb, x_ir, x_ir.shape[1] * S * S, fe["w"], fe["b"], fe["g"], fe["be"], x0, B, S, 1)
        else:               # general window / shift form of CAttentionBlock
            e = plan.buf("fe.e", (B * t * t, 192), torch.float32)
            ops.patch_embed4_fwd(x_rgb, x_ir, x_ir.shape[1] * S * S, fe["w"], fe["b"], e, B, S)
            ops.cross_attn_ln_fwd(e, fe["g"], fe["be"], x0, B, S, ca_ws, ca_shift)
        # (3) encoder + head (recorded)
        if plan.fwd_main is None:
            with ops.Recorder() as rec:
                self._forward_main(plan, P)
            plan.fwd_main = rec.calls
        else:
            self._replay(plan, "fwd_main", plan.fwd_main, self.probes_fwd)
        # (4) Detect: live (fresh output tensor every call)
        T1 = B * t * t
        pred = torch.empty(B, self.na, t, t, self.no, device=self.dev, dtype=torch.float32)
        hu = next(u for u in self.head_units if u["k"] == self.head_out[0])
        ops.gemm_nt([SegSpec(plan.bufs[self._unit_out_name(hu)])], P["w"][self.det_name + "m.0.weight"], pred, T1, self.na * self.no,
                    self.head_out[1], bias=self.params[self.det_name + "m.0.bias"], detect=(self.na, self.no, t * t))
        return pred

    def _replay(self, plan: Plan, key: str, calls, probes):
        """Re-issue a recorded launch list.  With SODT_HIPGRAPH=1 (single-process runs without probes) the list is captured
        into a hipGraph on its first replay - every launch goes to the caller's stream and all pointers are plan-owned, so the
        capture is exact - and the graph is launched from then on: one host call instead of ~350 ctypes calls per list.
        Off by default: the step is GPU-bound (43 ms of kernels against ~0.5 ms of host replay that runs ahead of the GPU;
        185.6 vs 185.3 img/s measured), and the data-parallel path replays in segments around its all-reduce buckets."""
        if probes or not USE_HIPGRAPH or self.ddp is not None:
            ops.replay(calls, probes=probes)
            return
        g = plan.graphs.get(key)
        if g is None:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.graph(g, stream=side):
                ops.replay(calls)
            torch.cuda.current_stream().wait_stream(side)
            plan.graphs[key] = g
        g.replay()

    def _forward_main(self, plan: Plan, P):
        B, S = plan.B, plan.S
        t = S // 4
        p, w = self.params, P["w"]
        T1 = B * t * t
        x0 = plan.bufs["x0"]
        if plan.training and not self.fused:
            ops.zero_(plan.zpool("f"))               # BatchNorm column-statistics accumulators of every head conv
        # 1x1 token-mixing conv + bias + pos_embed (only when t matches: backbone_vit.py:215-217)
        x = plan.buf("xe", (T1, 192))
        use_pos = (t == self.img_t)
        plan.saved["use_pos"] = use_pos
        ops.gemm_nt([SegSpec(x0)], w[E + "patch_embed.proj.weight"], x, T1, 192, 192, bias=p[E + "patch_embed.proj.bias"],
                    resid=w[E + "pos_embed"] if use_pos else None, rmod=t * t if use_pos else 0)
        enc = self.model.image_encoder
        outs = {}
        H = t
        for i, blk in enumerate(enc.stage1):
            x = self._block_fwd(plan, P, f"stage1.{i}", blk, x, B, H, H)
            outs[i] = x
        f0 = plan.buf("f0", (T1, 256))
        ops.gemm_nt([SegSpec(outs[4]), SegSpec(outs[5])], w[E + "neck1.weight"], f0, T1, 256, 384)
        x = self._merge_fwd(plan, P, "pmerging1", x, B, H, H, 192)
        H //= 2
        for i, blk in enumerate(enc.stage2):
            x = self._block_fwd(plan, P, f"stage2.{i}", blk, x, B, H, H)
        T2 = B * H * H
        f1 = plan.buf("f1", (T2, 256))
        ops.gemm_nt([SegSpec(x)], w[E + "neck2.weight"], f1, T2, 256, 384)
        x = self._merge_fwd(plan, P, "pmerging2", x, B, H, H, 384)
        H //= 2
        for i, blk in enumerate(enc.stage3):
            x = self._block_fwd(plan, P, f"stage3.{i}", blk, x, B, H, H)
        T3 = B * H * H
        f2 = plan.buf("f2", (T3, 512))
        ops.gemm_nt([SegSpec(x)], w[E + "neck3.weight"], f2, T3, 512, 768)
        # ---- head (models/model.yaml:65-74 and the variants _check_head accepts), token-major
        self._head_fwd(plan, P)

    def _head_fwd(self, plan: Plan, P):
        B, t = plan.B, plan.S // 4
        b = plan.bufs
        for u in self.head_units:
            H = t >> u["level"]
            M = B * H * H
            tag, pname = f"h{u['k']}", f"detect.{u['k']}."
            plain = all(shr == 0 for _, _, shr in u["parts"]) and ((u["kind"] == "Conv" and u["ksize"] == 1) or u["kind"] == "SPP")
            segs = []
            for ref, c, shr in u["parts"]:
                src = b[("f0", "f1", "f2")[ref[1]]] if ref[0] == "enc" else b[self._unit_out_name(self._unit(ref[1]))]
                segs.append(SegSpec(src) if plain else SegSpec(src, c, 0, 0, 0, 1, shr, H >> shr, H >> shr))
            if u["kind"] == "Conv":
                if u["ksize"] == 3:
                    segs = [SegSpec(s_.t, s_.klen, 0, dy, dx, 1, s_.shr, s_.Hi, s_.Wi) for (dy, dx) in TAPS3 for s_ in segs]
                self._conv_fwd(plan, P, tag, pname, segs, None if plain else (H, H), M, u["c1"] * u["ksize"] ** 2, u["c2"], u["ksize"])
            elif u["kind"] == "C3":
                self._c3_fwd(plan, P, tag, pname, segs, (H, H), M, u["c1"], u["c2"])
            else:
                self._spp_fwd(plan, P, tag, pname, segs, (H, H), M, u["c1"], u["c2"], B)

    def _unit(self, k):
        return next(u for u in self.head_units if u["k"] == k)

    # ------------------------------------------------------------------ Swin block
    def _block_geo(self, blk, H, W):
        ws, shift = blk.window_size, blk.shift_size
        if min(H, W) <= ws:
            ws, shift = min(H, W), 0
            if ws < 8 or 64 % ws:
                raise NotImplementedError(f"a {H}x{W}-token stage is ONE {ws}x{ws} window (backbone_vit.py:1042-1045: no partition below "
                                          "the window size); the attention kernels walk 64-token tiles of whole window rows (window "
                                          "side 8, 16 or 32): below S = 576 use S = 128, 256 or 512 (from 576 on every multiple of 64 runs)")
        L2 = 2 * ws - 1
        if blk.attn.relative_position_bias_table.shape[0] != L2 * L2:
            raise ValueError(f"input resolution gives a {ws}x{ws} window but the model was built with a "
                             f"{blk.attn.window_size[0]}-window bias table; build Model with the matching img_size")
        return ws, shift

    def _block_fwd(self, plan, P, tag, blk, x_in, B, H, W):
        ops.set_tag(tag)
        p, w = self.params, P["w"]
        pre = E + tag + "."
        Cc = blk.dim
        M = B * H * W
        ws, shift = self._block_geo(blk, H, W)
        xn1 = plan.buf(tag + ".xn1", (M, Cc))
        st1 = plan.buf(tag + ".st1", (M, 2), torch.float32)
        xm = plan.buf(tag + ".xm", (M, Cc))
        xn2 = plan.buf(tag + ".xn2", (M, Cc))
        st2 = plan.buf(tag + ".st2", (M, 2), torch.float32)
        ao = plan.buf(tag + ".ao", (M, Cc))
        wpk = P["wmsa"].get(pre) if (ws == 8 and H % 8 == 0 and W % 8 == 0) else None
        fused = wpk is not None
        # Resolution not a multiple of the window: the reference zero-pads AFTER norm1 and crops after the attention
        # (backbone_vit.py:619-672, window_partition / window_unpartition), so a pad token enters the attention as the qkv BIAS.
        # Unshifted blocks (stage 3 at S = 640, 768, ...: 40 x 40, 48 x 48 tokens against the 32-token window) run here through the
        # spatial K-segments: the QKV GEMM writes the padded grid (rows outside read zeros), the attention runs on it, the
        # projection reads it back cropped.  Shifted blocks would also need the reference's mask of the UNPADDED grid: not built.
        Hp, Wp = H + (-H) % ws, W + (-W) % ws
        padded = (Hp, Wp) != (H, W)
        if padded and shift > 0:
            raise NotImplementedError(f"{tag}: {H}x{W} tokens are not a multiple of the {ws}-token window of a SHIFTED block (the "
                                      "reference's zero padding, backbone_vit.py:619-643, is built for unshifted blocks only): choose "
                                      "an input size S that is a multiple of 64")
        if fused:
            # LN1 + QKV + window attention + proj + residual + LN2 in ONE launch (csrc/wmsa_hg.hip / wmsa_block.hip); training also
            # writes what the backward needs: xn1, the attention output, the LayerNorm statistics and the window-major
            # log-sum-exp.  q / k / v are saved by the f32 parity kernel only (sodt_window_attn_bwd_wm reads them back); the bf16
            # backward recomputes them from xn1 and the parameter pack (sodt_wmsa_block_bwd): 604 MB less written per launch
            qkvw = plan.buf(tag + ".qkvw", (M // 64, HEADS, 3, 64, Cc // HEADS)) if (plan.training and plan.dt == torch.float32) else None
            lsew = plan.buf(tag + ".lsew", (M // 64, HEADS, 64), torch.float32)
            if plan.training:
                ops.wmsa_block_fwd(x_in, wpk, xm, xn2, st1, st2, xn1, qkvw, lsew, ao, B, H, W, Cc, HEADS, ws, shift)
            else:
                ops.wmsa_block_fwd(x_in, wpk, xm, xn2, None, None, None, None, None, None, B, H, W, Cc, HEADS, ws, shift)
        elif padded:
            Mp = B * Hp * Wp
            ops.layernorm_fwd(x_in, p[pre + "norm1.weight"], p[pre + "norm1.bias"], xn1, st1, M, Cc)
            qkv = plan.buf(tag + ".qkv", (Mp, 3 * Cc))
            ops.gemm_nt([SegSpec(xn1, Cc, 0, 0, 0, 1, 0, H, W)], w[pre + "attn.qkv.weight"], qkv, Mp, 3 * Cc, Cc, spatial=(Hp, Wp),
                        bias=p[pre + "attn.qkv.bias"])
            lse = plan.buf(tag + ".lse", (Mp, HEADS), torch.float32)
            aop = plan.buf(tag + ".aop", (Mp, Cc))
            ops.window_attn_fwd(qkv, P["bias_t"][pre + "attn.relative_position_bias_table"], aop, lse, B, Hp, Wp, Cc, HEADS, ws, 0)
            ops.gemm_nt([SegSpec(aop, Cc, 0, 0, 0, 1, 0, Hp, Wp)], w[pre + "attn.proj.weight"], xm, M, Cc, Cc, spatial=(H, W),
                        bias=p[pre + "attn.proj.bias"], resid=x_in)
            ops.layernorm_fwd(xm, p[pre + "norm2.weight"], p[pre + "norm2.bias"], xn2, st2, M, Cc)
        else:
            ops.layernorm_fwd(x_in, p[pre + "norm1.weight"], p[pre + "norm1.bias"], xn1, st1, M, Cc)
            qkv = plan.buf(tag + ".qkv", (M, 3 * Cc))
            ops.gemm_nt([SegSpec(xn1)], w[pre + "attn.qkv.weight"], qkv, M, 3 * Cc, Cc, bias=p[pre + "attn.qkv.bias"])
            lse = plan.buf(tag + ".lse", (M, HEADS), torch.float32)
            ops.window_attn_fwd(qkv, P["bias_t"][pre + "attn.relative_position_bias_table"], ao, lse, B, H, W, Cc, HEADS, ws, shift)
            ops.gemm_nt([SegSpec(ao)], w[pre + "attn.proj.weight"], xm, M, Cc, Cc, bias=p[pre + "attn.proj.bias"], resid=x_in)
            ops.layernorm_fwd(xm, p[pre + "norm2.weight"], p[pre + "norm2.bias"], xn2, st2, M, Cc)
        xo = plan.buf(tag + ".xo", (M, Cc))
        if blk.mlp.linear and self.use_fused_mlp and ops.mlp_fused_ok(M, Cc, plan.dt) and ops.mlp_recompute_ok(M, Cc, plan.dt) and (pre + "mlp.fc1.weight") in P["wcat"]:
            # fc1 + GELU + fc2 + residual in ONE launch (csrc/mlp.hip): the 4C-wide hidden activation leaves the CU only in training,
            # as GELU(h) for fc2's weight gradient (the backward recomputes h inside the dh GEMM, as below)
            ha = plan.buf(tag + ".ha", (M, 4 * Cc)) if plan.training else None
            ops.mlp_fwd(xn2, w[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], w[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"],
                        xm, xo, ha, M, Cc)
        elif blk.mlp.linear:
            ha = plan.buf(tag + ".ha", (M, 4 * Cc))
            if ops.mlp_recompute_ok(M, Cc, ha.dtype) and (pre + "mlp.fc1.weight") in P["wcat"]:
                # only GELU(h) is written; backward recomputes h inside the dh GEMM
                ops.gemm_nt([SegSpec(xn2)], w[pre + "mlp.fc1.weight"], ha, M, 4 * Cc, Cc, bias=p[pre + "mlp.fc1.bias"], gelu_only=True)
            else:
                hp = plan.buf(tag + ".hp", (M, 4 * Cc))
                ops.gemm_nt([SegSpec(xn2)], w[pre + "mlp.fc1.weight"], hp, M, 4 * Cc, Cc, bias=p[pre + "mlp.fc1.bias"], gelu_out=ha)
            ops.gemm_nt([SegSpec(ha)], w[pre + "mlp.fc2.weight"], xo, M, Cc, 4 * Cc, bias=p[pre + "mlp.fc2.bias"], resid=xm)
        elif pre in P["cmlp"]:
            # fc1 folded into the 2x2 convolution (csrc/convmlp.hip): composed weights, the convolution straight on xn2, a correction
            # on the last column / row (where the padded fc1 output is zero including its bias)
            cm = P["cmlp"][pre]
            ops.convmlp_compose(p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.conv1.weight"], p[pre + "mlp.conv1.bias"],
                                cm["weff"], cm["weffT"], cm["beff"], cm["vtap"], Cc)
            cp = plan.buf(tag + ".cp", (M, Cc))
            ca = plan.buf(tag + ".ca", (M, Cc))
            segs = [SegSpec(xn2, Cc, 0, dy, dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_nt(segs, cm["weff"], cp, M, Cc, 4 * Cc, spatial=(H, W), bias=cm["beff"], gelu_out=ca)
            ops.convmlp_border_fix(cp, ca, cm["vtap"], B, H, W, Cc)
            ops.gemm_nt([SegSpec(ca)], w[pre + "mlp.fc2.weight"], xo, M, Cc, Cc, bias=p[pre + "mlp.fc2.bias"], resid=xm)
        else:
            u = plan.buf(tag + ".u", (M, Cc))
            ops.gemm_nt([SegSpec(xn2)], w[pre + "mlp.fc1.weight"], u, M, Cc, Cc, bias=p[pre + "mlp.fc1.bias"])
            cp = plan.buf(tag + ".cp", (M, Cc))
            ca = plan.buf(tag + ".ca", (M, Cc))
            segs = [SegSpec(u, Cc, 0, dy, dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_nt(segs, w[pre + "mlp.conv1.weight"], cp, M, Cc, 4 * Cc, spatial=(H, W), bias=p[pre + "mlp.conv1.bias"],
                        gelu_out=ca)
            ops.gemm_nt([SegSpec(ca)], w[pre + "mlp.fc2.weight"], xo, M, Cc, Cc, bias=p[pre + "mlp.fc2.bias"], resid=xm)
        plan.saved[tag] = dict(x_in=x_in, geo=(B, H, W, Cc, ws, shift), fused=fused, wpk=wpk, pad=(Hp, Wp) if padded else None)
        return xo

    def _block_bwd(self, plan, P, tag, blk, dY, dX):
        """dY: gradient wrt the block output; writes the gradient wrt the block input into dX."""
        ops.set_tag(tag + ".bwd")
        p, wT, g, b = self.params, P["wT"], self.g, plan.bufs
        pre = E + tag + "."
        sv = plan.saved[tag]
        B, H, W, Cc, ws, shift = sv["geo"]
        M = B * H * W
        x_in = sv["x_in"]
        xm, xn2, xn1, ao = b[tag + ".xm"], b[tag + ".xn2"], b[tag + ".xn1"], b[tag + ".ao"]
        dxn = plan.buf(f"g.dxn.{Cc}", (M, Cc))
        dxm = plan.buf(f"g.dxm.{Cc}", (M, Cc))
        if blk.mlp.linear:
            ha = b[tag + ".ha"]
            dh = plan.buf(f"g.dh.{Cc}", (M, 4 * Cc))
            ops.gemm_tn(dY, [SegSpec(ha)], g[pre + "mlp.fc2.weight"], M, Cc, 4 * Cc, dbias=g[pre + "mlp.fc2.bias"])
            if (tag + ".hp") in b:
                ops.gemm_nt([SegSpec(dY)], wT[pre + "mlp.fc2.weight"], dh, M, 4 * Cc, Cc, dgelu_aux=b[tag + ".hp"])
            else:       # dh = (dY fc2.weight) * gelu'(xn2 fc1.weight^T + b1): pre-activation recomputed in the first K half
                ops.gemm_nt([SegSpec(xn2), SegSpec(dY)], P["wcat"][pre + "mlp.fc1.weight"], dh, M, 4 * Cc, 2 * Cc,
                            bias=p[pre + "mlp.fc1.bias"], dgelu_rc=True)
            ops.gemm_tn(dh, [SegSpec(xn2)], g[pre + "mlp.fc1.weight"], M, 4 * Cc, Cc, dbias=g[pre + "mlp.fc1.bias"])
            dln2 = (dh, 4 * Cc)
        elif pre in P["cmlp"]:
            cm = P["cmlp"][pre]
            cp, ca = b[tag + ".cp"], b[tag + ".ca"]
            dc = plan.buf(f"g.dc.{Cc}", (M, Cc))
            ops.gemm_tn(dY, [SegSpec(ca)], g[pre + "mlp.fc2.weight"], M, Cc, Cc, dbias=g[pre + "mlp.fc2.bias"])
            ops.gemm_nt([SegSpec(dY)], wT[pre + "mlp.fc2.weight"], dc, M, Cc, Cc, dgelu_aux=cp)
            # d(Weff) = dc^T xn2(taps) (+ the column sums of dc), then the parameter gradients of fc1 / conv1 by the chain rule
            scr = plan.buf(f"g.cmlp.{Cc}", (Cc * 4 * Cc + 4 * Cc,), torch.float32)
            dweff, colsum, bs = scr[: Cc * 4 * Cc].view(Cc, 4 * Cc), scr[Cc * 4 * Cc: Cc * 4 * Cc + Cc], scr[Cc * 4 * Cc + Cc:].view(3, Cc)
            ops.zero_(scr)
            segs = [SegSpec(xn2, Cc, 0, dy, dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_tn(dc, segs, dweff, M, Cc, 4 * Cc, spatial=(H, W), dbias=colsum)
            ops.convmlp_border_sums(dc, bs, B, H, W, Cc)
            ops.convmlp_decompose(dweff, colsum, bs, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"], p[pre + "mlp.conv1.weight"],
                                  g[pre + "mlp.conv1.weight"], g[pre + "mlp.conv1.bias"], g[pre + "mlp.fc1.weight"], g[pre + "mlp.fc1.bias"], Cc)
            dln2 = None           # d(xn2) comes from ONE tap GEMM with the composed weights (no du, no du W1)
        else:
            u, cp, ca = b[tag + ".u"], b[tag + ".cp"], b[tag + ".ca"]
            dc = plan.buf(f"g.dc.{Cc}", (M, Cc))
            du = plan.buf(f"g.du.{Cc}", (M, Cc))
            ops.gemm_tn(dY, [SegSpec(ca)], g[pre + "mlp.fc2.weight"], M, Cc, Cc, dbias=g[pre + "mlp.fc2.bias"])
            ops.gemm_nt([SegSpec(dY)], wT[pre + "mlp.fc2.weight"], dc, M, Cc, Cc, dgelu_aux=cp)
            segs = [SegSpec(u, Cc, 0, dy, dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_tn(dc, segs, g[pre + "mlp.conv1.weight"], M, Cc, 4 * Cc, spatial=(H, W), dbias=g[pre + "mlp.conv1.bias"],
                        kperm=(Cc, 4))
            segs = [SegSpec(dc, Cc, 0, -dy, -dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_nt(segs, wT[pre + "mlp.conv1.weight"], du, M, Cc, 4 * Cc, spatial=(H, W))
            ops.gemm_tn(du, [SegSpec(xn2)], g[pre + "mlp.fc1.weight"], M, Cc, Cc, dbias=g[pre + "mlp.fc1.bias"])
            dln2 = (du, Cc)
        # d(xn2) = dln2 @ fc1.weight, then the LayerNorm-2 backward (+ the residual path's dY).  (Rounds 4-5 carried a GEMM with the
        # LayerNorm backward as its epilogue for the 192-column case: slower than these two launches, removed in round 6 -
        # profiles/r04_lnfold_ab.md)
        if dln2 is None:
            segs = [SegSpec(dc, Cc, 0, -dy, -dx, 1, 0, H, W) for (dy, dx) in TAPS2]
            ops.gemm_nt(segs, cm["weffT"], dxn, M, Cc, 4 * Cc, spatial=(H, W))
            ops.layernorm_bwd(dxn, xm, b[tag + ".st2"], p[pre + "norm2.weight"], dY, dxm, g[pre + "norm2.weight"], g[pre + "norm2.bias"], M, Cc)
        else:
            ops.gemm_nt([SegSpec(dln2[0])], wT[pre + "mlp.fc1.weight"], dxn, M, Cc, dln2[1])
            ops.layernorm_bwd(dxn, xm, b[tag + ".st2"], p[pre + "norm2.weight"], dY, dxm, g[pre + "norm2.weight"], g[pre + "norm2.bias"], M, Cc)
        # attention
        if sv.get("pad"):
            self._padded_attn_bwd(plan, P, tag, pre, sv, dxm, dxn, dX)
            return
        ops.gemm_tn(dxm, [SegSpec(ao)], g[pre + "attn.proj.weight"], M, Cc, Cc, dbias=g[pre + "attn.proj.bias"])
        dao = dxn
        ops.gemm_nt([SegSpec(dxm)], wT[pre + "attn.proj.weight"], dao, M, Cc, Cc)
        dqkv = plan.buf(f"g.dqkv.{Cc}", (M, 3 * Cc))
        L2 = 2 * ws - 1
        dbt = plan.buf(f"g.dbt.{L2}", (HEADS, L2 * L2), torch.float32, zero=True)
        if sv["fused"] and plan.dt == torch.bfloat16:
            ops.wmsa_block_bwd(xn1, sv["wpk"], P["bias_t"][pre + "attn.relative_position_bias_table"], dao, b[tag + ".lsew"],
                               dqkv, dbt, B, H, W, Cc, HEADS, ws, shift)
        elif sv["fused"]:
            ops.window_attn_bwd_wm(b[tag + ".qkvw"], P["bias_t"][pre + "attn.relative_position_bias_table"], dao, b[tag + ".lsew"],
                                   dqkv, dbt, B, H, W, Cc, HEADS, ws, shift)
        else:
            scratch = plan.buf(f"g.attn_scratch.{Cc}", (M * (Cc + HEADS),), torch.float32, zero=True) if ws * ws > 64 else None
            ops.window_attn_bwd(b[tag + ".qkv"], P["bias_t"][pre + "attn.relative_position_bias_table"], ao, dao, b[tag + ".lse"], dqkv,
                                dbt, scratch, B, H, W, Cc, HEADS, ws, shift)
        ops.transpose_f32(dbt, g[pre + "attn.relative_position_bias_table"], HEADS, L2 * L2, accumulate=2)
        ops.gemm_tn(dqkv, [SegSpec(xn1)], g[pre + "attn.qkv.weight"], M, 3 * Cc, Cc, dbias=g[pre + "attn.qkv.bias"])
        ops.gemm_nt([SegSpec(dqkv)], wT[pre + "attn.qkv.weight"], dxn, M, Cc, 3 * Cc)
        ops.layernorm_bwd(dxn, x_in, b[tag + ".st1"], p[pre + "norm1.weight"], dxm, dX, g[pre + "norm1.weight"], g[pre + "norm1.bias"], M, Cc)

    def _padded_attn_bwd(self, plan, P, tag, pre, sv, dxm, dxn, dX):
        """Attention half of the backward of an UNSHIFTED block whose resolution is not a multiple of its window (see _block_fwd): the
        projection's input gradient lands on the padded grid (zeros outside: the crop's adjoint), the attention backward runs on it,
        qkv.weight sees zero rows for the pad tokens while qkv.bias collects their gradient (their q / k / v ARE the bias), and the
        input gradient is read back cropped."""
        p, wT, g, b = self.params, P["wT"], self.g, plan.bufs
        B, H, W, Cc, ws, _ = sv["geo"]
        Hp, Wp = sv["pad"]
        M, Mp = B * H * W, B * Hp * Wp
        aop, qkv, lse = b[tag + ".aop"], b[tag + ".qkv"], b[tag + ".lse"]
        ops.gemm_tn(dxm, [SegSpec(aop, Cc, 0, 0, 0, 1, 0, Hp, Wp)], g[pre + "attn.proj.weight"], M, Cc, Cc, spatial=(H, W),
                    dbias=g[pre + "attn.proj.bias"])
        daop = plan.buf(f"g.daop.{Cc}", (Mp, Cc))
        ops.gemm_nt([SegSpec(dxm, Cc, 0, 0, 0, 1, 0, H, W)], wT[pre + "attn.proj.weight"], daop, Mp, Cc, Cc, spatial=(Hp, Wp))
        dqkv = plan.buf(f"g.dqkvp.{Cc}", (Mp, 3 * Cc))
        L2 = 2 * ws - 1
        dbt = plan.buf(f"g.dbt.{L2}", (HEADS, L2 * L2), torch.float32, zero=True)
        scratch = plan.buf(f"g.attn_scratchp.{Cc}", (Mp * (Cc + HEADS),), torch.float32, zero=True) if ws * ws > 64 else None
        ops.window_attn_bwd(qkv, P["bias_t"][pre + "attn.relative_position_bias_table"], aop, daop, lse, dqkv, dbt, scratch,
                            B, Hp, Wp, Cc, HEADS, ws, 0)
        ops.transpose_f32(dbt, g[pre + "attn.relative_position_bias_table"], HEADS, L2 * L2, accumulate=2)
        ops.gemm_tn(dqkv, [SegSpec(b[tag + ".xn1"], Cc, 0, 0, 0, 1, 0, H, W)], g[pre + "attn.qkv.weight"], Mp, 3 * Cc, Cc,
                    spatial=(Hp, Wp), dbias=g[pre + "attn.qkv.bias"])
        ops.gemm_nt([SegSpec(dqkv, 3 * Cc, 0, 0, 0, 1, 0, Hp, Wp)], wT[pre + "attn.qkv.weight"], dxn, M, Cc, 3 * Cc, spatial=(H, W))
        ops.layernorm_bwd(dxn, sv["x_in"], b[tag + ".st1"], p[pre + "norm1.weight"], dxm, dX, g[pre + "norm1.weight"],
                          g[pre + "norm1.bias"], M, Cc)

    # ------------------------------------------------------------------ PatchMerging
    def _merge_fwd(self, plan, P, tag, x, B, H, W, Cc):
        ops.set_tag(tag)
        p, w = self.params, P["w"]
        pre = E + tag + "."
        M2 = B * (H // 2) * (W // 2)
        z = plan.buf(tag + ".z", (M2, 2 * Cc))
        segs = [SegSpec(x, Cc, 0, dy, dx, 2, 0, H, W) for (dy, dx) in MERGE]
        ops.gemm_nt(segs, w[pre + "reduction.weight"], z, M2, 2 * Cc, 4 * Cc, spatial=(H // 2, W // 2))
        y = plan.buf(tag + ".y", (M2, 2 * Cc))
        st = plan.buf(tag + ".st", (M2, 2), torch.float32)
        ops.layernorm_fwd(z, p[pre + "norm.weight"], p[pre + "norm.bias"], y, st, M2, 2 * Cc)
        plan.saved[tag] = dict(x=x, geo=(B, H, W, Cc))
        return y

    def _merge_bwd(self, plan, P, tag, dY, dX):
        p, wT, g, b = self.params, P["wT"], self.g, plan.bufs
        pre = E + tag + "."
        sv = plan.saved[tag]
        B, H, W, Cc = sv["geo"]
        M2 = B * (H // 2) * (W // 2)
        dz = plan.buf(tag + ".dz", (M2, 2 * Cc))
        ops.layernorm_bwd(dY, b[tag + ".z"], b[tag + ".st"], p[pre + "norm.weight"], None, dz, g[pre + "norm.weight"],
                          g[pre + "norm.bias"], M2, 2 * Cc)
        segs = [SegSpec(sv["x"], Cc, 0, dy, dx, 2, 0, H, W) for (dy, dx) in MERGE]
        ops.gemm_tn(dz, segs, g[pre + "reduction.weight"], M2, 2 * Cc, 4 * Cc, spatial=(H // 2, W // 2))
        for tap, (dy, dx) in enumerate(MERGE):   # the four taps tile the input grid exactly once
            ops.gemm_nt([SegSpec(dz, 2 * Cc, 0, 0, 0, 1, 0, H // 2, W // 2)], wT[pre + "reduction.weight"], dX, M2, Cc, 2 * Cc,
                        spatial=(H // 2, W // 2), w_off=tap * Cc * 2 * Cc, oscatter=(2, dy, dx, H, W))

    # ------------------------------------------------------------------ head units
    def _conv_fwd(self, plan, P, tag, pname, segs, spatial, M, K, Cout, k, out=None):
        """Conv2d(bias=False)+BN+SiLU (common.py:38-50) as GEMM (+f64 column stats) -> finalize -> normalise+SiLU.
        out: write the result into the first Cout columns of this wider [M][ld] buffer (a concat slice) instead of tag.y."""
        ops.set_tag(tag)
        p, w, bufs = self.params, P["w"], self.buffers
        y = plan.buf(tag + ".y", (M, Cout)) if out is None else out
        ldy = y.shape[-1]
        wname = pname + "conv.weight"
        if self.fused:
            ones = P["ones"].setdefault(Cout, torch.ones(Cout, device=self.dev))
            ops.gemm_nt(segs, w[wname], y, M, Cout, K, spatial=spatial, affine=(ones, p[pname + "conv.bias"]), ldc=ldy)
        elif plan.training:
            z = plan.buf(tag + ".z", (M, Cout))
            stats = plan.zbuf("f", tag + ".stats", (L.STATS_REPL, 2, Cout), torch.float64)   # zeroed with the pool (_forward_main)
            mr = plan.buf(tag + ".mr", (2, Cout), torch.float32)
            direct = self._direct3(plan, segs, K, Cout, k)
            if direct:
                # the 3x3 at 64 -> 64 channels of the stride-4 C3's Bottleneck: the direct kernel reads its input once (csrc/conv3.hip;
                # the nine-segment GEMM ran at 0.11-0.15 of its HBM roofline), the batch statistics come from the stored output
                H, W = spatial
                ops.conv3_c64_fwd(segs[4].t, w[wname], z, M // (H * W), H, W)
                ops.col_stats(z, stats, M, Cout)
            else:
                ops.gemm_nt(segs, w[wname], z, M, Cout, K, spatial=spatial, stats=stats)
            ops.bn_finalize(stats, mr, bufs[pname + "bn.running_mean"], bufs[pname + "bn.running_var"], M, Cout, 1e-3, 0.03)
            ops.bn_silu_fwd(z, mr, p[pname + "bn.weight"], p[pname + "bn.bias"], y, ldy, M, Cout)
        else:
            mr = plan.buf(tag + ".mr", (2, Cout), torch.float32)
            sc = plan.buf(tag + ".sc", (2, Cout), torch.float32)
            ops.bn_finalize(None, mr, bufs[pname + "bn.running_mean"], bufs[pname + "bn.running_var"], M, Cout, 1e-3, 0.03)
            ops.bn_affine(mr, p[pname + "bn.weight"], p[pname + "bn.bias"], sc[0], sc[1], Cout)
            ops.gemm_nt(segs, w[wname], y, M, Cout, K, spatial=spatial, affine=(sc[0], sc[1]), ldc=ldy)
        plan.saved[tag] = dict(segs=segs, spatial=spatial, M=M, K=K, Cout=Cout, k=k, pname=pname,
                               direct=bool(plan.training and not self.fused and self._direct3(plan, segs, K, Cout, k)))
        return y

    def _direct3(self, plan, segs, K, Cout, k):
        """a 3x3 Conv at 64 -> 64 channels on one dense bf16 tensor: sodt_conv3x3_c64_* (forward, input gradient, weight gradient)"""
        if not (self.use_direct_conv3 and k == 3 and Cout == 64 and K == 576 and plan.dt == torch.bfloat16 and len(segs) == 9):
            return False
        s0 = segs[4]
        return (s0.klen == 64 and s0.ld == 64 and s0.coff == 0 and s0.shr == 0 and s0.mul == 1 and s0.dy == 0 and s0.dx == 0
                and all(s_.t is s0.t for s_ in segs))

    def _conv_bwd(self, plan, tag, dy, lddy, dy_off):
        """BN+SiLU backward and the conv weight gradient; returns dz (gradient at the conv output)."""
        ops.set_tag(tag + ".bwd")
        p, g, b = self.params, self.g, plan.bufs
        sv = plan.saved[tag]
        M, K, Cout, k, pname = sv["M"], sv["K"], sv["Cout"], sv["k"], sv["pname"]
        red = plan.zbuf("b", tag + ".red", (2, Cout), torch.float64)                          # zeroed with the pool (_backward_main)
        dz = plan.buf(tag + ".dz", (M, Cout))
        ops.bn_silu_bwd_reduce(dy, lddy, b[tag + ".z"], b[tag + ".mr"], p[pname + "bn.weight"], p[pname + "bn.bias"], red, M, Cout,
                               dy_off=dy_off)
        ops.bn_silu_bwd_apply(dy, lddy, b[tag + ".z"], b[tag + ".mr"], p[pname + "bn.weight"], p[pname + "bn.bias"], red, dz,
                              g[pname + "bn.weight"], g[pname + "bn.bias"], M, Cout, dy_off=dy_off)
        cin = K // (k * k)
        if sv.get("direct"):
            H, W = sv["spatial"]
            scr = plan.buf("g.c64.scratch", (ops.conv3_c64_wgrad_scratch_floats(),), torch.float32)
            ops.conv3_c64_wgrad(dz, sv["segs"][4].t, g[pname + "conv.weight"], None, scr, M // (H * W), H, W)
        else:
            ops.gemm_tn(dz, sv["segs"], g[pname + "conv.weight"], M, Cout, K, spatial=sv["spatial"],
                        kperm=(cin, k * k) if k > 1 else None)
        return dz

    def _c3_fwd(self, plan, P, tag, pname, segs, spatial, M, c1, c2):
        c_ = c2 // 2
        H, W = spatial
        a1 = self._conv_fwd(plan, P, tag + ".cv1", pname + "cv1.", segs, spatial, M, c1, c_, 1)
        a2 = self._conv_fwd(plan, P, tag + ".m1", pname + "m.0.cv1.", [SegSpec(a1)], None, M, c_, c_, 1)
        s3 = [SegSpec(a2, c_, 0, dy, dx, 1, 0, H, W) for (dy, dx) in TAPS3]
        a3 = self._conv_fwd(plan, P, tag + ".m2", pname + "m.0.cv2.", s3, spatial, M, 9 * c_, c_, 3)
        b1 = self._conv_fwd(plan, P, tag + ".cv2", pname + "cv2.", segs, spatial, M, c1, c_, 1)
        out = self._conv_fwd(plan, P, tag + ".cv3", pname + "cv3.", [SegSpec(a3), SegSpec(b1)], None, M, 2 * c_, c2, 1)
        plan.saved[tag] = dict(M=M, c1=c1, c2=c2, spatial=spatial, pname=pname)
        return out

    def _c3_bwd(self, plan, P, tag, dout, ldd, d_off):
        """returns d(concatenated input) [M][c1]."""
        wT = P["wT"]
        sv = plan.saved[tag]
        M, c1, c2, (H, W), pname = sv["M"], sv["c1"], sv["c2"], sv["spatial"], sv["pname"]
        c_ = c2 // 2
        dz3 = self._conv_bwd(plan, tag + ".cv3", dout, ldd, d_off)
        dcat = plan.buf(tag + ".dcat", (M, 2 * c_))
        ops.gemm_nt([SegSpec(dz3)], wT[pname + "cv3.conv.weight"], dcat, M, 2 * c_, c2)
        din = plan.buf(tag + ".din", (M, c1))
        dzb = self._conv_bwd(plan, tag + ".cv2", dcat, 2 * c_, c_)
        ops.gemm_nt([SegSpec(dzb)], wT[pname + "cv2.conv.weight"], din, M, c1, c_)
        dza3 = self._conv_bwd(plan, tag + ".m2", dcat, 2 * c_, 0)
        da = plan.buf(tag + ".da", (M, c_))
        if plan.saved[tag + ".m2"].get("direct"):
            ops.conv3_c64_fwd(dza3, wT[pname + "m.0.cv2.conv.weight"], da, M // (H * W), H, W, flip=True)
        else:
            segs = [SegSpec(dza3, c_, 0, -dy, -dx, 1, 0, H, W) for (dy, dx) in TAPS3]
            ops.gemm_nt(segs, wT[pname + "m.0.cv2.conv.weight"], da, M, c_, 9 * c_, spatial=(H, W))
        dza2 = self._conv_bwd(plan, tag + ".m1", da, c_, 0)
        ops.gemm_nt([SegSpec(dza2)], wT[pname + "m.0.cv1.conv.weight"], da, M, c_, c_)
        dza1 = self._conv_bwd(plan, tag + ".cv1", da, c_, 0)
        ops.gemm_nt([SegSpec(dza1)], wT[pname + "cv1.conv.weight"], din, M, c1, c_, resid=din)
        return din

    # ------------------------------------------------------------------ SPP (common.py:129-140)
    def _spp_fwd(self, plan, P, tag, pname, segs, spatial, M, c1, c2, B):
        """cv1 (1x1 Conv+BN+SiLU) -> MaxPool 5 / 9 / 13 (stride 1, same padding: ONE 5x5 kernel in cascade, csrc/pool.hip) ->
        Concat -> cv2.  The concat never exists as a copy: cv1 and the pools write their slices of one [M][4 c_] buffer,
        which is cv2's single K-segment."""
        c_ = c1 // 2
        H, W = spatial
        cat = plan.buf(tag + ".cat", (M, 4 * c_))
        self._conv_fwd(plan, P, tag + ".cv1", pname + "cv1.", segs, spatial if any(s_.shr or s_.Hi for s_ in segs) else None, M, c1, c_, 1, out=cat)
        arg = plan.buf(tag + ".arg", (3, M, c_), torch.uint8) if plan.training else None
        for i in range(3):                                   # m5, m9 = m5 o m5, m13 = m5 o m5 o m5 into the next slices
            ops.maxpool5_fwd(cat, cat, arg[i] if arg is not None else None, B, H, W, c_, ldx=4 * c_, ldy=4 * c_, x_off=i * c_,
                             y_off=(i + 1) * c_)
        out = self._conv_fwd(plan, P, tag + ".cv2", pname + "cv2.", [SegSpec(cat)], None, M, 4 * c_, c2, 1)
        plan.saved[tag] = dict(M=M, c1=c1, c2=c2, spatial=spatial, pname=pname, B=B)
        return out

    def _spp_bwd(self, plan, P, tag, dout, ldd, d_off):
        """returns d(input) [M][c1]."""
        wT, b = P["wT"], plan.bufs
        sv = plan.saved[tag]
        M, c1, c2, (H, W), pname, B = sv["M"], sv["c1"], sv["c2"], sv["spatial"], sv["pname"], sv["B"]
        c_ = c1 // 2
        dz2 = self._conv_bwd(plan, tag + ".cv2", dout, ldd, d_off)
        dcat = plan.buf(tag + ".dcat", (M, 4 * c_))
        ops.gemm_nt([SegSpec(dz2)], wT[pname + "cv2.conv.weight"], dcat, M, 4 * c_, c2)
        for i in (2, 1, 0):                                  # d m13 -> m9 -> m5 -> a1, accumulated into the lower slice
            ops.maxpool5_bwd(dcat, b[tag + ".arg"][i], dcat, B, H, W, c_, lddy=4 * c_, lddx=4 * c_, dy_off=(i + 1) * c_, dx_off=i * c_,
                             accumulate=True)
        dz1 = self._conv_bwd(plan, tag + ".cv1", dcat, 4 * c_, 0)
        din = plan.buf(tag + ".din", (M, c1))
        ops.gemm_nt([SegSpec(dz1)], wT[pname + "cv1.conv.weight"], din, M, c1, c_)
        return din

    # ================================================================== super-resolution branch (model.py:284-287)
    def _ref_buf(self, plan, ref):
        return plan.bufs[f"f{ref[1]}"] if ref[0] == "enc" else plan.bufs[self._unit_out_name(self._unit(ref[1]))]

    def _sr_forward(self, plan: Plan):
        """output_sr = model_up(low-level, deep) on the tapped features: live launches (sr.SRBranch; its weights are re-laid out
        from the masters every call).  The taps are K-segment views of the head's buffers - an Upsample / Concat entry is index
        arithmetic here too."""
        from .sr import SRBranch
        B, t = plan.B, plan.S // 4
        br = getattr(plan, "sr", None)
        if br is None:
            names = [n for n in self.params if n.startswith("model_up.")]
            br = plan.sr = SRBranch({n: self.params[n] for n in names}, plan.dt, {n: self.g[n] for n in names},
                                    dec="model_up.sr_decoder.", edsr="model_up.edsr.")
            br._prep_version = (self.params_version(), self._param_epoch)
        else:
            # re-lay the 82 SR tensors out only when a parameter may have changed: in-place writes through the Parameter objects
            # (version counters) or the fused optimizer / invalidate_params() (epoch) - not on every forward of an accumulation step
            ver = (self.params_version(), self._param_epoch)
            if getattr(br, "_prep_version", None) != ver:
                br.prepare()
                br._prep_version = ver
        def segs(parts, level):
            h = t >> level
            return [SegSpec(self._ref_buf(plan, ref), c, 0, 0, 0, 1, shr, h >> shr, h >> shr) for ref, c, shr in parts]
        return br.forward(segs(self.sr_parts[0], 0), segs(self.sr_parts[1], 1), B, t, t)

    def _sr_backward(self, plan: Plan, dsr):
        """SR gradients: parameters into the flat buffer, inputs into plan.sr's dense buffers (zero when the SR output took no part
        in the loss); the recorded head backward adds them to the tapped features' gradients (_backward_main: sr_extra)."""
        br = plan.sr
        if dsr is None:
            for k in ("g.low", "g.x"):
                if k in br.bufs:
                    ops.zero_(br.bufs[k])
            return
        br.backward(dsr)

    def _sr_extra(self, plan: Plan):
        """ref -> [(dense gradient buffer, ld, column offset, channels, shr)]: what the SR branch adds to d(feature)."""
        B, t = plan.B, plan.S // 4
        br = plan.sr
        c1 = sum(c for _, c, _ in self.sr_parts[0])
        c2 = sum(c for _, c, _ in self.sr_parts[1])
        d_low = br._buf("g.low", (B * t * t, c1))
        d_x = br._buf("g.x", (B * (t // 2) ** 2, c2))
        extra: Dict[tuple, list] = {}
        for parts, buf, level in ((self.sr_parts[0], d_low, 0), (self.sr_parts[1], d_x, 1)):
            coff = 0
            for ref, c, shr in parts:
                extra.setdefault(ref, []).append((buf, buf.shape[1], coff, c, shr, t >> level))
                coff += c
        return extra

    def _add_extra(self, plan: Plan, extra, ref, target):
        """target (buffer, ld, column offset) += the SR branch's gradient for feature `ref` (summed over the 2^shr x 2^shr children
        where the tap was an upsampled view)."""
        B = plan.B
        dst, ld, off = target
        for i, (buf, lds, coff, c, shr, H) in enumerate(extra.get(ref, ())):
            if shr == 0:
                ops.add_rows(dst, buf, B * H * H, c, ldd=ld, dcol=off, lds=lds, scol=coff)
            else:
                Hs = H >> shr
                tmp = plan.buf(f"sr.dup.{ref[0]}{ref[1]}.{i}", (B * Hs * Hs, c))
                ops.gather_sum_rows(buf, lds, tmp, c, B, Hs, Hs, shr, c, d_off=coff)
                ops.add_rows(dst, tmp, B * Hs * Hs, c, ldd=ld, dcol=off)

    # ================================================================== backward
    def _backward(self, plan: Plan, x_rgb, x_ir, dpred, dsr=None):
        if self.fused or not plan.training:
            raise RuntimeError("backward needs a training-mode, un-fused model")
        P = self._prep_for(plan.dt)
        B, S = plan.B, plan.S
        t = S // 4
        T1 = B * t * t
        fresh = self._claim_grads()
        if fresh:
            ops.zero_(self.flat_grad)
        if self.ddp is not None:
            self.ddp.begin_backward(self.flat_grad, fresh)
        # (1) Detect backward: live (dpred pointer changes)
        dzd = plan.buf("g.dzd", (T1, self.det_np))
        if dpred is None:                            # SR-only loss
            ops.zero_(dzd)
        else:
            ops.detect_unpermute(dpred, dzd, self.det_np, B, t * t, self.na, self.no)
        if self.sr:
            self._sr_backward(plan, dsr)             # live too: its input gradients land in plan.sr's buffers
        overlap = False
        if plan.bwd_main is None:
            with ops.Recorder() as rec:
                self._backward_main(plan, P)
            plan.bwd_main = rec.calls
        elif self.ddp is not None and self.ddp.world > 1 and self.ddp.sync and plan.bwd_marks:
            # replay in segments: at every mark one more bucket of the flat buffer is final and its all-reduce starts
            overlap = True
            pos = 0
            for idx, lo, hi in plan.bwd_marks:
                ops.replay(plan.bwd_main, probes=self.probes_bwd, start=pos, end=idx)
                self.ddp.reduce_async(self.flat_grad[lo:hi])
                pos = idx
            ops.replay(plan.bwd_main, probes=self.probes_bwd, start=pos)
        else:
            self._replay(plan, "bwd_main", plan.bwd_main, self.probes_bwd)
        self._frontend_bwd(plan, P, x_rgb, x_ir)
        if self.ddp is not None:
            if overlap:
                self.ddp.finish(self.flat_grad[:self.ddp_split])
            else:
                self.ddp.reduce(self.flat_grad)

    def replay_encoder_backward(self, plan: Plan, x_rgb, x_ir):
        """Diagnostic / test hook: re-run the ENCODER part of the recorded backward (necks, stages 3..1, PatchMergings, patch
        embed, front end) from whatever the head-input gradient buffers `plan.enc_gin` hold now, accumulating into the flat
        gradient buffer.  The encoder has no cross-image coupling (LayerNorm, windows, per-image shifts), so a batch's encoder
        gradients are the sum of its images' - tests/test_fullsize_gpu.py checks the B = 8 training step that way.  Needs one
        ordinary backward on this plan first (it records the launches) and the activations of the latest forward."""
        if plan.bwd_main is None or getattr(plan, "enc_bwd_start", None) is None:
            raise RuntimeError("replay_encoder_backward: run one backward on this plan first")
        ops.replay(plan.bwd_main, start=plan.enc_bwd_start)
        self._frontend_bwd(plan, self._prep_for(plan.dt), x_rgb, x_ir)

    def _frontend_bwd(self, plan: Plan, P, x_rgb, x_ir):
        # (3) front end: live
        B, S = plan.B, plan.S
        t = S // 4
        fe = P["fe"]
        cb = self.model.image_encoder.chan_block
        ca_ws, ca_shift = int(cb.window_size), int(cb.shift_size)
        if ca_ws == 1:
            ws = plan.buf("fe.ws", (ops.frontend_bwd_workspace_bytes(B, S) // 4,), torch.float32)
            ops.frontend_bwd(x_rgb, x_ir, x_ir.shape[1] * S * S, fe["w"], fe["b"], fe["g"], fe["be"], plan.bufs["g.dx0"],
                             self.g_fe_w, self.g_fe_b, self.g_fe_g, self.g_fe_be, B, S, 1, ws)
        else:
            de = plan.buf("fe.de", (B * t * t, 192), torch.float32)
            ops.zero_(de)
            ops.cross_attn_ln_bwd(plan.bufs["fe.e"], fe["g"], plan.bufs["g.dx0"], de, self.g_fe_g, self.g_fe_be, B, S, ca_ws, ca_shift)
            ops.patch_embed4_bwd(x_rgb, x_ir, x_ir.shape[1] * S * S, de, self.g_fe_w, self.g_fe_b, B, S)

    def _backward_main(self, plan: Plan, P):
        B, S = plan.B, plan.S
        t = S // 4
        wT, g, b = P["wT"], self.g, plan.bufs
        enc = self.model.image_encoder
        ops.zero_(plan.zpool("b"))                   # BN-backward reduction accumulators of every head conv
        T1, T2, T3 = B * t * t, B * (t // 2) ** 2, B * (t // 4) ** 2
        h2, h4 = t // 2, t // 4
        # ---- Detect
        dzd = b["g.dzd"]
        hu = self._unit(self.head_out[0])
        cd = self.head_out[1]
        ops.gemm_tn(dzd, [SegSpec(b[self._unit_out_name(hu)])], g[self.det_name + "m.0.weight"], T1, self.na * self.no, cd, ldy=self.det_np, lddw=cd,
                    dbias=g[self.det_name + "m.0.bias"])
        dyd = plan.buf("g.dyd", (T1, cd))
        ops.gemm_nt([SegSpec(dzd)], wT[self.det_name + "m.0.weight"], dyd, T1, cd, self.det_np)
        # ---- head units in reverse: every unit returns d(its concatenated input) [M][c1]; the gradient of an Upsample /
        #      Concat input is a column slice of it (nearest x2^shr upsample: summed over the 2^shr x 2^shr children)
        gout = {("unit", hu["k"]): (dyd, cd, 0)}                     # ref -> (buffer, ld, column offset)
        extra = self._sr_extra(plan) if self.sr else {}
        for u in reversed(self.head_units):
            H = t >> u["level"]
            M = B * H * H
            tag = f"h{u['k']}"
            dy, ld, off = gout.pop(("unit", u["k"]))
            self._add_extra(plan, extra, ("unit", u["k"]), (dy, ld, off))
            if u["kind"] == "Conv":
                dz = self._conv_bwd(plan, tag, dy, ld, off)
                din = plan.buf(tag + ".din", (M, u["c1"]))
                w_ = wT[f"detect.{u['k']}.conv.weight"]
                if u["ksize"] == 1:
                    ops.gemm_nt([SegSpec(dz)], w_, din, M, u["c1"], u["c2"])
                else:
                    segs = [SegSpec(dz, u["c2"], 0, -dy_, -dx_, 1, 0, H, H) for (dy_, dx_) in TAPS3]
                    ops.gemm_nt(segs, w_, din, M, u["c1"], 9 * u["c2"], spatial=(H, H))
            elif u["kind"] == "C3":
                din = self._c3_bwd(plan, P, tag, dy, ld, off)
            else:
                din = self._spp_bwd(plan, P, tag, dy, ld, off)
            coff = 0
            for ref, c, shr in u["parts"]:
                if shr == 0:
                    gout[ref] = (din, u["c1"], coff)
                else:
                    Hs = H >> shr
                    dsrc = plan.buf(f"{tag}.dup{coff}", (B * Hs * Hs, c))
                    ops.gather_sum_rows(din, u["c1"], dsrc, c, B, Hs, Hs, shr, c, d_off=coff)
                    gout[ref] = (dsrc, c, 0)
                coff += c
        for j in range(3):
            self._add_extra(plan, extra, ("enc", j), gout[("enc", j)])
        (gf0, ld0, off0), (gf1, ld1, off1), (gf2, ld2, off2) = gout[("enc", 0)], gout[("enc", 1)], gout[("enc", 2)]
        # where the head's backward ends and the encoder's begins, and the buffers the head left d(f0), d(f1), d(f2) in
        # (token-major rows, `ld` columns, the feature's columns at `off`): replay_encoder_backward re-runs the rest from there
        plan.enc_bwd_start = ops.recorded_count()
        plan.enc_gin = [(gf0, ld0, off0, 256), (gf1, ld1, off1, 256), (gf2, ld2, off2, 512)]
        # ---- neck3 + stage 3
        s3out = b["stage3.0.xo"]
        ops.gemm_tn(gf2, [SegSpec(s3out)], g[E + "neck3.weight"], T3, 512, 768, ldy=ld2, y_off=off2)
        d3a = plan.buf("g.dA.768", (T3, 768))
        d3b = plan.buf("g.dB.768", (T3, 768))
        ops.gemm_nt([SegSpec(gf2, 512, off2)], wT[E + "neck3.weight"], d3a, T3, 768, 512)
        self._block_bwd(plan, P, "stage3.0", enc.stage3[0], d3a, d3b)
        # head, neck 3 and stage 3 gradients are final: first data-parallel bucket (ddp.GradReducer.reduce_async)
        plan.bwd_marks.append((ops.recorded_count(), self.ddp_split3, self.flat_grad.numel()))
        # ---- pmerging2 -> d(stage2 out) ; + neck2
        dA = plan.buf("g.dA.384", (T2, 384))
        dB = plan.buf("g.dB.384", (T2, 384))
        self._merge_bwd(plan, P, "pmerging2", d3b, dA)
        s2out = b["stage2.3.xo"]
        ops.gemm_tn(gf1, [SegSpec(s2out)], g[E + "neck2.weight"], T2, 256, 384, ldy=ld1, y_off=off1)
        ops.gemm_nt([SegSpec(gf1, 256, off1)], wT[E + "neck2.weight"], dA, T2, 384, 256, resid=dA)
        cur, other = dA, dB
        for i in reversed(range(4)):
            self._block_bwd(plan, P, f"stage2.{i}", enc.stage2[i], cur, other)
            cur, other = other, cur
        # ---- pmerging1 -> d(stage1 out5) ; + neck1
        dA = plan.buf("g.dA.192", (T1, 192))
        dB = plan.buf("g.dB.192", (T1, 192))
        self._merge_bwd(plan, P, "pmerging1", cur, dA)
        o4, o5 = b["stage1.4.xo"], b["stage1.5.xo"]
        ops.gemm_tn(gf0, [SegSpec(o4), SegSpec(o5)], g[E + "neck1.weight"], T1, 256, 384, ldy=ld0, y_off=off0)
        df0 = SegSpec(gf0, 256, off0)
        ops.gemm_nt([df0], wT[E + "neck1.weight"], dA, T1, 192, 256, w_off=192 * 256, resid=dA)        # d out5 += df0 @ Wn1[:,192:]
        # every gradient from PatchMerging 1 to the end of the flat buffer is final here: data-parallel runs start
        # their all-reduce now, under the stage-1 backward (ddp.GradReducer.reduce_async)
        plan.bwd_marks.append((ops.recorded_count(), self.ddp_split, self.ddp_split3))
        self._block_bwd(plan, P, "stage1.5", enc.stage1[5], dA, dB)
        ops.gemm_nt([df0], wT[E + "neck1.weight"], dB, T1, 192, 256, resid=dB)                           # d out4 += df0 @ Wn1[:,:192]
        cur, other = dB, dA
        for i in reversed(range(5)):
            self._block_bwd(plan, P, f"stage1.{i}", enc.stage1[i], cur, other)
            cur, other = other, cur
        # ---- patch_embed 1x1 + pos_embed
        if plan.saved["use_pos"]:
            ops.batch_sum(cur, g[E + "pos_embed"], B, t * t * 192)
        ops.gemm_tn(cur, [SegSpec(b["x0"])], g[E + "patch_embed.proj.weight"], T1, 192, 192, dbias=g[E + "patch_embed.proj.bias"])
        dx0 = plan.buf("g.dx0", (T1, 192))
        ops.gemm_nt([SegSpec(cur)], wT[E + "patch_embed.proj.weight"], dx0, T1, 192, 192)
