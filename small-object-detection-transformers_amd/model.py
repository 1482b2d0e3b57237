"""Drop-in boundary: the reference's ``Model(cfg=...)`` / ``forward(x_rgb, x_ir)`` API
(basics/models/model.py:73-348) over the MI355X HIP engine.

The module tree below only *holds parameters* under the reference's names and shapes
(273 state_dict entries) so that reference checkpoints load and
``ComputeLoss`` / ``ModelEMA`` / ``check_anchors`` keep working; none of these
containers computes anything in torch -- ``Model.forward`` hands the whole graph to
``engine.Engine`` (hand-written HIP forward and backward).  Calling a container's own
``forward`` raises: there is no torch fallback.

Differences from the reference that are deliberate:
  * resolution is a parameter: stage resolutions are t, t/2, t/4 with t = S/4 instead of
    the literals (128,128)/(64,64)/(32,32) (backbone_vit.py:119,136,153);
  * ``cfg`` may name ``SRyolo_MF.yaml`` -- this package ships a config of that name whose
    backbone row is models/model.yaml:48 (the only backbone row the fork can parse) and
    whose head is the identical Conv/Up/Cat/C3/Conv/Up/Cat/C3/Detect graph.
"""
from __future__ import annotations

import math
import os
from copy import deepcopy
from pathlib import Path
from typing import List, Optional

import torch
import torch.nn as nn

NUM_HEADS = 12          # backbone_vit.py:19 (the yaml's "6" is discarded by model.py:423)
SHIFTS = (0, 2, 0, 2, 0, 2, 0, 2)   # backbone_vit.py:114
CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


class _Container(nn.Module):
    """Parameter holder; computation happens in the HIP engine only."""

    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError(f"{type(self).__name__} is a parameter container of the MI355X engine; "
                           "run it through Model.forward (there is no torch fallback)")


def make_divisible(x, divisor):   # basics/utils/general.py:180-182
    return math.ceil(x / divisor) * divisor


def _relative_position_index(ws: int) -> torch.Tensor:   # backbone_vit.py:940-951
    c = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _shift_mask(H: int, W: int, ws: int, shift: int) -> torch.Tensor:   # backbone_vit.py:1058-1077
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    ph, pw = (-H) % ws, (-W) % ws             # window_partition's zero padding (backbone_vit.py:632-639): pad tokens get region id 0
    if ph or pw:
        img = torch.nn.functional.pad(img, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    mw = img.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


# ------------------------------------------------------------------------- encoder
class PatchEmbed(_Container):          # backbone_vit.py:742-773
    def __init__(self, kernel_size, stride, padding, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class CAttentionBlock(_Container):     # backbone_vit.py:407-467 (CAttention itself has no parameters)
    def __init__(self, embedding_dim: int, num_heads: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.window_size = 1           # :438
        self.shift_size = 0
        self.num_heads = num_heads
        self.register_buffer("attn_mask", None)


class WindowAttention(_Container):     # backbone_vit.py:913-959
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, (window_size, window_size), num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", _relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class Mlp(_Container):                 # backbone_vit.py:863-882
    def __init__(self, in_features, hidden_features, linear_mlp=True):
        super().__init__()
        self.linear = linear_mlp
        if linear_mlp:
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.fc2 = nn.Linear(hidden_features, in_features)
        else:
            self.fc1 = nn.Linear(in_features, in_features)
            self.conv1 = nn.Conv2d(in_features, in_features, 2)
            self.fc2 = nn.Linear(in_features, in_features)


class SwinTransformerBlock(_Container):   # backbone_vit.py:1011-1082
    def __init__(self, dim, input_resolution, num_heads, window_size, shift_size, mlp_ratio=4.0, linear_mlp=True):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, tuple(input_resolution), num_heads
        self.window_size, self.shift_size = window_size, shift_size
        if min(self.input_resolution) <= self.window_size:     # :1042-1045
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, self.window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), linear_mlp)
        mask = None
        if self.shift_size > 0:
            H, W = self.input_resolution
            mask = _shift_mask(H, W, self.window_size, self.shift_size)
        self.register_buffer("attn_mask", mask)   # kept for state_dict parity; the kernel derives it from coordinates


class PatchMerging(_Container):        # backbone_vit.py:823-837
    def __init__(self, input_resolution, dim):
        super().__init__()
        self.input_resolution, self.dim = tuple(input_resolution), dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(2 * dim)


class ImageEncoderViT(_Container):     # backbone_vit.py:11-188
    def __init__(self, img_size=512, patch_size=4, embed_dim=192, in_chans=4, out_chans=256, window_size=0,
                 num_heads=NUM_HEADS, mlp_ratio=4.0):
        super().__init__()
        if embed_dim != 192:
            raise ValueError("embed_dim must be 192 = 4 x 48 (backbone_vit.py:55,73)")
        self.img_size = img_size
        t = img_size // 4
        self.patch_embed = PatchEmbed((1, 1), (1, 1), (0, 0), 192, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, t, t, embed_dim))
        self.channel_embed_r = PatchEmbed((patch_size, patch_size), (4, 4), (1, 1), 1, 48)   # default padding (1,1): :751
        self.channel_embed_g = PatchEmbed((patch_size, patch_size), (4, 4), (0, 0), 1, 48)
        self.channel_embed_b = PatchEmbed((patch_size, patch_size), (4, 4), (0, 0), 1, 48)
        self.channel_embed_i = PatchEmbed((patch_size, patch_size), (4, 4), (0, 0), 1, 48)
        self.chan_block = CAttentionBlock(48, num_heads)
        self.stage1 = nn.ModuleList([
            SwinTransformerBlock(embed_dim, (t, t), num_heads, 8, SHIFTS[i], mlp_ratio, SHIFTS[i] == 0) for i in range(6)])
        self.pmerging1 = PatchMerging((t, t), embed_dim)
        self.stage2 = nn.ModuleList([
            SwinTransformerBlock(384, (t // 2, t // 2), num_heads, 8, SHIFTS[i], mlp_ratio, SHIFTS[i] == 0) for i in range(4)])
        self.pmerging2 = PatchMerging((t // 2, t // 2), 384)
        self.stage3 = nn.ModuleList([
            SwinTransformerBlock(768, (t // 4, t // 4), num_heads, 32, SHIFTS[i], mlp_ratio, True) for i in range(1)])
        self.neck3 = nn.Conv2d(768, 512, kernel_size=1, bias=False)
        self.neck2 = nn.Conv2d(384, 256, kernel_size=1, bias=False)
        self.neck1 = nn.Conv2d(384, 256, kernel_size=1, bias=False)


# ------------------------------------------------------------------------- head blocks
def autopad(k, p=None):     # common.py:26-30
    return k // 2 if p is None else p


class Conv(_Container):     # common.py:38-52
    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        if s != 1 or g != 1 or k not in (1, 3) or act is not True:
            raise NotImplementedError("HIP Conv supports k in {1,3}, stride 1, groups 1, SiLU (the head of model.yaml)")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU()

    def fuseforward(self, x):   # name kept for Model.fuse() parity
        raise RuntimeError("parameter container; run through Model.forward")


class Bottleneck(_Container):   # common.py:55-65
    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2


class C3(_Container):           # common.py:114-127
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])


class SPP(_Container):          # common.py:129-140; engine._spp_fwd/_spp_bwd (cascade of sodt_maxpool5 + K-segment concat)
    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        if tuple(k) != (5, 9, 13):
            raise NotImplementedError("SPP pools (5, 9, 13): the cascade of three 5x5 pools (csrc/pool.hip)")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])


class Concat(_Container):       # common.py:275-282
    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension


class Upsample(_Container):     # nn.Upsample(None, 2, 'nearest') of model.yaml:66,71
    def __init__(self, size=None, scale_factor=2, mode="nearest"):
        super().__init__()
        if size is not None or scale_factor != 2 or mode != "nearest":
            raise NotImplementedError("only nn.Upsample(None, 2, 'nearest')")
        self.scale_factor, self.mode = scale_factor, mode


class Detect(_Container):       # model.py:32-70
    stride = None
    export = False

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)


def check_anchor_order(m):      # basics/utils/autoanchor.py:13-21
    a = m.anchor_grid.prod(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da.sign() != ds.sign():
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


_MODULES = {"Conv": Conv, "C3": C3, "Concat": Concat, "Detect": Detect, "nn.Upsample": Upsample,
            "ImageEncoderViT": ImageEncoderViT, "Bottleneck": Bottleneck, "SPP": SPP}


def parse_model(d: dict, string: str, ch: List[int]):
    """yaml -> module graph; follows basics/models/model.py:350-435 for the rows model.yaml uses."""
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    rows = d[string]
    if string == "head":
        ch[0] = 256
        ch.append(256)
        ch.append(512)      # model.py:367-370
    for i, (f, n, m, args) in enumerate(rows):
        args = list(args)
        if m not in _MODULES:
            raise NotImplementedError(f"module {m!r} is outside the hot path this package builds (SURVEY.md section 8)")
        cls = _MODULES[m]
        for j, a in enumerate(args):
            if isinstance(a, str):
                args[j] = {"nc": nc, "anchors": anchors, "None": None, "False": False, "True": True}.get(a, a)
        n = max(round(n * gd), 1) if n > 1 else n
        if cls in (Conv, C3, Bottleneck, SPP):      # model.py:378-381
            c1, c2 = ch[f], args[0]
            c2 = make_divisible(c2 * gw, 8) if c2 != no else c2
            args = [c1, c2, *args[1:]]
            if cls is C3:
                args.insert(2, n)
                n = 1
        elif cls is Concat:
            c2 = sum(ch[x] for x in f)
        elif cls is Detect:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        else:
            c2 = ch[f if f < 0 else f + 1] if not isinstance(f, list) else c2
        if string == "backbone":
            if cls is not ImageEncoderViT or len(args) < 6:
                raise NotImplementedError("backbone row must be [-1, 1, ImageEncoderViT, [img, -, 192, 4, 256, -]] (models/model.yaml:48)")
            m_ = ImageEncoderViT(img_size=args[0], patch_size=4, embed_dim=args[2], in_chans=args[3], out_chans=args[4],
                                 window_size=args[5])
        else:
            if n != 1:
                raise NotImplementedError("repeated head modules")
            m_ = cls(*args)
        m_.i, m_.f, m_.type = i, f, m
        m_.np = sum(x.numel() for x in m_.parameters())
        save.extend(x % (i + 0.00001) for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        ch.append(c2)
    if string == "backbone":
        return layers[0], sorted(save)
    return nn.Sequential(*layers), sorted(save)


class NMS(nn.Module):
    """common.py:285-295: non_max_suppression as a module (thresholds are class attributes, as in the reference)."""
    conf = 0.25
    iou = 0.45
    classes = None

    def forward(self, x):
        from .nms import non_max_suppression          # HIP path (csrc/nms.hip); raises on CPU tensors
        return non_max_suppression(x[0], conf_thres=self.conf, iou_thres=self.iou, classes=self.classes)


class Model(nn.Module):
    """basics/models/model.py:73-348 -- same constructor, forward signature, return tuples and attributes."""
    export = False

    def __init__(self, cfg="model.yaml", input_mode="RGB", ch_steam=3, ch=3, nc=None, anchors=None, config=None,
                 sr=False, factor=2):
        super().__init__()
        self.init_params = dict(cfg=cfg, input_mode=input_mode, ch_steam=ch_steam, ch=ch, nc=nc, anchors=anchors,
                                config=config, sr=sr, factor=factor)
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            import yaml
            path = cfg if os.path.exists(cfg) else os.path.join(CONFIG_DIR, os.path.basename(cfg))
            self.yaml_file = Path(path).name
            with open(path) as f:
                self.yaml = yaml.load(f, Loader=yaml.SafeLoader)
        self.sr = bool(sr)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        if anchors:
            self.yaml["anchors"] = round(anchors)
        self.image_encoder, self.save1 = parse_model(deepcopy(self.yaml), "backbone", ch=[ch])
        self.detect, self.save2 = parse_model(deepcopy(self.yaml), "head", ch=[ch])
        if self.sr:
            # model.py:109-117.  The reference cannot get here (it imports `models.deeplabedsr`, which does not exist, and its
            # l1 / l2 = 4 / 8 would feed 256 channels into the 128-channel conv1: SURVEY.md section 8, config reality row 5);
            # this builds what the constructor describes - DeepLab(3 or 4, c1, c2, factor) - and the engine taps the first
            # feature-list entries that HAVE c1 channels on the stride-4 grid and c2 channels on the stride-8 grid (y[8] and
            # y[5] in models/model.yaml).  An interpretation: graph parity unpinned; the modules themselves are pinned
            # (tests/golden/sr.pt).
            from .sr import DeepLab
            if factor != 2:
                raise NotImplementedError("the super-resolution branch is built for factor=2 (Train.py:98 down_factor default)")
            self.model_up = DeepLab(3 if input_mode in ("IR", "RGB") else 4, self.yaml["c1"], self.yaml["c2"], factor=factor)
            self.l1, self.l2 = self.yaml["l1"], self.yaml["l2"]
        m = self.detect[-1]
        if isinstance(m, Detect):
            m.stride = torch.tensor([4.])                       # model.py:130
            m.anchors /= m.stride.view(-1, 1, 1)
            check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        for mod in self.modules():                              # initialize_weights, torch_utils.py:145-154
            if type(mod) is nn.BatchNorm2d:
                mod.eps = 1e-3
                mod.momentum = 0.03
        self.compute_dtype: Optional[torch.dtype] = None        # None: bf16 under autocast, else f32
        self.materialize_features = False
        self._engine = None
        self._nms: Optional[NMS] = None      # set by .nms(); kept outside self.model so state_dict keys do not move

    def load_state_dict(self, state_dict, *args, **kwargs):
        # resume / best weights rewrite the f32 masters: the engine's run-dtype mirror must be re-cast at the next forward
        res = super().load_state_dict(state_dict, *args, **kwargs)
        if getattr(self, "_engine", None) is not None:
            self._engine.invalidate_params()
        return res

    # ------------------------------------------------------------------ reference helpers
    def _initialize_biases(self, cf=None):      # model.py:299-307
        m = self.detect[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):                              # model.py:317-325 / torch_utils.py:182-203
        for m in self.modules():
            if type(m) is Conv and hasattr(m, "bn"):
                conv, bn = m.conv, m.bn
                fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                                  bias=True).requires_grad_(False).to(conv.weight.device)
                w_bn = bn.weight.div(torch.sqrt(bn.eps + bn.running_var))
                fused.weight.copy_(conv.weight * w_bn.view(-1, 1, 1, 1))
                fused.bias.copy_(bn.bias - bn.weight.mul(bn.running_mean).div(torch.sqrt(bn.running_var + bn.eps)))
                m.conv = fused
                delattr(m, "bn")
        self._engine = None
        return self

    def nms(self, mode=True):                    # model.py:327-339: add or remove the NMS stage (eval only)
        if mode and self._nms is None:
            self._nms = NMS()
            self.eval()
        elif not mode:
            self._nms = None
        return self

    def info(self, verbose=False, img_size=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_g = sum(x.numel() for x in self.parameters() if x.requires_grad)
        print(f"Model Summary: {len(list(self.modules()))} layers, {n_p} parameters, {n_g} gradients")

    def __deepcopy__(self, memo):               # ModelEMA (torch_utils.py:283): copy parameters, not the engine
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            # the engine and the data-parallel reducer (ddp.attach: process-group handles) belong to the training model only
            new.__dict__[k] = None if k in ("_engine", "grad_reducer", "_pending_ddp") else deepcopy(v, memo)
        return new

    def __getstate__(self):                     # checkpoints pickle the module object (Train.py:531-532)
        d = self.__dict__.copy()
        d["_engine"] = None
        return d

    # ------------------------------------------------------------------ forward
    def _get_engine(self):
        if self._engine is None:
            from .engine import Engine          # imports the HIP library; raises if it is missing
            self._engine = Engine(self)
            if getattr(self, "_pending_ddp", None) is not None:
                self._engine.ddp = self._pending_ddp
        return self._engine

    def forward(self, x, ir=None, input_mode="RGB+IR", augment=False, profile=False):
        """model.py:151-211.  train -> ([pred], y);  eval -> (z, [pred], y)."""
        if input_mode != "RGB+IR":
            raise NotImplementedError("only input_mode='RGB+IR' (the 4-channel encoder of model.yaml) is built")
        if augment:
            raise NotImplementedError("TTA (augment=True) is outside the hot path")
        if ir is None:
            raise ValueError("ir is required")
        if not x.is_cuda:
            raise RuntimeError("the MI355X engine needs CUDA/HIP tensors: there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        eng = self._get_engine()
        dt = self.compute_dtype
        if dt is None:
            dt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        training = self.training or self.export
        pred, feats, out_sr = eng.run(x, ir, dt, training)
        if training:
            if self.sr:                                          # model.py:203-205
                return [pred], out_sr, feats + [[pred]]
            return [pred], feats + [[pred]]
        z = eng.decode(pred)
        if getattr(self, "_nms", None) is not None:
            # the reference's `return y[0], y[1], features` (model.py:211) indexes the NMS list and fails for B == 1;
            # here the detections take the place of z and the raw head output stays second
            return self._nms((z, [pred])), [pred], feats + [(z, [pred])]
        return z, [pred], feats + [(z, [pred])]
