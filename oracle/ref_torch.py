"""CPU ORACLE (test infrastructure, never the product path).

A plain-PyTorch fp32/fp64 restatement of the reference hot path
(`models/model.yaml` graph: cross-channel attention -> conv-enhanced Swin encoder
-> YOLOv5 C3 head -> Detect), written functionally over a ``state_dict`` that uses
the reference's own parameter names, and parameterised in resolution (the
reference hard-codes 512x512).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path
(``small-object-detection-transformers_amd``) never does: it fails loudly when the
HIP library is missing.

Pinning: ``oracle/gen_golden.py`` imports the reference itself (CPU, this
container only) and checks this restatement against it at 512x512 and on tiny
per-module instantiations; the resulting vectors live in ``tests/golden/``.

Every function cites the reference lines it follows (paths relative to
/root/reference/).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

# ----------------------------------------------------------------------------
# architecture constants of models/model.yaml as built by parse_model
# (basics/models/model.py:350-435, basics/models/backbone_vit.py:11-188)
# ----------------------------------------------------------------------------
NUM_HEADS = 12                      # backbone_vit.py:19 (yaml's 6 is discarded, model.py:423)
EMBED = 192                         # model.yaml:48 args[2]
CH_EMBED = 48                       # backbone_vit.py:73
STAGE_DIMS = (192, 384, 768)        # backbone_vit.py:118,135,152
STAGE_DEPTHS = (6, 4, 1)            # backbone_vit.py:116,133,150
STAGE_WINDOWS = (8, 8, 32)          # backbone_vit.py:121,138,155
SHIFTS = (0, 2, 0, 2, 0, 2, 0, 2)   # backbone_vit.py:114
LN_EPS = 1e-5                       # nn.LayerNorm default (Appendix B of SURVEY.md)
BN_EPS = 1e-3                       # utils/torch_utils.py:150
BN_MOMENTUM = 0.03                  # utils/torch_utils.py:151
ANCHORS_PX = ((10., 13.), (16., 30.), (33., 23.))   # models/model.yaml:8
DET_STRIDE = 4.0                    # basics/models/model.py:130


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def window_partition(x: Tensor, ws: int) -> Tensor:
    """backbone_vit.py:619-643: zero padding at the bottom / right when H or W is not a multiple of the window (the reference
    pads AFTER norm1, so a pad token enters the attention as the qkv bias; pinned by tests/golden/pad_block.pt)."""
    B, H, W, C = x.shape
    ph, pw = (-H) % ws, (-W) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_unpartition(w: Tensor, ws: int, H: int, W: int) -> Tensor:
    """backbone_vit.py:646-672 (the padding of window_partition is cropped away)."""
    Hp, Wp = H + (-H) % ws, W + (-W) % ws
    B = w.shape[0] // ((Hp // ws) * (Wp // ws))
    x = w.view(B, Hp // ws, Wp // ws, ws, ws, -1)
    x = x.permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous() if (Hp > H or Wp > W) else x


def shift_mask(H: int, W: int, ws: int, shift: int, dtype=torch.float32) -> Tensor:
    """SW-MSA mask, 0 / -100 (backbone_vit.py:1058-1077 and :441-459)."""
    img = torch.zeros((1, H, W, 1), dtype=dtype)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def relative_position_index(ws: int) -> Tensor:
    """backbone_vit.py:940-951."""
    ch = torch.arange(ws)
    cw = torch.arange(ws)
    coords = torch.stack(torch.meshgrid([ch, cw], indexing="ij"))
    cf = torch.flatten(coords, 1)
    rel = cf[:, :, None] - cf[:, None, :]
    rel = rel.permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w, b, LN_EPS)


# ----------------------------------------------------------------------------
# front end: channel embeds + cross-channel attention + 1x1 mix (+pos)
# ----------------------------------------------------------------------------
def channel_embeds(sd: SD, x4: Tensor, pfx: str = "image_encoder.") -> Tuple[Tensor, ...]:
    """get_channels + 4x PatchEmbed(1->48, k4, s4); R has padding 1, the rest 0
    (backbone_vit.py:195-199, :69-98, :751, :810-820).  Returns NHWC tensors."""
    outs = []
    for ci, name in enumerate("rgbi"):
        pad = 1 if name == "r" else 0
        y = F.conv2d(x4[:, ci:ci + 1], sd[f"{pfx}channel_embed_{name}.proj.weight"],
                     sd[f"{pfx}channel_embed_{name}.proj.bias"], stride=4, padding=pad)
        outs.append(y.permute(0, 2, 3, 1))
    return tuple(outs)


def cattention(q: Tensor, kv: Tensor, heads: int, mask: Optional[Tensor]) -> Tensor:
    """CAttention.forward (backbone_vit.py:589-616): no projections, mask added
    BEFORE the 1/sqrt(d) scale."""
    B_, N, C = q.shape
    d = C // heads

    def sep(t):
        return t.reshape(B_, N, heads, d).transpose(1, 2)
    qh, kh, vh = sep(q), sep(kv), sep(kv)
    attn = qh @ kh.permute(0, 1, 3, 2)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, heads, N, N)
    attn = torch.softmax(attn / math.sqrt(d), dim=-1)
    out = attn @ vh
    return out.transpose(1, 2).reshape(B_, N, C)


def cattention_block(sd: SD, r: Tensor, g: Tensor, b: Tensor, ir: Tensor,
                     window_size: int = 1, shift: int = 0,
                     pfx: str = "image_encoder.chan_block.") -> Tuple[Tensor, ...]:
    """CAttentionBlock.forward (backbone_vit.py:469-561).  The reference hard-codes
    window_size=1, shift 0 (:438, :100-103); the general form is kept so the HIP
    kernel's general path has an oracle."""
    B, H, W, C = r.shape
    ws = window_size
    mask = shift_mask(H, W, ws, shift, r.dtype) if shift > 0 else None

    def part(t):
        if shift > 0:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        return window_partition(t, ws).reshape(-1, ws * ws, C)

    def unpart(t):
        t = window_unpartition(t.view(-1, ws, ws, C), ws, H, W)
        if shift > 0:
            t = torch.roll(t, shifts=(shift, shift), dims=(1, 2))
        return t
    rw, gw, bw, iw = part(r), part(g), part(b), part(ir)
    # pairs (q, kv): (R,G), (G,B), (B,IR), (IR,G)  -- backbone_vit.py:508-521
    r_out = unpart(cattention(rw, gw, NUM_HEADS, mask))
    g_out = unpart(cattention(gw, bw, NUM_HEADS, mask))
    b_out = unpart(cattention(bw, iw, NUM_HEADS, mask))
    i_out = unpart(cattention(iw, gw, NUM_HEADS, mask))
    x1 = layer_norm(r + r_out, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"])
    x2 = layer_norm(g + g_out, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    x3 = layer_norm(b + b_out, sd[pfx + "norm3.weight"], sd[pfx + "norm3.bias"])
    x4 = layer_norm(ir + i_out, sd[pfx + "norm4.weight"], sd[pfx + "norm4.bias"])
    return x1, x2, x3, x4


def frontend(sd: SD, x4: Tensor, pfx: str = "image_encoder.", ca_window: int = 1, ca_shift: int = 0) -> Tensor:
    """backbone_vit.py:195-217 -> (B, t, t, 192) NHWC token grid."""
    r, g, b, i = channel_embeds(sd, x4, pfx)
    r, g, b, i = cattention_block(sd, r, g, b, i, ca_window, ca_shift, pfx + "chan_block.")
    x = torch.cat((r, g, b, i), dim=-1)                           # :210
    w = sd[pfx + "patch_embed.proj.weight"].view(EMBED, EMBED)    # 1x1 conv, :51-57
    x = x @ w.t() + sd[pfx + "patch_embed.proj.bias"]
    pe = sd.get(pfx + "pos_embed")
    if pe is not None and x.shape[1] == pe.shape[1]:              # :215-217 quirk
        x = x + pe
    return x


# ----------------------------------------------------------------------------
# Swin block
# ----------------------------------------------------------------------------
def window_attention(sd: SD, pfx: str, xw: Tensor, ws: int, mask: Optional[Tensor]) -> Tensor:
    """WindowAttention.forward (backbone_vit.py:961-992)."""
    B_, N, C = xw.shape
    hd = C // NUM_HEADS
    qkv = (xw @ sd[pfx + "qkv.weight"].t() + sd[pfx + "qkv.bias"])
    qkv = qkv.reshape(B_, N, 3, NUM_HEADS, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * (hd ** -0.5)
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws).view(-1)
    bias = sd[pfx + "relative_position_bias_table"][idx].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, NUM_HEADS, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, NUM_HEADS, N, N)
    attn = torch.softmax(attn, dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return x @ sd[pfx + "proj.weight"].t() + sd[pfx + "proj.bias"]


def mlp(sd: SD, pfx: str, x: Tensor, H: int, W: int, linear: bool) -> Tensor:
    """Mlp.forward (backbone_vit.py:884-908)."""
    if linear:
        h = F.gelu(x @ sd[pfx + "fc1.weight"].t() + sd[pfx + "fc1.bias"])
        return h @ sd[pfx + "fc2.weight"].t() + sd[pfx + "fc2.bias"]
    u = x @ sd[pfx + "fc1.weight"].t() + sd[pfx + "fc1.bias"]
    bs = u.shape[0]
    u = u.permute(0, 2, 1).contiguous().view(bs, -1, H, W)
    u = F.pad(u, (0, 1, 0, 1))                                     # :896 right/bottom zero pad
    u = F.conv2d(u, sd[pfx + "conv1.weight"], sd[pfx + "conv1.bias"])   # 2x2, :897
    u = u.permute(0, 2, 3, 1).contiguous().view(bs, H * W, -1)
    u = F.gelu(u)
    return u @ sd[pfx + "fc2.weight"].t() + sd[pfx + "fc2.bias"]


def swin_block(sd: SD, pfx: str, x: Tensor, H: int, W: int, window: int, shift: int, linear_mlp: bool) -> Tensor:
    """SwinTransformerBlock.forward (backbone_vit.py:1084-1130) incl. the window
    clamp of the ctor (:1042-1045)."""
    B, L, C = x.shape
    assert L == H * W
    ws = window
    if min(H, W) <= ws:
        shift = 0
        ws = min(H, W)
    shortcut = x
    xn = layer_norm(x, sd[pfx + "norm1.weight"], sd[pfx + "norm1.bias"]).view(B, H, W, C)
    mask = None
    if shift > 0:
        xn = torch.roll(xn, shifts=(-shift, -shift), dims=(1, 2))
        mask = shift_mask(H, W, ws, shift, x.dtype)
    xw = window_partition(xn, ws).view(-1, ws * ws, C)
    aw = window_attention(sd, pfx + "attn.", xw, ws, mask)
    xs = window_unpartition(aw.view(-1, ws, ws, C), ws, H, W)
    if shift > 0:
        xs = torch.roll(xs, shifts=(shift, shift), dims=(1, 2))
    x = shortcut + xs.view(B, H * W, C)
    xn2 = layer_norm(x, sd[pfx + "norm2.weight"], sd[pfx + "norm2.bias"])
    return x + mlp(sd, pfx + "mlp.", xn2, H, W, linear_mlp)


def patch_merging(sd: SD, pfx: str, x: Tensor, H: int, W: int) -> Tensor:
    """PatchMerging.forward (backbone_vit.py:839-860)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C) @ sd[pfx + "reduction.weight"].t()
    return layer_norm(x, sd[pfx + "norm.weight"], sd[pfx + "norm.bias"])


def image_encoder(sd: SD, x4: Tensor, pfx: str = "image_encoder.", taps: Optional[dict] = None,
                  ca_window: int = 1, ca_shift: int = 0) -> List[Tensor]:
    """ImageEncoderViT.forward (backbone_vit.py:190-272), resolution-parameterised:
    stage resolutions are t, t/2, t/4 with t = S/4."""
    x = frontend(sd, x4, pfx, ca_window, ca_shift)
    B, h, w, c = x.shape
    if taps is not None:
        taps["frontend"] = x
    x = x.view(B, h * w, c)
    z = []
    for i in range(STAGE_DEPTHS[0]):
        x = swin_block(sd, f"{pfx}stage1.{i}.", x, h, w, STAGE_WINDOWS[0], SHIFTS[i], SHIFTS[i] == 0)
        if taps is not None:
            taps[f"stage1.{i}"] = x
        if i in (4, 5):
            z.append(x.view(B, h, w, c))
    y0 = torch.cat(z, dim=-1)
    x = patch_merging(sd, pfx + "pmerging1.", x, h, w)
    h2, w2 = h // 2, w // 2
    for i in range(STAGE_DEPTHS[1]):
        x = swin_block(sd, f"{pfx}stage2.{i}.", x, h2, w2, STAGE_WINDOWS[1], SHIFTS[i], SHIFTS[i] == 0)
        if taps is not None:
            taps[f"stage2.{i}"] = x
    y1 = x.view(B, h2, w2, -1)
    x = patch_merging(sd, pfx + "pmerging2.", x, h2, w2)
    h3, w3 = h2 // 2, w2 // 2
    for i in range(STAGE_DEPTHS[2]):
        # stage 3 block is built with the default linear_mlp=True (backbone_vit.py:151-160)
        x = swin_block(sd, f"{pfx}stage3.{i}.", x, h3, w3, STAGE_WINDOWS[2], SHIFTS[i], True)
        if taps is not None:
            taps[f"stage3.{i}"] = x
    y2 = x.view(B, h3, w3, -1)

    def neck(t, name):   # bias-free 1x1 conv, NHWC -> NCHW (:268-270)
        wgt = sd[pfx + name + ".weight"]
        return (t @ wgt.view(wgt.shape[0], -1).t()).permute(0, 3, 1, 2).contiguous()
    return [neck(y0, "neck1"), neck(y1, "neck2"), neck(y2, "neck3")]


# ----------------------------------------------------------------------------
# head (YOLOv5 blocks, NCHW)
# ----------------------------------------------------------------------------
def conv_bn_silu(sd: SD, pfx: str, x: Tensor, training: bool, new_stats: Optional[dict]) -> Tensor:
    """Conv.forward / fuseforward (common.py:38-52): Conv2d(bias=False)+BN(eps 1e-3,
    momentum 0.03)+SiLU; k in {1,3}, autopad."""
    w = sd[pfx + "conv.weight"]
    k = w.shape[-1]
    z = F.conv2d(x, w, None, stride=1, padding=k // 2)
    if pfx + "bn.weight" not in sd:                       # fused model (model.py:317-325)
        z = z + sd[pfx + "conv.bias"].view(1, -1, 1, 1)
        return F.silu(z)
    rm, rv = sd[pfx + "bn.running_mean"], sd[pfx + "bn.running_var"]
    if training:
        mean = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        if new_stats is not None:
            n = z.numel() / z.shape[1]
            new_stats[pfx + "bn.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            new_stats[pfx + "bn.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * n / (n - 1)
    else:
        mean, var = rm, rv
    zn = (z - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
    zn = zn * sd[pfx + "bn.weight"].view(1, -1, 1, 1) + sd[pfx + "bn.bias"].view(1, -1, 1, 1)
    return F.silu(zn)


def c3(sd: SD, pfx: str, x: Tensor, training: bool, new_stats: Optional[dict]) -> Tensor:
    """C3.forward with n=1 Bottleneck(shortcut=False) (common.py:114-127, :55-65)."""
    a = conv_bn_silu(sd, pfx + "cv1.", x, training, new_stats)
    a = conv_bn_silu(sd, pfx + "m.0.cv1.", a, training, new_stats)
    a = conv_bn_silu(sd, pfx + "m.0.cv2.", a, training, new_stats)
    b = conv_bn_silu(sd, pfx + "cv2.", x, training, new_stats)
    return conv_bn_silu(sd, pfx + "cv3.", torch.cat((a, b), 1), training, new_stats)


def spp(sd: SD, pfx: str, x: Tensor, training: bool, new_stats: Optional[dict]) -> Tensor:
    """SPP.forward (common.py:129-140): cv1 -> MaxPool2d(5 | 9 | 13, stride 1, same padding) -> cat -> cv2."""
    a = conv_bn_silu(sd, pfx + "cv1.", x, training, new_stats)
    pools = [F.max_pool2d(a, k, 1, k // 2) for k in (5, 9, 13)]
    return conv_bn_silu(sd, pfx + "cv2.", torch.cat([a] + pools, 1), training, new_stats)


def detect_raw(sd: SD, pfx: str, x: Tensor, na: int = 3) -> Tensor:
    """Detect.forward train branch (model.py:48-55)."""
    w = sd[pfx + "m.0.weight"]
    z = F.conv2d(x, w, sd[pfx + "m.0.bias"])
    bs, _, ny, nx = z.shape
    return z.view(bs, na, -1, ny, nx).permute(0, 1, 3, 4, 2).contiguous()


def detect_decode(raw: Tensor, anchor_grid: Tensor, stride: float = DET_STRIDE) -> Tensor:
    """Detect.forward eval branch (model.py:57-64)."""
    bs, na, ny, nx, no = raw.shape
    yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
    grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).to(raw)
    y = raw.sigmoid()
    xy = (y[..., 0:2] * 2. - 0.5 + grid) * stride
    wh = (y[..., 2:4] * 2) ** 2 * anchor_grid.view(1, na, 1, 1, 2).to(raw)
    return torch.cat((xy, wh, y[..., 4:]), -1).view(bs, -1, no)


def head(sd: SD, feats: Sequence[Tensor], training: bool, new_stats: Optional[dict] = None,
         pfx: str = "detect.") -> Tuple[Tensor, List[Tensor]]:
    """head graph models/model.yaml:65-74 as wired by Model.forward_once (model.py:268-281)."""
    y = list(feats)                                              # y[0..2]
    y.append(conv_bn_silu(sd, pfx + "0.", y[2], training, new_stats))          # y3
    y.append(F.interpolate(y[3], scale_factor=2, mode="nearest"))              # y4
    y.append(torch.cat((y[4], y[1]), 1))                                       # y5
    y.append(c3(sd, pfx + "3.", y[5], training, new_stats))                    # y6
    y.append(conv_bn_silu(sd, pfx + "4.", y[6], training, new_stats))          # y7
    y.append(F.interpolate(y[7], scale_factor=2, mode="nearest"))              # y8
    y.append(torch.cat((y[8], y[0]), 1))                                       # y9
    y.append(c3(sd, pfx + "7.", y[9], training, new_stats))                    # y10
    raw = detect_raw(sd, pfx + "8.", y[10])
    return raw, y


def head_graph(sd: SD, feats: Sequence[Tensor], rows, training: bool, new_stats: Optional[dict] = None,
               pfx: str = "detect.") -> Tuple[Tensor, List[Tensor]]:
    """Any head yaml (rows of [from, number, module, args], models/model.yaml:65-74) run as Model.forward_once runs the
    nn.Sequential parse_model built from it (model.py:268-281): the input of row i is y[f] (f = -1: the previous row), or
    the list of them; y[0..2] are the encoder outputs and row i appends y[3 + i].  Modules: Conv, C3, SPP, nn.Upsample,
    Concat, Detect (common.py:38-140,275-282; model.py:48-55)."""
    y = list(feats)
    x = None
    raw = None
    for i, (f, _n, m, _args) in enumerate(rows):
        if isinstance(f, int):
            xin = x if f == -1 else y[f]
        else:
            xin = [x if j == -1 else y[j] for j in f]
        p = f"{pfx}{i}."
        if m == "Conv":
            x = conv_bn_silu(sd, p, xin, training, new_stats)
        elif m == "C3":
            x = c3(sd, p, xin, training, new_stats)
        elif m == "SPP":
            x = spp(sd, p, xin, training, new_stats)
        elif m == "nn.Upsample":
            x = F.interpolate(xin, scale_factor=2, mode="nearest")
        elif m == "Concat":
            x = torch.cat(xin, 1)
        elif m == "Detect":
            raw = detect_raw(sd, p, xin[0])
            break
        else:
            raise NotImplementedError(m)
        y.append(x)
    return raw, y


def model_forward(sd: SD, x_rgb: Tensor, x_ir: Tensor, training: bool = True,
                  new_stats: Optional[dict] = None, taps: Optional[dict] = None, ca_window: int = 1, ca_shift: int = 0,
                  head_rows=None):
    """Model.forward, input_mode='RGB+IR' (model.py:191-192, :207-211, :245-294).
    train -> ([raw], y) ; eval -> (z, [raw], y).  head_rows: a head yaml other than models/model.yaml:65-74 (head_graph)."""
    x4 = torch.cat([x_rgb, x_ir[:, 0:1]], 1)
    feats = image_encoder(sd, x4, taps=taps, ca_window=ca_window, ca_shift=ca_shift)
    if head_rows is None:
        raw, y = head(sd, feats, training, new_stats)
        det = "detect.8."
    else:
        raw, y = head_graph(sd, feats, head_rows, training, new_stats)
        det = f"detect.{len(head_rows) - 1}."
    if training:
        return [raw], y + [[raw]]
    z = detect_decode(raw, sd[det + "anchor_grid"])
    return z, [raw], y + [(z, [raw])]


# ----------------------------------------------------------------------------
# super-resolution auxiliary branch (DeepLab = Decoder + EDSR x8), used in training with --super
# ----------------------------------------------------------------------------
def sr_decoder(sd: SD, pfx: str, x: Tensor, low: Tensor, factor: int = 2) -> Tensor:
    """Decoder.forward (basics/models/sr_decoder_noBN_noD.py:27-45): 1x1 convs (no bias) + ReLU on both inputs, bilinear
    (align_corners=True) resize of both to low's size x (factor // 2), concat (x first), 3x3 - ReLU - 3x3 - ReLU - 1x1(+bias)."""
    lo = F.relu(F.conv2d(low, sd[pfx + "conv1.weight"]))
    xx = F.relu(F.conv2d(x, sd[pfx + "conv2.weight"]))
    size = [d * (factor // 2) for d in lo.shape[2:]]
    xx = F.interpolate(xx, size=size, mode="bilinear", align_corners=True)
    if factor > 1:
        lo = F.interpolate(lo, size=size, mode="bilinear", align_corners=True)
    z = torch.cat((xx, lo), 1)
    z = F.relu(F.conv2d(z, sd[pfx + "last_conv.0.weight"], None, padding=1))
    z = F.relu(F.conv2d(z, sd[pfx + "last_conv.2.weight"], None, padding=1))
    return F.conv2d(z, sd[pfx + "last_conv.4.weight"], sd[pfx + "last_conv.4.bias"])


def edsr(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """EDSR.forward (basics/models/edsr.py:55-102): head 3x3; ResBlocks (3x3 - ReLU - 3x3, + input; res_scale 1) and a closing 3x3,
    + head output; tail: log2(scale) x (3x3 to 4 n_feat channels + PixelShuffle(2)), 3x3 to num_channels.  Depth / scale are read
    off the state dict (body.<i>.body.0 / tail.0.<2j>)."""
    def conv(name, t):
        return F.conv2d(t, sd[pfx + name + ".weight"], sd[pfx + name + ".bias"], padding=1)
    h = conv("head.0", x)
    r = h
    i = 0
    while f"{pfx}body.{i}.body.0.weight" in sd:
        r = r + conv(f"body.{i}.body.2", F.relu(conv(f"body.{i}.body.0", r)))
        i += 1
    r = conv(f"body.{i}", r) + h
    j = 0
    while f"{pfx}tail.0.{j}.weight" in sd:
        r = F.pixel_shuffle(conv(f"tail.0.{j}", r), 2)
        j += 2
    return conv("tail.1", r)


def deeplab_sr(sd: SD, pfx: str, low: Tensor, x: Tensor, factor: int = 2) -> Tensor:
    """DeepLab.forward (basics/models/deeplabedsr.py:61-73): EDSR(sr_decoder(x, low_level_feat, factor))."""
    return edsr(sd, pfx + "edsr.", sr_decoder(sd, pfx + "sr_decoder.", x, low, factor))


# ----------------------------------------------------------------------------
# NMS spec (greedy; torchvision.ops.nms semantics: suppress IoU > thr)
# basics/utils/general.py:425-512 with merge=False path as the pinned spec
# ----------------------------------------------------------------------------
def xywh2xyxy(x: Tensor) -> Tensor:
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def greedy_nms(boxes: Tensor, scores: Tensor, thr: float) -> Tensor:
    """Reference semantics of torchvision.ops.nms (general.py:496): sort by score
    descending (stable), keep a box unless IoU with an already kept box is > thr."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    suppressed = torch.zeros(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        xx1 = torch.maximum(b[i, 0], b[i + 1:, 0])
        yy1 = torch.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = torch.minimum(b[i, 2], b[i + 1:, 2])
        yy2 = torch.minimum(b[i, 3], b[i + 1:, 3])
        inter = (xx2 - xx1).clamp(0) * (yy2 - yy1).clamp(0)
        iou = inter / (area[i] + area[i + 1:] - inter)
        suppressed[i + 1:] |= iou > thr
    return order[torch.tensor(keep, dtype=torch.long)]


def xywh2xyxy(x: Tensor) -> Tensor:
    """general.py:270-277."""
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def box_iou(box1: Tensor, box2: Tensor) -> Tensor:
    """general.py:392-414."""
    def area(b):
        return (b[2] - b[0]) * (b[3] - b[1])
    a1, a2 = area(box1.T), area(box2.T)
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


def non_max_suppression(prediction: Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45, classes=None,
                        agnostic: bool = False, multi_label: bool = False, nms_fn=None, return_index: bool = False,
                        labels=()):
    """general.py:425-512 incl. the autolabelling rows (:451-458: `labels[xi]` = (nl, 5) [cls, x, y, w, h] appended
    after the confidence filter as rows with obj = 1 and a one-hot class; their row index is N + k), with torchvision.ops.nms restated
    as greedy_nms (stable descending score order).  The over-max_nms truncation uses a stable
    descending sort where the reference's argsort leaves tie order unspecified.
    return_index: also return, per image, the candidate id  row * nc + class  (row = index into the
    image's prediction rows) of every detection kept."""
    nms_fn = nms_fn or greedy_nms
    nc = prediction.shape[2] - 5
    xc = prediction[..., 4] > conf_thres
    max_wh, max_det, max_nms = 4096, 300, 30000
    multi_label = multi_label and nc > 1
    output = [torch.zeros((0, 6))] * prediction.shape[0]
    index = [torch.zeros((0,), dtype=torch.long)] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        rows = xc[xi].nonzero().view(-1)
        x = x[xc[xi]].clone()
        if labels and len(labels[xi]):
            l = labels[xi]
            v = torch.zeros((len(l), nc + 5))
            v[:, :4] = l[:, 1:5]
            v[:, 4] = 1.0
            v[range(len(l)), l[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
            rows = torch.cat((rows, prediction.shape[1] + torch.arange(len(l))))
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
            ids = rows[i] * nc + j
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            sel = conf.view(-1) > conf_thres
            x = torch.cat((box, conf, j.float()), 1)[sel]
            ids = (rows * nc + j.view(-1))[sel]
        if classes is not None:
            sel = (x[:, 5:6] == torch.tensor(classes)).any(1)
            x, ids = x[sel], ids[sel]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            o = torch.argsort(x[:, 4], descending=True, stable=True)[:max_nms]
            x, ids = x[o], ids[o]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms_fn(boxes, scores, iou_thres)
        if i.shape[0] > max_det:
            i = i[:max_det]
        if 1 < n < 3E3:
            iou = box_iou(boxes[i], boxes) > iou_thres
            weights = iou * scores[None]
            x[i, :4] = torch.mm(weights, x[:, :4]).float() / weights.sum(1, keepdim=True)
            i = i[iou.sum(1) > 1]
        output[xi] = x[i]
        index[xi] = ids[i]
    return (output, index) if return_index else output


def synthetic_predictions(B: int, N: int, nc: int, seed: int = 0, img: float = 1024.0, clusters: int = 40) -> Tensor:
    """Decoded-head-shaped (B, N, 5+nc) test rows: boxes scattered around `clusters` centres so
    that many overlap above the IoU threshold; objectness / class scores spread over (0, 1)."""
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(B, clusters, 2, generator=g) * img
    size = 8 + torch.rand(B, clusters, 2, generator=g) * 56
    which = torch.randint(0, clusters, (B, N), generator=g)
    bi = torch.arange(B)[:, None]
    xy = ctr[bi, which] + torch.randn(B, N, 2, generator=g) * 3.0
    wh = size[bi, which] * (0.8 + 0.4 * torch.rand(B, N, 2, generator=g))
    obj = torch.rand(B, N, 1, generator=g) ** 2
    cls = torch.rand(B, N, nc, generator=g) ** 3
    fav = torch.randint(0, nc, (B, clusters), generator=g)
    cls.scatter_(2, fav[bi, which][..., None], 0.5 + 0.5 * torch.rand(B, N, 1, generator=g))
    return torch.cat([xy, wh, obj, cls], 2).contiguous()


# ----------------------------------------------------------------------------
# parameter construction with the reference's names/shapes
# ----------------------------------------------------------------------------
# ----------------------------------------------------------------------------
# ComputeLoss (basics/utils/loss.py:90-224) with bbox_iou(CIoU) (basics/utils/general.py:347-389), functional
# ----------------------------------------------------------------------------
LOSS_HYP = dict(box=0.05, cls=0.5, cls_pw=1.0, obj=1.0, obj_pw=1.0, anchor_t=4.0, fl_gamma=0.0)   # models/hyp.scratch.yaml


def bbox_ciou(box1: Tensor, box2: Tensor, eps: float = 1e-7) -> Tensor:
    """general.py:347-389 with x1y1x2y2=False, CIoU=True.  box1 (4, n) xywh, box2 (n, 4) xywh."""
    box2 = box2.T
    b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
    b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
    b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
    b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * \
            (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def build_targets(pred: Tensor, targets: Tensor, anchors: Tensor, anchor_t: float = 4.0):
    """loss.py:165-224 for the single detection layer of model.yaml.  pred (B, na, ny, nx, no); targets (nt, 6) =
    (image, class, x, y, w, h) normalised; anchors (na, 2) in grid units.  Returns tcls, tbox, (b, a, gj, gi), anch."""
    na, nt = anchors.shape[0], targets.shape[0]
    gain = torch.ones(7, dtype=targets.dtype)
    ai = torch.arange(na, dtype=targets.dtype).view(na, 1).repeat(1, nt)
    t7 = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=targets.dtype) * g
    gain[2:6] = torch.tensor(pred.shape, dtype=targets.dtype)[[3, 2, 3, 2]]
    t = t7 * gain
    if nt:
        r = t[:, :, 4:6] / anchors[:, None]
        j = torch.max(r, 1. / r).max(2)[0] < anchor_t
        t = t[j]
        gxy = t[:, 2:4]
        gxi = gain[[2, 3]] - gxy
        j, k = ((gxy % 1. < g) & (gxy > 1.)).T
        l, m = ((gxi % 1. < g) & (gxi > 1.)).T
        j = torch.stack((torch.ones_like(j), j, k, l, m))
        t = t.repeat((5, 1, 1))[j]
        offsets = (torch.zeros_like(gxy)[None] + off[:, None])[j]
    else:
        t = t7[0]
        offsets = 0
    b, c = t[:, :2].long().T
    gxy, gwh = t[:, 2:4], t[:, 4:6]
    gij = (gxy - offsets).long()
    gi, gj = gij.T
    a = t[:, 6].long()
    # the reference clamps gj / gi IN PLACE and they are views of gij (loss.py:219), so tbox (:221) sees the clamped cell
    idx = (b, a, gj.clamp_(0, int(gain[3]) - 1), gi.clamp_(0, int(gain[2]) - 1))
    return c, torch.cat((gxy - gij, gwh), 1), idx, anchors[a]


def compute_loss(pred: Tensor, targets: Tensor, anchors: Tensor, hyp: Optional[dict] = None, gr: float = 1.0, nc: int = 8):
    """ComputeLoss.__call__ (loss.py:116-163), one layer (balance[0] = 4.0 for nl == 1, :110), BCE without focal term,
    no label smoothing (smooth_BCE(0.0), :104).  Returns (loss * batch, lbox, lobj, lcls) like the reference."""
    h = dict(LOSS_HYP if hyp is None else hyp)
    lcls, lbox, lobj = pred.new_zeros(1), pred.new_zeros(1), pred.new_zeros(1)
    tcls, tbox, (b, a, gj, gi), anch = build_targets(pred, targets, anchors, h["anchor_t"])
    tobj = torch.zeros_like(pred[..., 0])
    n = b.shape[0]
    if n:
        ps = pred[b, a, gj, gi]
        pxy = ps[:, :2].sigmoid() * 2. - 0.5
        pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anch
        iou = bbox_ciou(torch.cat((pxy, pwh), 1).T, tbox)
        lbox = lbox + (1.0 - iou).mean()
        tobj[b, a, gj, gi] = (1.0 - gr) + gr * iou.detach().clamp(0).type(tobj.dtype)     # duplicates: the last entry wins (CPU)
        if nc > 1:
            tc = torch.zeros_like(ps[:, 5:])
            tc[range(n), tcls] = 1.0
            lcls = lcls + F.binary_cross_entropy_with_logits(ps[:, 5:], tc, pos_weight=pred.new_tensor([h["cls_pw"]]))
    lobj = lobj + F.binary_cross_entropy_with_logits(pred[..., 4], tobj, pos_weight=pred.new_tensor([h["obj_pw"]])) * 4.0
    lbox, lobj, lcls = lbox * h["box"], lobj * h["obj"], lcls * h["cls"]
    return (lbox + lobj + lcls) * pred.shape[0], lbox, lobj, lcls


def synthetic_targets(B: int, per_image: int = 32, nc: int = 8, seed: int = 0) -> Tensor:
    """per image `per_image` boxes, class ~U{0..nc-1}, centre ~U(0.05, 0.95), size ~U(0.01, 0.05)."""
    g = torch.Generator().manual_seed(seed)
    n = B * per_image
    img = torch.arange(B).repeat_interleave(per_image).float()
    cls = torch.randint(0, nc, (n,), generator=g).float()
    xy = 0.05 + 0.9 * torch.rand(n, 2, generator=g)
    wh = 0.01 + 0.04 * torch.rand(n, 2, generator=g)
    return torch.cat((img[:, None], cls[:, None], xy, wh), 1)


def state_dict_spec(img_size: int = 512, nc: int = 8) -> Dict[str, Tuple[int, ...]]:
    """name -> shape for every *parameter and float buffer* of Model(model.yaml).
    Integer / resolution-dependent buffers (relative_position_index, attn_mask,
    num_batches_tracked) are regenerated, never stored in goldens."""
    t = img_size // 4
    s: Dict[str, Tuple[int, ...]] = {}
    e = "image_encoder."
    s[e + "pos_embed"] = (1, t, t, EMBED)
    s[e + "patch_embed.proj.weight"] = (EMBED, EMBED, 1, 1)
    s[e + "patch_embed.proj.bias"] = (EMBED,)
    for c in "rgbi":
        s[e + f"channel_embed_{c}.proj.weight"] = (CH_EMBED, 1, 4, 4)
        s[e + f"channel_embed_{c}.proj.bias"] = (CH_EMBED,)
    for i in range(1, 5):
        s[e + f"chan_block.norm{i}.weight"] = (CH_EMBED,)
        s[e + f"chan_block.norm{i}.bias"] = (CH_EMBED,)
    for si, (C, depth, ws) in enumerate(zip(STAGE_DIMS, STAGE_DEPTHS, STAGE_WINDOWS), start=1):
        res = t >> (si - 1)
        if res <= ws:          # SwinTransformerBlock ctor clamp (backbone_vit.py:1042-1045)
            ws = res
        for i in range(depth):
            p = e + f"stage{si}.{i}."
            linear = (SHIFTS[i] == 0) or si == 3
            for n in ("norm1", "norm2"):
                s[p + n + ".weight"] = (C,)
                s[p + n + ".bias"] = (C,)
            s[p + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, NUM_HEADS)
            s[p + "attn.qkv.weight"] = (3 * C, C)
            s[p + "attn.qkv.bias"] = (3 * C,)
            s[p + "attn.proj.weight"] = (C, C)
            s[p + "attn.proj.bias"] = (C,)
            if linear:
                s[p + "mlp.fc1.weight"] = (4 * C, C)
                s[p + "mlp.fc1.bias"] = (4 * C,)
                s[p + "mlp.fc2.weight"] = (C, 4 * C)
                s[p + "mlp.fc2.bias"] = (C,)
            else:
                s[p + "mlp.fc1.weight"] = (C, C)
                s[p + "mlp.fc1.bias"] = (C,)
                s[p + "mlp.conv1.weight"] = (C, C, 2, 2)
                s[p + "mlp.conv1.bias"] = (C,)
                s[p + "mlp.fc2.weight"] = (C, C)
                s[p + "mlp.fc2.bias"] = (C,)
    for i, C in ((1, 192), (2, 384)):
        s[e + f"pmerging{i}.reduction.weight"] = (2 * C, 4 * C)
        s[e + f"pmerging{i}.norm.weight"] = (2 * C,)
        s[e + f"pmerging{i}.norm.bias"] = (2 * C,)
    s[e + "neck3.weight"] = (512, 768, 1, 1)
    s[e + "neck2.weight"] = (256, 384, 1, 1)
    s[e + "neck1.weight"] = (256, 384, 1, 1)

    def conv(p, c1, c2, k):
        s[p + "conv.weight"] = (c2, c1, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[p + "bn." + n] = (c2,)

    def c3spec(p, c1, c2):
        c_ = c2 // 2
        conv(p + "cv1.", c1, c_, 1)
        conv(p + "cv2.", c1, c_, 1)
        conv(p + "cv3.", 2 * c_, c2, 1)
        conv(p + "m.0.cv1.", c_, c_, 1)
        conv(p + "m.0.cv2.", c_, c_, 3)
    conv("detect.0.", 512, 256, 1)
    c3spec("detect.3.", 512, 256)
    conv("detect.4.", 256, 128, 1)
    c3spec("detect.7.", 384, 128)
    s["detect.8.anchors"] = (1, 3, 2)
    s["detect.8.anchor_grid"] = (1, 1, 3, 1, 1, 2)
    s["detect.8.m.0.weight"] = (3 * (nc + 5), 128, 1, 1)
    s["detect.8.m.0.bias"] = (3 * (nc + 5),)
    return s


def _hash01(name: str, n: int) -> Tensor:
    """Closed-form pseudo-random values in [-1, 1): a function of (key name, flat
    index) only, reproducible anywhere without the reference or a torch RNG."""
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    i = torch.arange(n, dtype=torch.float64)
    v = torch.sin(i * 12.9898 + (h % 9973) * 0.618 + 1.0) * 43758.5453
    return ((v - torch.floor(v)) * 2.0 - 1.0)


def procedural_state_dict(img_size: int = 512, nc: int = 8, dtype=torch.float32) -> SD:
    """Procedural weights: same statistics class as the reference's default init
    (uniform(-1/sqrt(fan_in), +)) so activations stay O(1) through 11 blocks, but
    every value is a closed-form function of (name, index)."""
    return procedural_from_shapes(state_dict_spec(img_size, nc), dtype)


def procedural_from_shapes(shapes: Dict[str, Tuple[int, ...]], dtype=torch.float32) -> SD:
    """The same closed-form weights for ANY name -> shape table (e.g. the float entries of a model's own state_dict, for
    head graphs other than models/model.yaml)."""
    sd: SD = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        n = 1
        for d in shape:
            n *= d
        u = _hash01(name, n)
        if name.endswith("anchors"):
            t = torch.tensor(ANCHORS_PX, dtype=torch.float64).view(1, 3, 2) / DET_STRIDE
        elif name.endswith("anchor_grid"):
            t = torch.tensor(ANCHORS_PX, dtype=torch.float64).view(1, 1, 3, 1, 1, 2)
        elif "norm" in name and name.endswith(".weight") or name.endswith("bn.weight"):
            t = (1.0 + 0.1 * u).view(shape)
        elif name.endswith("running_var"):
            t = (1.0 + 0.25 * u).view(shape)
        elif name.endswith("running_mean"):
            t = (0.1 * u).view(shape)
        elif name.endswith("pos_embed") or name.endswith("relative_position_bias_table"):
            t = (0.2 * u).view(shape)
        elif name.endswith(".bias"):
            t = (0.05 * u).view(shape)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = (u * (1.7 / math.sqrt(fan_in))).view(shape)
        sd[name] = t.to(dtype).contiguous()
    return sd


def synthetic_inputs(B: int, S: int, seed: int = 0, dtype=torch.float32) -> Tuple[Tensor, Tensor]:
    """x_rgb, x_ir ~ U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, 3, S, S, generator=g, dtype=torch.float32).to(dtype),
            torch.rand(B, 3, S, S, generator=g, dtype=torch.float32).to(dtype))
