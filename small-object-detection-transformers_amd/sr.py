"""Super-resolution auxiliary branch on the HIP kernels: DeepLab = Decoder + EDSR x8
(basics/models/deeplabedsr.py:35-73, sr_decoder_noBN_noD.py:6-45, edsr.py:55-102), forward and hand-written backward,
token-major ([B][H][W][C] rows) like the rest of the path.

    low (c1 @ H x W)     -> 1x1 conv (no bias) + ReLU ----------------------------------\
    x   (c2 @ H/2 x W/2) -> 1x1 conv (no bias) + ReLU -> bilinear x2 (align_corners) ----+-> [x | low] -> 3x3 + ReLU -> 3x3 + ReLU -> 1x1 + b
    -> EDSR: 3x3 head; depth x (3x3 + ReLU -> 3x3, + input); 3x3, + head; 3 x (3x3 to 256 + PixelShuffle 2); 3x3 to `ch`
    -> (B, ch, 8H, 8W) float32

(factor 2 as model.py:113-115 builds it: the Decoder resizes both inputs to low's size x factor / 2 = low's own size, which is the
identity for `low` and an exact x2 for x.)

Every convolution is a K-segment GEMM (`sodt_gemm_nt`: a 3x3 = nine spatially shifted views of its input, csrc/gemm.hip) with bias /
ReLU / residual fused into the epilogue (SODT_EPI_BIAS | SODT_EPI_RELU | SODT_EPI_RESID); its input gradient is the same GEMM over
the negated taps with the transposed weights (the ReLU mask of the layer below as SODT_EPI_DRELU, the residual path's gradient as
SODT_EPI_RESID), its weight gradient `sodt_gemm_tn` writing straight into the torch-layout gradient.  The resize, PixelShuffle and the
NCHW float32 boundary are the kernels of csrc/sr.hip; the GEMM layouts of the weights come from `sodt_prep_weights`.  torch only
owns the memory.

The reference reaches this branch through Model(sr=True), which is unreachable in the fork (wrong import path, then a channel
mismatch: SURVEY.md section 8, config reality row 5); the classes themselves import, and tests/golden/sr.pt pins this module
against them (Decoder, EDSR(depth 2) and the full DeepLab(4, 128, 512)).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import ops
from .ops import SegSpec

TAPS3 = tuple((dy, dx) for dy in (-1, 0, 1) for dx in (-1, 0, 1))


def sr_param_shapes(ch: int, c1: int, c2: int, depth: int = 16, width: int = 64) -> Dict[str, Tuple[int, ...]]:
    """state_dict entries of DeepLab(ch, c1, c2) (names relative to the module), in the reference's registration order."""
    s: Dict[str, Tuple[int, ...]] = {
        "sr_decoder.conv1.weight": (c1 // 2, c1, 1, 1), "sr_decoder.conv2.weight": (c2 // 2, c2, 1, 1),
        "sr_decoder.last_conv.0.weight": (256, (c1 + c2) // 2, 3, 3), "sr_decoder.last_conv.2.weight": (128, 256, 3, 3),
        "sr_decoder.last_conv.4.weight": (64, 128, 1, 1), "sr_decoder.last_conv.4.bias": (64,),
        "edsr.head.0.weight": (width, 64, 3, 3), "edsr.head.0.bias": (width,)}
    for i in range(depth):
        for j in (0, 2):
            s[f"edsr.body.{i}.body.{j}.weight"] = (width, width, 3, 3)
            s[f"edsr.body.{i}.body.{j}.bias"] = (width,)
    s[f"edsr.body.{depth}.weight"] = (width, width, 3, 3)
    s[f"edsr.body.{depth}.bias"] = (width,)
    for j in (0, 2, 4):
        s[f"edsr.tail.0.{j}.weight"] = (4 * width, width, 3, 3)
        s[f"edsr.tail.0.{j}.bias"] = (4 * width,)
    s["edsr.tail.1.weight"] = (ch, width, 3, 3)
    s["edsr.tail.1.bias"] = (ch,)
    return s


class DeepLab(torch.nn.Module):
    """Parameter container with the reference DeepLab's module tree (deeplabedsr.py:35-58 -> Decoder sr_decoder_noBN_noD.py:7-25,
    EDSR edsr.py:55-80), so that state_dict keys, registration order and initial statistics match: Decoder convolutions
    kaiming_normal_ (sr_decoder_noBN_noD.py:60-64), EDSR convolutions nn.Conv2d's default.  It is never called: the branch runs
    inside the engine (SRBranch)."""

    def __init__(self, ch, c1=128, c2=512, factor=2, depth=16, width=64):
        super().__init__()
        nn = torch.nn
        conv3 = lambda i, o, bias=True: nn.Conv2d(i, o, 3, padding=1, bias=bias)
        dec = nn.Module()
        dec.conv1 = nn.Conv2d(c1, c1 // 2, 1, bias=False)
        dec.conv2 = nn.Conv2d(c2, c2 // 2, 1, bias=False)
        dec.last_conv = nn.Sequential(conv3((c1 + c2) // 2, 256, False), nn.ReLU(), conv3(256, 128, False), nn.ReLU(), nn.Conv2d(128, 64, 1))
        for m in dec.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
        self.sr_decoder = dec
        ed = nn.Module()
        ed.head = nn.Sequential(conv3(64, width))
        blocks = []
        for _ in range(depth):
            rb = nn.Module()
            rb.body = nn.Sequential(conv3(width, width), nn.ReLU(True), conv3(width, width))
            blocks.append(rb)
        ed.body = nn.Sequential(*blocks, conv3(width, width))
        up = []
        for _ in range(3):                                   # x8 = three (conv to 4 n_feat, PixelShuffle 2) stages (edsr.py:19-24)
            up += [conv3(width, 4 * width), nn.PixelShuffle(2)]
        ed.tail = nn.Sequential(nn.Sequential(*up), conv3(width, ch))
        self.edsr = ed
        self.factor, self.ch, self.c1, self.c2 = factor, ch, c1, c2

    def forward(self, low_level_feat, x):
        raise RuntimeError("model_up runs inside the MI355X engine (Model.forward with sr=True); it has no torch forward")


class _Conv:
    """GEMM views of one Conv2d: w [Np][taps*Cin] (forward), wT [Cin][taps*Np] (input gradient) in the run dtype; Np = Cout rounded
    up to 8 (zero rows / columns: the closing 64 -> ch convolution has 3 or 4 outputs).  The forward GEMM runs at N = Cout over the
    padded rows, as Detect's does (engine.py: det_np)."""
    __slots__ = ("name", "cout", "cin", "k", "taps", "np_", "w", "wT", "bias", "dw_pad", "db_pad", "bias_pad", "wTp")


class SRBranch:
    """One DeepLab(ch, c1, c2, factor=2) (or only its Decoder / EDSR half: whatever `params` holds) bound to its float32 master
    parameters `params[name]` and gradient accumulators `grads[name]` (same shapes, float32; default: freshly zeroed tensors)."""

    def __init__(self, params: Dict[str, torch.Tensor], dt: torch.dtype, grads: Optional[Dict[str, torch.Tensor]] = None,
                 dec: str = "sr_decoder.", edsr: str = "edsr."):
        self.p, self.dt, self.dec, self.ed = params, dt, dec, edsr
        self.dev = next(iter(params.values())).device
        self.g = grads if grads is not None else {k: torch.zeros_like(v) for k, v in params.items()}
        self.depth = 0
        while f"{edsr}body.{self.depth}.body.0.weight" in params:
            self.depth += 1
        self.c: Dict[str, _Conv] = {}
        descs = []
        for k, v in params.items():
            if not k.endswith(".weight"):
                continue
            assert v.dtype == torch.float32 and v.is_contiguous() and v.dim() == 4
            c = _Conv()
            c.name = k[: -len(".weight")]
            c.cout, c.cin, c.k = v.shape[0], v.shape[1], v.shape[2]
            c.taps = c.k * c.k
            c.np_ = (c.cout + 7) // 8 * 8
            c.bias = params.get(c.name + ".bias")
            # Cout % 8 != 0: the weight gradient goes through an [Np][K] scratch (8-column dY: the pipelined TN kernel takes it;
            # the narrow-N fallback kernel needed 14.6 ms per launch at 67 M rows) and its first Cout rows are added to .grad
            c.dw_pad = torch.zeros(c.np_, c.cin * c.taps, device=self.dev) if c.np_ != c.cout else None
            c.db_pad = torch.zeros(c.np_, device=self.dev) if (c.np_ != c.cout and c.bias is not None) else None
            # ... and the forward runs at N = Np over the zero rows of w with a zero-padded bias copy (bf16: the pipelined kernel
            # needs N % 8 == 0; the 128x128 K-loop kernel took 21.8 ms for the closing 64 -> 4 convolution at 67 M rows)
            c.bias_pad = torch.zeros(c.np_, device=self.dev) if c.db_pad is not None else None
            if c.taps == 1 and c.np_ == c.cout and dt == torch.float32:
                c.w = v.detach().view(c.cout, c.cin)
            else:
                c.w = torch.zeros(c.np_, c.taps * c.cin, device=self.dev, dtype=dt)                 # [n][tap*Cin + c]
                descs.append(self._desc(v, c.w, (c.cout, c.cin, c.taps), (0, 2, 1), c.taps * c.cin, 0))
            c.wT = torch.zeros(c.cin, c.taps * c.np_, device=self.dev, dtype=dt)                      # [c][tap*Np + n]
            descs.append(self._desc(v, c.wT, (c.cout, c.cin, c.taps), (1, 2, 0), c.taps * c.np_, c.np_ if c.np_ != c.cout else 0))
            # Upsampler stage (64 -> 256 + PixelShuffle 2, edsr.py:14-24) on the direct kernels: the input gradient of plane p = 2 i + j
            # wants wTp[p][k][tap * 64 + c] = W[4 c + p][k][tap] - the plain transpose of W viewed as [64 (c)][4 * 576 (p, k, tap)]
            c.wTp = None
            if dt == torch.bfloat16 and c.k == 3 and c.cin == 64 and c.cout == 256:
                c.wTp = torch.zeros(4, 64, 576, device=self.dev, dtype=dt)
                descs.append(self._desc(v, c.wTp, (64, 4 * 576, 1), (1, 2, 0), 64, 0))
            self.c[c.name] = c
        arr = (L.PrepDesc * len(descs))(*descs)
        self._tab = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.dev), len(descs),
                     max(d.d0 * d.d1 * d.d2 for d in descs))
        self.bufs: Dict[str, torch.Tensor] = {}
        self.prepare()

    @staticmethod
    def _desc(src, dst, dims, perm, dst_ld, inner_ld):
        d = L.PrepDesc()
        d.src, d.dst = src.data_ptr(), dst.data_ptr()
        d.d0, d.d1, d.d2 = dims
        d.p0, d.p1, d.p2 = perm
        d.dst_ld, d.inner_ld = dst_ld, inner_ld
        return d

    def prepare(self):
        """(Re-)lay the float32 masters out for the GEMMs (one launch); call after every optimizer step."""
        t, n, mx = self._tab
        ops.prep_weights(t, n, mx, L.BF16 if self.dt == torch.bfloat16 else L.F32)
        for c in self.c.values():
            if c.bias_pad is not None:
                ops.cast(c.bias.detach(), c.bias_pad, c.cout)

    # ------------------------------------------------------------------ helpers
    def _buf(self, name, shape, dtype=None):
        t = self.bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = self.bufs[name] = torch.zeros(shape, device=self.dev, dtype=self.dt if dtype is None else dtype)
        return t

    @staticmethod
    def _bias8(c):
        """bias operand of the direct 64 -> <= 8 kernel (it reads 8 floats): the zero-padded copy when cout < 8; at cout == 8 there is
        no padded copy (np_ == cout) and the parameter itself already holds the 8 floats (advisor r5)."""
        if c.bias is None:
            return None
        return c.bias_pad if c.bias_pad is not None else c.bias

    def _conv(self, name, parts: Sequence[SegSpec], H, W, M, out, *, relu=False, resid=None, ldc=None, c_off=0):
        """out[:, c_off : c_off + Cout] = act(conv(name)(input) + bias) [+ resid]; `parts`: the input as channel segments.
        Returns the K-segments (the weight gradient re-reads them)."""
        c = self.c[name]
        if c.k == 1:
            segs = list(parts)
            sp = (H, W) if any(s.shr or s.mul != 1 or s.dy or s.dx for s in segs) else None
        else:
            assert len(parts) == 1, "a 3x3 convolution takes one (materialised) input"
            s0 = parts[0]
            assert s0.shr == 0 and s0.mul == 1
            segs = [SegSpec(s0.t, s0.klen, s0.coff, dy, dx, 1, 0, H, W, ld=s0.ld) for (dy, dx) in TAPS3]
            sp = (H, W)
        padded = c.bias_pad is not None and ldc is None and out.shape[-1] == c.np_        # (the pad columns of `out` receive zeros)
        if self._direct64(c, parts[0], out, ldc):
            # EDSR's 64 -> 64 convolutions (head, the ResBlocks, the body's closing one: edsr.py:34-53, :64-70)
            ops.conv3_c64_fwd(parts[0].t, c.w, out, M // (H * W), H, W, bias=c.bias, relu=relu, resid=resid)
            return segs, sp
        if self._direct3(c, parts[0], out, ldc) and not relu and resid is None:
            # EDSR's closing 64 -> ch convolution (edsr.py:81-84): its own kernel reads the 64-channel input once (csrc/conv3.hip)
            ops.conv3_n8_fwd(parts[0].t, c.w, self._bias8(c), out, M // (H * W), H, W, cout=c.cout)
            return segs, sp
        ops.gemm_nt(segs, c.w, out, M, c.np_ if padded else c.cout, c.taps * c.cin, spatial=sp, bias=c.bias_pad if padded else c.bias,
                    relu=relu, resid=resid, ldc=ldc, c_off=c_off)
        return segs, sp

    def _direct3(self, c, s0, out, ldc=None):
        """True when conv `c` on the dense 64-channel input view s0 with the [M][8] output `out` can take the kernels of csrc/conv3.hip."""
        return (ops.conv3_n8_ok(out, c.cin, c.np_, c.k) and ldc is None and out.shape[-1] == c.np_ and s0.ld == 64 and s0.klen == 64
                and s0.coff == 0 and s0.shr == 0 and s0.mul == 1)

    @staticmethod
    def _direct64(c, s0, out, ldc=None):
        """True when conv `c` is a 3x3 at 64 -> 64 channels on dense bf16 rows (csrc/conv3.hip: sodt_conv3x3_c64_*)."""
        return (out.dtype == torch.bfloat16 and c.k == 3 and c.cin == 64 and c.cout == 64 and ldc is None and out.shape[-1] == 64
                and s0.ld == 64 and s0.klen == 64 and s0.coff == 0 and s0.shr == 0 and s0.mul == 1)

    def _conv_bwd(self, name, dy, lddy, fwd, H, W, M, dx, *, dx_n=None, w_row0=0, drelu_aux=None, aux_off=0, resid=None, wgrad=True):
        """Weight / bias gradients of conv `name` (+= into self.g) and, when dx is given,
        dx = conv^T(dy) for input channels [w_row0, w_row0 + dx_n), masked by drelu_aux > 0, + resid.
        dy: [M][lddy] with the gradient in the first Cout columns and zeros up to Np."""
        c = self.c[name]
        segs, sp = fwd
        if c.k == 3 and lddy == 64 and self._direct64(c, segs[4], dy) and dx_n is None and w_row0 == 0 and aux_off == 0:
            B = M // (H * W)
            if wgrad:
                scr = self._buf("c64.scratch", (ops.conv3_c64_wgrad_scratch_floats(),), torch.float32)
                ops.conv3_c64_wgrad(dy, segs[4].t, self.g[name + ".weight"], self.g[name + ".bias"] if c.bias is not None else None, scr, B, H, W)
            if dx is not None:
                ops.conv3_c64_fwd(dy, c.wT, dx, B, H, W, drelu_aux=drelu_aux, resid=resid, flip=True)
            return
        if (c.k == 3 and lddy == c.np_ and self._direct3(c, segs[4], dy) and dx_n is None and w_row0 == 0 and drelu_aux is None
                and resid is None):
            B = M // (H * W)
            if wgrad:
                scr = self._buf("c3.scratch", (ops.conv3_n8_wgrad_scratch_floats(),), torch.float32)
                ops.conv3_n8_wgrad(dy, segs[4].t, self.g[name + ".weight"], self.g[name + ".bias"] if c.bias is not None else None, scr,
                                   B, H, W, c.cout)
            if dx is not None:
                ops.conv3_n8_dgrad(dy, c.wT, dx, B, H, W, cout=c.cout)
            return
        if wgrad and c.dw_pad is None:
            ops.gemm_tn(dy, segs, self.g[name + ".weight"].view(c.cout, -1), M, c.cout, c.taps * c.cin, ldy=lddy, spatial=sp,
                        dbias=self.g[name + ".bias"] if c.bias is not None else None, kperm=(c.cin, c.taps) if c.k > 1 else None)
        elif wgrad:
            gw = self.g[name + ".weight"]
            ops.zero_(c.dw_pad)
            if c.db_pad is not None:
                ops.zero_(c.db_pad)
            ops.gemm_tn(dy, segs, c.dw_pad, M, c.np_, c.taps * c.cin, ldy=lddy, spatial=sp, dbias=c.db_pad,
                        kperm=(c.cin, c.taps) if c.k > 1 else None)
            ops.add_rows(gw.view(1, -1), c.dw_pad.view(1, -1), 1, gw.numel(), lds=gw.numel())
            if c.db_pad is not None:
                gb = self.g[name + ".bias"]
                ops.add_rows(gb.view(1, -1), c.db_pad.view(1, -1), 1, c.cout, lds=c.cout)      # (cout = 3: sodt_add_rows' short-row path)
        if dx is None:
            return
        n = c.cin if dx_n is None else dx_n
        if c.k == 1:
            bsegs, bsp = [SegSpec(dy, c.np_, 0, ld=lddy)], None
        else:
            bsegs = [SegSpec(dy, c.np_, 0, -ty, -tx, 1, 0, H, W, ld=lddy) for (ty, tx) in TAPS3]
            bsp = (H, W)
        ops.gemm_nt(bsegs, c.wT, dx, M, n, c.taps * c.np_, spatial=bsp, w_off=w_row0 * c.taps * c.np_, drelu_aux=drelu_aux,
                    aux_off=aux_off, resid=resid)

    # ------------------------------------------------------------------ Decoder (sr_decoder_noBN_noD.py:27-45)
    def decoder_forward(self, low_parts: Sequence[SegSpec], x_parts: Sequence[SegSpec], B: int, H: int, W: int) -> torch.Tensor:
        """low_parts: the low-level feature (c1 channels on the H x W grid) as K-segments (upsampling views allowed: set shr / Hi / Wi);
        x_parts: the deep feature (c2 channels on the H/2 x W/2 grid).  Returns the [B*H*W][64] buffer."""
        assert H % 2 == 0 and W % 2 == 0
        d, c = self.dec, self.c
        h, w = H // 2, W // 2
        M1, Mh = B * H * W, B * h * w
        ca, cx = c[d + "conv1"].cout, c[d + "conv2"].cout
        assert ca % 8 == 0 and cx % 8 == 0
        self.geo = (B, H, W)
        cat = self._buf("cat", (M1, cx + ca))
        # low-level: 1x1 + ReLU straight into its concat slice; deep: 1x1 + ReLU, then bilinear x2 into the other slice
        self.f_c1 = self._conv(d + "conv1", low_parts, H, W, M1, cat, relu=True, ldc=cx + ca, c_off=cx)
        xr = self._buf("xr", (Mh, cx))
        self.f_c2 = self._conv(d + "conv2", x_parts, h, w, Mh, xr, relu=True)
        ops.bilinear_up2_fwd(xr, cat, B, h, w, cx, ldy=cx + ca, ycol=0)
        d1 = self._buf("d1", (M1, 256))
        self.f_l0 = self._conv(d + "last_conv.0", [SegSpec(cat)], H, W, M1, d1, relu=True)
        d2 = self._buf("d2", (M1, 128))
        self.f_l2 = self._conv(d + "last_conv.2", [SegSpec(d1)], H, W, M1, d2, relu=True)
        d3 = self._buf("d3", (M1, 64))
        self.f_l4 = self._conv(d + "last_conv.4", [SegSpec(d2)], H, W, M1, d3)
        return d3

    def decoder_backward(self, d3g: torch.Tensor):
        """d3g [B*H*W][64].  Returns (d_low [B*H*W][c1], d_x [B*(H/2)*(W/2)][c2]): the gradients of the two inputs on their own
        grids, dense (the gradient of an upsampling / concatenating input view is reduced / split by the caller)."""
        B, H, W = self.geo
        d, c, b = self.dec, self.c, self.bufs
        h, w = H // 2, W // 2
        M1, Mh = B * H * W, B * h * w
        ca, cx = c[d + "conv1"].cout, c[d + "conv2"].cout
        d2g = self._buf("g.d2", (M1, 128))
        self._conv_bwd(d + "last_conv.4", d3g, 64, self.f_l4, H, W, M1, d2g, drelu_aux=b["d2"])
        d1g = self._buf("g.d1", (M1, 256))
        self._conv_bwd(d + "last_conv.2", d2g, 128, self.f_l2, H, W, M1, d1g, drelu_aux=b["d1"])
        dcu = self._buf("g.cu", (M1, cx))
        da = self._buf("g.a", (M1, ca))
        self._conv_bwd(d + "last_conv.0", d1g, 256, self.f_l0, H, W, M1, dcu, dx_n=cx, w_row0=0)
        self._conv_bwd(d + "last_conv.0", d1g, 256, self.f_l0, H, W, M1, da, dx_n=ca, w_row0=cx, drelu_aux=b["cat"], aux_off=cx,
                       wgrad=False)
        dxr = self._buf("g.xr", (Mh, cx))
        ops.bilinear_up2_bwd(dcu, dxr, B, h, w, cx, relu_out=b["xr"])
        d_low = self._buf("g.low", (M1, c[d + "conv1"].cin))
        d_x = self._buf("g.x", (Mh, c[d + "conv2"].cin))
        self._conv_bwd(d + "conv1", da, ca, self.f_c1, H, W, M1, d_low)
        self._conv_bwd(d + "conv2", dxr, cx, self.f_c2, h, w, Mh, d_x)
        return d_low, d_x

    # ------------------------------------------------------------------ EDSR (edsr.py:55-102)
    def edsr_forward(self, x: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        """x [B*H*W][64] -> (B, ch, 8H, 8W) float32."""
        e, c = self.ed, self.c
        M1 = B * H * W
        self.egeo = (B, H, W)
        hd = self._buf("e.h", (M1, 64))
        self.f_h = self._conv(e + "head.0", [SegSpec(x)], H, W, M1, hd)
        r = hd
        self.f_rb = []
        for i in range(self.depth):
            u = self._buf(f"e.u{i}", (M1, 64))
            f0 = self._conv(f"{e}body.{i}.body.0", [SegSpec(r)], H, W, M1, u, relu=True)
            rn = self._buf(f"e.r{i}", (M1, 64))
            f2 = self._conv(f"{e}body.{i}.body.2", [SegSpec(u)], H, W, M1, rn, resid=r)
            self.f_rb.append((f0, f2, u))
            r = rn
        rb = self._buf("e.rb", (M1, 64))
        self.f_b = self._conv(f"{e}body.{self.depth}", [SegSpec(r)], H, W, M1, rb, resid=hd)
        cur, gh, gw = rb, H, W
        self.f_t = []
        j = 0
        while f"{e}tail.0.{2 * j}" in c:
            cj = c[f"{e}tail.0.{2 * j}"]
            pj = self._buf(f"e.p{j}", (B * 4 * gh * gw, 64))
            if cj.wTp is not None:
                # conv(64 -> 256) + PixelShuffle(2): plane p = 2 i + j (output channels 4 c + p) is a 64 -> 64 convolution whose pixel
                # (y, x) is stored at (2 y + i, 2 x + j) of the shuffled tensor - no 256-channel tensor, no shuffle launch
                for pl in range(4):
                    ops.conv3_c64_fwd(cur, cj.w, pj, B, gh, gw, bias=cj.bias, geo=ops.conv3_geo(w_row=(4, pl), out=(2, pl >> 1, pl & 1)))
                fj = ("direct", cur)
            else:
                z = self._buf(f"e.z{j}", (B * gh * gw, 256))
                fj = self._conv(f"{e}tail.0.{2 * j}", [SegSpec(cur)], gh, gw, B * gh * gw, z)
                ops.pixel_shuffle2(z, pj, B, gh, gw, 64)
            self.f_t.append((fj, gh, gw))
            cur, gh, gw = pj, 2 * gh, 2 * gw
            j += 1
        ct = c[e + "tail.1"]
        y = torch.empty(B, ct.cout, gh, gw, device=self.dev, dtype=torch.float32)
        if ops.conv3_n8_ok(cur, ct.cin, ct.np_, ct.k) and ct.cout <= 4 and cur.shape[-1] == 64:
            # the closing convolution writes the (B, ch, 8H, 8W) float32 output itself (no [M][8] rows, no conversion launch)
            ops.conv3_n8_fwd(cur, ct.w, self._bias8(ct), None, B, gh, gw, y_nchw=y, cout=ct.cout)
            self.f_o = ("direct", cur)
        else:
            o = self._buf("e.o", (B * gh * gw, ct.np_))
            self.f_o = self._conv(e + "tail.1", [SegSpec(cur)], gh, gw, B * gh * gw, o)
            ops.nchw_f32_from_rows(o, y, B, ct.cout, gh, gw)
        return y

    def edsr_backward(self, dy: torch.Tensor) -> torch.Tensor:
        """dy (B, ch, 8H, 8W) float32 -> d(input) [B*H*W][64]."""
        B, H, W = self.egeo
        e, c = self.ed, self.c
        M1 = B * H * W
        nt = len(self.f_t)
        gh, gw = H << nt, W << nt
        ct = c[e + "tail.1"]
        assert dy.dtype == torch.float32 and dy.is_contiguous() and tuple(dy.shape) == (B, ct.cout, gh, gw)
        dcur = self._buf(f"g.p{nt}", (B * gh * gw, 64))
        if self.f_o[0] == "direct":
            # ... and its gradients read the float32 (B, ch, 8H, 8W) gradient in place
            name = e + "tail.1"
            scr = self._buf("c3.scratch", (ops.conv3_n8_wgrad_scratch_floats(),), torch.float32)
            ops.conv3_n8_wgrad(None, self.f_o[1], self.g[name + ".weight"], self.g[name + ".bias"] if ct.bias is not None else None, scr,
                               B, gh, gw, ct.cout, dy_nchw=dy)
            ops.conv3_n8_dgrad(None, ct.wT, dcur, B, gh, gw, dy_nchw=dy, cout=ct.cout)
        else:
            do = self._buf("g.o", (B * gh * gw, ct.np_))
            ops.rows_from_nchw_f32(dy, do, B, ct.cout, gh, gw)
            self._conv_bwd(e + "tail.1", do, ct.np_, self.f_o, gh, gw, B * gh * gw, dcur)
        for j in reversed(range(nt)):
            fj, hj, wj = self.f_t[j]
            dsrc = self._buf(f"g.p{j}", (B * hj * wj, 64))
            if fj[0] == "direct":
                # the four planes of the fine gradient are read in place: weight / bias gradient rows 4 c + p, and the input gradient
                # summed over the planes (launch p adds to what launches < p stored: the residual operand is the output itself)
                name, cj = f"{e}tail.0.{2 * j}", c[f"{e}tail.0.{2 * j}"]
                scr = self._buf("c64.scratch", (ops.conv3_c64_wgrad_scratch_floats(),), torch.float32)
                for pl in range(4):
                    ops.conv3_c64_wgrad(dcur, fj[1], self.g[name + ".weight"], self.g[name + ".bias"], scr, B, hj, wj,
                                        geo=ops.conv3_geo(w_row=(4, pl), out=(2, pl >> 1, pl & 1)))
                    ops.conv3_c64_fwd(dcur, cj.wTp[pl], dsrc, B, hj, wj, flip=True, resid=dsrc if pl else None,
                                      geo=ops.conv3_geo(inp=(2, pl >> 1, pl & 1)))
            else:
                dz = self._buf(f"g.z{j}", (B * hj * wj, 256))
                ops.pixel_shuffle2(dcur, dz, B, hj, wj, 64, inverse=True)
                self._conv_bwd(f"{e}tail.0.{2 * j}", dz, 256, fj, hj, wj, B * hj * wj, dsrc)
            dcur = dsrc
        d_rb = dcur            # d(closing-conv output + head output): the head receives it directly and through the residual chain
        dr = self._buf("g.r", (M1, 64))
        self._conv_bwd(f"{e}body.{self.depth}", d_rb, 64, self.f_b, H, W, M1, dr)
        for i in reversed(range(self.depth)):
            f0, f2, u = self.f_rb[i]
            du = self._buf("g.u", (M1, 64))
            self._conv_bwd(f"{e}body.{i}.body.2", dr, 64, f2, H, W, M1, du, drelu_aux=u)
            drn = self._buf(f"g.r{i % 2}", (M1, 64))
            self._conv_bwd(f"{e}body.{i}.body.0", du, 64, f0, H, W, M1, drn, resid=dr)
            dr = drn
        ops.add_rows(dr, d_rb, M1, 64)
        dx = self._buf("g.ein", (M1, 64))
        self._conv_bwd(e + "head.0", dr, 64, self.f_h, H, W, M1, dx)
        return dx

    # ------------------------------------------------------------------ DeepLab.forward (deeplabedsr.py:61-73)
    def forward(self, low_parts, x_parts, B, H, W):
        return self.edsr_forward(self.decoder_forward(low_parts, x_parts, B, H, W), B, H, W)

    def backward(self, dy):
        return self.decoder_backward(self.edsr_backward(dy))
