"""Thin Python wrappers over the C ABI (include/sodt_hip.h).

Every wrapper builds the argument record once, launches on torch's current stream and,
when a ``Recorder`` is active, also appends ``(cfunc, args)`` to it so that the engine
can replay a whole forward / backward with ~1 us of host work per kernel (shapes and
buffers are static per (batch, resolution, dtype) plan).  PyTorch is used only for
device memory and streams.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

_lib = L.load()

_active_recorder: Optional[list] = None


class Recorder:
    """with Recorder() as rec: ...ops...  -> rec.calls can be replayed with replay()."""

    def __init__(self):
        self.calls: List[Tuple] = []
        self.keep: List = []       # tensors / structs that must outlive the plan

    def __enter__(self):
        global _active_recorder
        assert _active_recorder is None, "nested recorders are not supported"
        _active_recorder = self.calls
        return self

    def __exit__(self, *exc):
        global _active_recorder
        _active_recorder = None
        return False


def recorded_count() -> Optional[int]:
    """Launches recorded so far by the active Recorder (None outside one): lets the engine mark split points."""
    return None if _active_recorder is None else len(_active_recorder)


_tag: str = ""          # label attached to recorded launches (set by the engine, read by bench probes)


def set_tag(tag: str) -> None:
    global _tag
    _tag = tag


def replay(calls: Sequence[Tuple], stream: Optional[int] = None, probes: Optional[dict] = None,
           start: int = 0, end: Optional[int] = None) -> None:
    """Re-issue recorded launches [start, end).  probes: {call index: (start_event, end_event)} -> the events are
    recorded on the launch stream around that one kernel (bench.py's live roofline measurement)."""
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream if stream is None else stream)
    if start or end is not None:
        calls = calls[start:end]
    if probes:
        for i, (fn, args, name, tag) in enumerate(calls, start):
            pr = probes.get(i)
            if pr is not None:
                pr[0].record()
            rc = fn(*args, st)
            if pr is not None:
                pr[1].record()
            if rc != 0:
                raise RuntimeError(f"{name} failed with status {rc}")
        return
    for fn, args, name, tag in calls:
        rc = fn(*args, st)
        if rc != 0:
            raise RuntimeError(f"{name} [{tag}] failed with status {rc}")


def _launch(name: str, *args) -> None:
    fn = getattr(_lib, name)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(*args, st)
    if rc != 0:
        raise RuntimeError(f"{name} failed with status {rc} (unsupported shape/alignment, see include/sodt_hip.h)")
    if _active_recorder is not None:
        _active_recorder.append((fn, args, name, _tag))


def dt_code(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float32:
        return L.F32
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class SegSpec:
    """One K-segment: a [rows][ld] token-major tensor view starting at channel ``coff``."""
    __slots__ = ("t", "klen", "coff", "dy", "dx", "mul", "shr", "Hi", "Wi", "ld")

    def __init__(self, t: torch.Tensor, klen: Optional[int] = None, coff: int = 0, dy: int = 0, dx: int = 0,
                 mul: int = 1, shr: int = 0, Hi: int = 0, Wi: int = 0, ld: Optional[int] = None):
        self.t = t
        self.ld = t.shape[-1] if ld is None else ld
        self.klen = (self.ld - coff) if klen is None else klen
        self.coff, self.dy, self.dx, self.mul, self.shr, self.Hi, self.Wi = coff, dy, dx, mul, shr, Hi, Wi


def _fill_aspec(a: L.ASpec, segs: Sequence[SegSpec], spatial: Optional[Tuple[int, int]]):
    assert 1 <= len(segs) <= L.MAX_SEG
    es = segs[0].t.element_size()
    for i, s in enumerate(segs):
        d = a.s[i]
        d.p = s.t.data_ptr() + s.coff * es
        d.ld, d.klen, d.dy, d.dx, d.mul, d.shr, d.Hi, d.Wi = s.ld, s.klen, s.dy, s.dx, s.mul, s.shr, s.Hi, s.Wi
    a.nseg = len(segs)
    if spatial is None:
        a.spatial, a.Ho, a.Wo = 0, 1, 1
    else:
        a.spatial, a.Ho, a.Wo = 1, spatial[0], spatial[1]


def gemm_nt(segs: Sequence[SegSpec], W: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, *,
            ldw: Optional[int] = None, ldc: Optional[int] = None, c_off: int = 0,
            spatial: Optional[Tuple[int, int]] = None, bias: Optional[torch.Tensor] = None,
            resid: Optional[torch.Tensor] = None, ldr: Optional[int] = None, r_off: int = 0, rmod: int = 0,
            gelu_out: Optional[torch.Tensor] = None, dgelu_aux: Optional[torch.Tensor] = None,
            stats: Optional[torch.Tensor] = None, affine: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
            out_f32: bool = False, detect: Optional[Tuple[int, int, int]] = None,
            oscatter: Optional[Tuple[int, int, int, int, int]] = None, w_off: int = 0,
            gelu_only: bool = False, dgelu_rc: bool = False, relu: bool = False, drelu_aux: Optional[torch.Tensor] = None, aux_off: int = 0) -> None:
    """out[M][N] = epilogue(concat_k(segs) @ W[N][K]^T); see SODT_EPI_* in include/sodt_hip.h."""
    g = L.GemmArgs()
    _fill_aspec(g.a, segs, spatial)
    es = W.element_size()
    g.W = W.data_ptr() + w_off * es
    g.ldw = W.shape[-1] if ldw is None else ldw
    oes = 4 if (out_f32 or detect is not None) else es
    g.C = out.data_ptr() + c_off * oes
    g.ldc = (out.shape[-1] if ldc is None else ldc)
    flags = 0
    if bias is not None:
        flags |= L.EPI_BIAS
        g.bias = bias.data_ptr()
    if resid is not None:
        flags |= L.EPI_RESID
        g.R = resid.data_ptr() + r_off * es
        g.ldr = resid.shape[-1] if ldr is None else ldr
        g.rmod = rmod
    if gelu_out is not None:
        flags |= L.EPI_GELU_DUAL
        g.C2 = gelu_out.data_ptr()
        g.ldc2 = gelu_out.shape[-1]
    if dgelu_aux is not None:
        flags |= L.EPI_DGELU
        g.aux = dgelu_aux.data_ptr()
        g.ldaux = dgelu_aux.shape[-1]
    if stats is not None:
        flags |= L.EPI_STATS
        g.stats = stats.data_ptr()
    if affine is not None:
        flags |= L.EPI_AFFINE_SILU
        g.scale, g.shift = affine[0].data_ptr(), affine[1].data_ptr()
    if out_f32:
        flags |= L.EPI_OUT_F32
    if detect is not None:
        flags |= L.EPI_DETECT
        g.det_na, g.det_no, g.det_hw = detect
    if oscatter is not None:
        g.oscatter = 1
        g.omul, g.ody, g.odx, g.OH, g.OW = oscatter
    if gelu_only:
        flags |= L.EPI_GELU
    if dgelu_rc:
        flags |= L.EPI_DGELU_RC
    if relu:
        flags |= L.EPI_RELU
    if drelu_aux is not None:       # gradient through a ReLU whose output is drelu_aux
        flags |= L.EPI_DRELU
        g.aux = drelu_aux.data_ptr() + aux_off * drelu_aux.element_size()
        g.ldaux = drelu_aux.shape[-1]
    g.M, g.N, g.K, g.flags = M, N, K, flags
    _launch("sodt_gemm_nt", C.byref(g), dt_code(W))


def mlp_recompute_ok(M: int, Cc: int, dtype: torch.dtype) -> bool:
    """Linear-GELU-Linear with only the activation saved (SODT_EPI_GELU forward, SODT_EPI_DGELU_RC backward): needs the
    pipelined bf16 kernel for N = 4C, K = 2C (csrc/gemm3.hip: N % 192 == 0, K-halves whole 64-wide steps, bias in LDS)."""
    return dtype == torch.bfloat16 and M >= 256 and (4 * Cc) % 192 == 0 and Cc % 64 == 0 and 4 * Cc <= 3072


def mlp_fused_ok(M: int, Cc: int, dtype: torch.dtype) -> bool:
    """sodt_mlp_fwd runs its fused kernel (csrc/mlp.hip) for this shape; otherwise it is the two-GEMM chain through hact."""
    return dtype == torch.bfloat16 and Cc == 192 and M >= 1


def mlp_fwd(xn: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor,
            resid: Optional[torch.Tensor], out: torch.Tensor, hact: Optional[torch.Tensor], M: int, Cc: int) -> None:
    """out = resid + fc2(GELU(fc1(xn))) (backbone_vit.py:884-890, :1128); hact (optional) receives GELU(fc1(xn)) [M][4C]."""
    assert xn.shape[-1] == Cc and out.shape[-1] == Cc and w1.shape == (4 * Cc, Cc) and w2.shape == (Cc, 4 * Cc)
    assert xn.is_contiguous() and out.is_contiguous() and w1.is_contiguous() and w2.is_contiguous()
    assert w1.dtype == xn.dtype and w2.dtype == xn.dtype and b1.dtype == torch.float32 and b2.dtype == torch.float32
    assert resid is None or (resid.is_contiguous() and resid.shape[-1] == Cc and resid.dtype == xn.dtype)
    assert hact is None or (hact.is_contiguous() and hact.shape[-1] == 4 * Cc and hact.dtype == xn.dtype)
    _launch("sodt_mlp_fwd", _p(xn), _p(w1), _p(b1), _p(w2), _p(b2), _p(resid), _p(out), _p(hact), M, Cc, dt_code(xn))


# ---- 2x2-conv MLP with fc1 folded into the convolution (csrc/convmlp.hip)
def convmlp_compose(fc1_w, fc1_b, conv_w, conv_b, weff, weffT, beff, vtap, Cc):
    assert fc1_w.dtype == torch.float32 and conv_w.dtype == torch.float32 and fc1_w.is_contiguous() and conv_w.is_contiguous()
    assert tuple(conv_w.shape) == (Cc, Cc, 2, 2) and tuple(fc1_w.shape) == (Cc, Cc) and tuple(weff.shape) == (Cc, 4 * Cc) == tuple(weffT.shape)
    _launch("sodt_convmlp_compose", _p(fc1_w), _p(fc1_b), _p(conv_w), _p(conv_b), _p(weff), _p(weffT), _p(beff), _p(vtap), Cc, dt_code(weff))


def convmlp_border_fix(cp, ca, vtap, B, H, W, Cc):
    _launch("sodt_convmlp_border_fix", _p(cp), _p(ca), _p(vtap), B, H, W, Cc, dt_code(cp))


def convmlp_border_sums(dc, bs, B, H, W, Cc):
    _launch("sodt_convmlp_border_sums", _p(dc), _p(bs), B, H, W, Cc, dt_code(dc))


def convmlp_decompose(dweff, colsum, bs, fc1_w, fc1_b, conv_w, g_conv_w, g_conv_b, g_fc1_w, g_fc1_b, Cc):
    for t in (dweff, colsum, bs, fc1_w, fc1_b, conv_w, g_conv_w, g_conv_b, g_fc1_w, g_fc1_b):
        assert t.dtype == torch.float32 and t.is_contiguous()
    _launch("sodt_convmlp_decompose", _p(dweff), _p(colsum), _p(bs), _p(fc1_w), _p(fc1_b), _p(conv_w), _p(g_conv_w), _p(g_conv_b),
            _p(g_fc1_w), _p(g_fc1_b), Cc)


def tn_splits(M: int, N: int, K: int, bf16: bool = False) -> int:
    """M-slices per dW tile so that the grid is one full round of workgroups (no tail): 256 x 192 tiles at one
    workgroup per CU when the short side is <= 192, 128 x 128 tiles at two per CU otherwise (csrc/gemm.hip).
    bf16 (M >= 1024): the pipelined kernel of csrc/gemm3.hip - 256 x 192 tiles, the 256 side on whichever of
    N / K pads less, one workgroup per CU, at least 16 stages of 32 rows per workgroup."""
    if bf16 and M >= 1024 and N % 8 == 0 and K % 8 == 0:
        a = ((N + 255) // 256) * ((K + 191) // 192)
        b = ((K + 255) // 256) * ((N + 191) // 192)
        tiles = b if b * 256 * 192 < a * 256 * 192 else a
        return max(1, min(M // 512, max(1, 256 // tiles)))
    if N <= 192 or K <= 192:
        tiles, slots = ((N + 255) // 256) * ((K + 191) // 192), 256
    else:
        tiles, slots = ((N + 127) // 128) * ((K + 127) // 128), 1536
    return max(1, min(M // 256 if M >= 256 else 1, max(1, slots // tiles)))


_tn_scratch: Optional[torch.Tensor] = None


def set_tn_scratch(t: Optional[torch.Tensor]) -> None:
    """f32 device scratch for the weight-gradient partial tiles (sodt_gemm_tn_args.partial); kept alive here because
    recorded launch plans hold its address."""
    global _tn_scratch
    assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
    _tn_scratch = t


def gemm_tn(dY: torch.Tensor, segs: Sequence[SegSpec], dW: torch.Tensor, M: int, N: int, K: int, *,
            ldy: Optional[int] = None, y_off: int = 0, spatial: Optional[Tuple[int, int]] = None,
            dbias: Optional[torch.Tensor] = None, lddw: Optional[int] = None, kperm: Optional[Tuple[int, int]] = None,
            splits: Optional[int] = None) -> None:
    """dW[N][K] (f32) += dY[M][N]^T @ concat_k(segs); dbias[N] += column sums of dY."""
    g = L.GemmTnArgs()
    g.dY = dY.data_ptr() + y_off * dY.element_size()
    g.ldy = dY.shape[-1] if ldy is None else ldy
    _fill_aspec(g.x, segs, spatial)
    assert dW.dtype == torch.float32
    g.dW = dW.data_ptr()
    g.lddw = K if lddw is None else lddw
    g.dbias = _p(dbias)
    g.M, g.N, g.K = M, N, K
    if kperm is not None:
        g.kperm_c, g.kperm_t = kperm
    g.splits = tn_splits(M, N, K, dY.dtype == torch.bfloat16) if splits is None else splits
    if _tn_scratch is not None and _tn_scratch.device == dY.device and g.splits > 1:
        g.partial, g.partial_floats = _tn_scratch.data_ptr(), _tn_scratch.numel()
    _launch("sodt_gemm_tn", C.byref(g), dt_code(dY))


def layernorm_fwd(x, gamma, beta, y, stats, M, Cc):
    _launch("sodt_layernorm_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(stats), M, Cc, dt_code(x))


def layernorm_bwd(dy, x, stats, gamma, dres, dx, dgamma, dbeta, M, Cc):
    _launch("sodt_layernorm_bwd", _p(dy), _p(x), _p(stats), _p(gamma), _p(dres), _p(dx), _p(dgamma), _p(dbeta),
            M, Cc, dt_code(x))


def window_attn_fwd(qkv, bias_t, out, lse, B, H, W, Cc, heads, ws, shift):
    _launch("sodt_window_attn_fwd", _p(qkv), _p(bias_t), _p(out), _p(lse), B, H, W, Cc, heads, ws, shift, dt_code(qkv))


def window_attn_bwd(qkv, bias_t, out, dout, lse, dqkv, dbias_t, scratch, B, H, W, Cc, heads, ws, shift):
    _launch("sodt_window_attn_bwd", _p(qkv), _p(bias_t), _p(out), _p(dout), _p(lse), _p(dqkv), _p(dbias_t), _p(scratch),
            B, H, W, Cc, heads, ws, shift, dt_code(qkv))


def wmsa_pack_bytes(Cc: int, heads: int, ws: int, dtype_code: int) -> int:
    """0 when the fused block kernel does not take this geometry (csrc/wmsa_block.hip: C 192, 12 heads, 8x8 windows)."""
    return int(_lib.sodt_wmsa_pack_bytes(Cc, heads, ws, dtype_code))


def wmsa_pack(qkv_w, qkv_b, proj_w, proj_b, table, n1w, n1b, n2w, n2b, wpk, Cc, heads, ws):
    _launch("sodt_wmsa_pack", _p(qkv_w), _p(qkv_b), _p(proj_w), _p(proj_b), _p(table), _p(n1w), _p(n1b), _p(n2w), _p(n2b),
            _p(wpk), Cc, heads, ws, L.BF16 if wpk.dtype == torch.bfloat16 else L.F32)


def wmsa_block_fwd(x, wpk, xm, xn2, st1, st2, xn1, qkvw, lsew, ao, B, H, W, Cc, heads, ws, shift):
    """x_mid = x + Proj(W-MSA(LN1 x)), xn2 = LN2(x_mid) in one launch; xn1 .. ao are the save-for-backward outputs (None: inference)."""
    _launch("sodt_wmsa_block_fwd", _p(x), _p(wpk), _p(xm), _p(xn2), _p(st1), _p(st2), _p(xn1), _p(qkvw), _p(lsew), _p(ao),
            B, H, W, Cc, heads, ws, shift, dt_code(x))


def window_attn_bwd_wm(qkvw, bias_t, dout, lsew, dqkv, dbias_t, B, H, W, Cc, heads, ws, shift):
    _launch("sodt_window_attn_bwd_wm", _p(qkvw), _p(bias_t), _p(dout), _p(lsew), _p(dqkv), _p(dbias_t),
            B, H, W, Cc, heads, ws, shift, dt_code(dout))


def wmsa_block_bwd(xn1, wpk, bias_t, dout, lsew, dqkv, dbias_t, B, H, W, Cc, heads, ws, shift):
    """Attention backward of the fused block (bf16) with q / k / v recomputed from the saved LN1 output and the parameter pack."""
    _launch("sodt_wmsa_block_bwd", _p(xn1), _p(wpk), _p(bias_t), _p(dout), _p(lsew), _p(dqkv), _p(dbias_t),
            B, H, W, Cc, heads, ws, shift, dt_code(dout))


# ---- super-resolution branch data movement (csrc/sr.hip)
def bilinear_up2_fwd(x, y, B, H, W, Cc, ldy=None, ycol=0):
    """y[:, ycol : ycol + C] = bilinear x2 (align_corners=True) of x [B*H*W][C]; y may be a wider [B*4HW][ldy] buffer."""
    ldy = y.shape[-1] if ldy is None else ldy
    _launch("sodt_bilinear_up2_fwd", x.data_ptr(), y.data_ptr() + ycol * y.element_size(), ldy, B, H, W, Cc, dt_code(x))


def bilinear_up2_bwd(dy, dx, B, H, W, Cc, lddy=None, dycol=0, relu_out=None):
    lddy = dy.shape[-1] if lddy is None else lddy
    _launch("sodt_bilinear_up2_bwd", dy.data_ptr() + dycol * dy.element_size(), lddy, dx.data_ptr(), _p(relu_out), B, H, W, Cc, dt_code(dx))


def pixel_shuffle2(src, dst, B, H, W, Cc, inverse=False):
    """forward: src [B*H*W][4C] -> dst [B*2H*2W][C] (nn.PixelShuffle(2)); inverse: src [B*2H*2W][C] -> dst [B*H*W][4C]."""
    _launch("sodt_pixel_shuffle2", src.data_ptr(), dst.data_ptr(), B, H, W, Cc, int(bool(inverse)), dt_code(src))


def conv3_n8_ok(t: torch.Tensor, cin: int, np_: int, k: int) -> bool:
    """EDSR's closing convolution (64 -> <= 8 channels, 3x3) has its own kernels in bf16 (csrc/conv3.hip)."""
    return t.dtype == torch.bfloat16 and cin == 64 and np_ == 8 and k == 3


def conv3_n8_fwd(x, w, bias, y, B, H, W, y_nchw=None, cout=8):
    """y [B*H*W][8] = conv3x3(x [B*H*W][64]; w [8][576] = [n][tap*64 + c]) + bias (f32[8] or None); y_nchw (B, cout, H, W) float32:
    the result goes there instead (unrounded) and y may be None."""
    _launch("sodt_conv3x3_c64n8_fwd", x.data_ptr(), w.data_ptr(), _p(bias), _p(y), _p(y_nchw), B, H, W, cout, dt_code(x))


def conv3_n8_dgrad(dy, wT, dx, B, H, W, dy_nchw=None, cout=8):
    """dx [B*H*W][64] = conv3x3^T(dy [B*H*W][8] or dy_nchw (B, cout <= 4, H, W) float32; wT [64][72] = [c][tap*8 + n])."""
    _launch("sodt_conv3x3_c64n8_dgrad", _p(dy), _p(dy_nchw), wT.data_ptr(), dx.data_ptr(), B, H, W, cout, dt_code(dx))


def conv3_n8_wgrad_scratch_floats() -> int:
    return int(_lib.sodt_conv3x3_c64n8_wgrad_scratch_bytes()) // 4


def conv3_n8_wgrad(dy, x, dw, db, scratch, B, H, W, cout, dy_nchw=None):
    """dw [cout][64][3][3] f32 += , db [cout] f32 += (or None); dy rows or dy_nchw float32; scratch: conv3_n8_wgrad_scratch_floats() float32."""
    assert dw.dtype == torch.float32 and dw.is_contiguous() and scratch.dtype == torch.float32
    _launch("sodt_conv3x3_c64n8_wgrad", _p(dy), _p(dy_nchw), x.data_ptr(), dw.data_ptr(), _p(db), scratch.data_ptr(), B, H, W, cout, dt_code(x))


def conv3_geo(w_row=(1, 0), inp=(1, 0, 0), out=(1, 0, 0)):
    """sodt_conv3_geo: output channel n <-> weight row w_row[0] * n + w_row[1]; input / output pixel (y, x) <-> (m y + i, m x + j)."""
    g = L.Conv3Geo()
    g.w_row_stride, g.w_row_off = w_row
    g.in_mul, g.in_i, g.in_j = inp
    g.out_mul, g.out_i, g.out_j = out
    return g


def conv3_c64_fwd(x, w, y, B, H, W, bias=None, relu=False, drelu_aux=None, resid=None, flip=False, geo=None):
    """y [B*H*W][64] = epilogue(conv3x3(x [B*H*W][64]; w [64][576] = [n][tap*64 + c])), bf16 (csrc/conv3.hip); flip: mirrored taps (the
    input gradient with the transposed weights); geo: conv3_geo(...) maps (H x W is the grid the kernel walks)."""
    flags = ((L.EPI_BIAS if bias is not None else 0) | (L.EPI_RELU if relu else 0) | (L.EPI_DRELU if drelu_aux is not None else 0)
             | (L.EPI_RESID if resid is not None else 0))
    _launch("sodt_conv3x3_c64_fwd", x.data_ptr(), w.data_ptr(), _p(bias), _p(resid), _p(drelu_aux), y.data_ptr(), B, H, W, flags,
            int(bool(flip)), C.byref(geo) if geo is not None else None, dt_code(x))


def conv3_c64_wgrad_scratch_floats() -> int:
    return int(_lib.sodt_conv3x3_c64_wgrad_scratch_bytes()) // 4


def conv3_c64_wgrad(dy, x, dw, db, scratch, B, H, W, geo=None):
    """dw [..][64][3][3] f32 +=, db f32 += (or None), bf16 operands; geo: dy through the out_* map, dw / db rows through w_row."""
    assert dw.dtype == torch.float32 and dw.is_contiguous() and scratch.dtype == torch.float32
    _launch("sodt_conv3x3_c64_wgrad", dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _p(db), scratch.data_ptr(), B, H, W,
            C.byref(geo) if geo is not None else None, dt_code(dy))


def add_rows(dst, src, M, Cc, ldd=None, dcol=0, lds=None, scol=0):
    _launch("sodt_add_rows", dst.data_ptr(), dst.shape[-1] if ldd is None else ldd, dcol, src.data_ptr(),
            src.shape[-1] if lds is None else lds, scol, C.c_long(M), Cc, dt_code(dst))


def nchw_f32_from_rows(rows, y, B, Cc, H, W):
    _launch("sodt_nchw_f32_from_rows", rows.data_ptr(), rows.shape[-1], y.data_ptr(), B, Cc, H, W, dt_code(rows))


def rows_from_nchw_f32(y, rows, B, Cc, H, W):
    _launch("sodt_rows_from_nchw_f32", y.data_ptr(), rows.data_ptr(), rows.shape[-1], B, Cc, H, W, dt_code(rows))


def frontend_fwd(rgb, ir_plane, ir_bstride, w, b, gamma, beta, out, B, S, ca_ws=1):
    _launch("sodt_frontend_fwd", _p(rgb), _p(ir_plane), ir_bstride, _p(w), _p(b), _p(gamma), _p(beta), _p(out),
            B, S, ca_ws, dt_code(out))


def frontend_bwd_workspace_bytes(B: int, S: int) -> int:
    return int(_lib.sodt_frontend_bwd_workspace_bytes(B, S))


def frontend_bwd(rgb, ir_plane, ir_bstride, w, b, gamma, beta, dout, dw, db, dgamma, dbeta, B, S, ca_ws=1, ws=None):
    """``ws``: f32 scratch of ``frontend_bwd_workspace_bytes`` (two-stage reduction); None = direct atomics."""
    _launch("sodt_frontend_bwd", _p(rgb), _p(ir_plane), ir_bstride, _p(w), _p(b), _p(gamma), _p(beta), _p(dout),
            _p(dw), _p(db), _p(dgamma), _p(dbeta), B, S, ca_ws, _p(ws), 0 if ws is None else ws.numel() * ws.element_size(),
            dt_code(dout))


def patch_embed4_fwd(rgb, ir_plane, ir_bstride, w, b, e, B, S):
    _launch("sodt_patch_embed4_fwd", _p(rgb), _p(ir_plane), ir_bstride, _p(w), _p(b), _p(e), B, S)


def patch_embed4_bwd(rgb, ir_plane, ir_bstride, de, dw, db, B, S):
    _launch("sodt_patch_embed4_bwd", _p(rgb), _p(ir_plane), ir_bstride, _p(de), _p(dw), _p(db), B, S)


def cross_attn_ln_fwd(e, gamma, beta, out, B, S, ws, shift):
    _launch("sodt_cross_attn_ln_fwd", _p(e), _p(gamma), _p(beta), _p(out), B, S, ws, shift, dt_code(out))


def cross_attn_ln_bwd(e, gamma, dout, de, dgamma, dbeta, B, S, ws, shift):
    _launch("sodt_cross_attn_ln_bwd", _p(e), _p(gamma), _p(dout), _p(de), _p(dgamma), _p(dbeta), B, S, ws, shift, dt_code(dout))


def bn_finalize(stats, mean_rstd, running_mean, running_var, count, Cc, eps, momentum):
    _launch("sodt_bn_finalize", _p(stats), _p(mean_rstd), _p(running_mean), _p(running_var), count, Cc,
            C.c_float(eps), C.c_float(momentum))


def bn_affine(mean_rstd, gamma, beta, scale, shift, Cc):
    _launch("sodt_bn_affine", _p(mean_rstd), _p(gamma), _p(beta), _p(scale), _p(shift), Cc)


def bn_silu_fwd(z, mean_rstd, gamma, beta, y, ldy, M, Cc):
    _launch("sodt_bn_silu_fwd", _p(z), _p(mean_rstd), _p(gamma), _p(beta), _p(y), ldy, M, Cc, dt_code(z))


def col_stats(z, stats, M, Cc):
    """stats [STATS_REPL][2][C] f64 += column sums / sums of squares of z [M][ld] (BatchNorm batch statistics of a stored conv output)."""
    _launch("sodt_col_stats", z.data_ptr(), z.shape[-1], stats.data_ptr(), C.c_long(M), Cc, dt_code(z))


def bn_silu_bwd_reduce(dy, lddy, z, mean_rstd, gamma, beta, red, M, Cc, dy_off=0):
    _launch("sodt_bn_silu_bwd_reduce", dy.data_ptr() + dy_off * dy.element_size(), lddy, _p(z), _p(mean_rstd), _p(gamma),
            _p(beta), _p(red), M, Cc, dt_code(z))


def bn_silu_bwd_apply(dy, lddy, z, mean_rstd, gamma, beta, red, dz, dgamma, dbeta, M, Cc, dy_off=0):
    _launch("sodt_bn_silu_bwd_apply", dy.data_ptr() + dy_off * dy.element_size(), lddy, _p(z), _p(mean_rstd), _p(gamma),
            _p(beta), _p(red), _p(dz), _p(dgamma), _p(dbeta), M, Cc, dt_code(z))


def copy_rows(src, lds, dst, ldd, B, Ho, Wo, shr, Cc, dst_off=0):
    _launch("sodt_copy_rows", _p(src), lds, dst.data_ptr() + dst_off * dst.element_size(), ldd, B, Ho, Wo, shr, Cc, dt_code(src))


def gather_sum_rows(d, ldd, dsrc, lds, B, Hs, Ws, shr, Cc, accumulate=False, d_off=0):
    _launch("sodt_gather_sum_rows", d.data_ptr() + d_off * d.element_size(), ldd, _p(dsrc), lds, B, Hs, Ws, shr, Cc,
            1 if accumulate else 0, dt_code(d))


def detect_unpermute(dpred, dz, ldz, B, HW, na, no):
    _launch("sodt_detect_unpermute", _p(dpred), _p(dz), ldz, B, HW, na, no, dt_code(dz))


def detect_decode(raw, anchor_grid, z, B, na, ny, nx, no, stride):
    _launch("sodt_detect_decode", _p(raw), _p(anchor_grid), _p(z), B, na, ny, nx, no, C.c_float(stride))


def nms_candidates(z_img, conf_thres, multi_label, class_allow, keys, count):
    """z_img (N, 5+nc) f32 of one image -> sort keys + device count (general.py:433-480)."""
    N, no = z_img.shape
    _launch("sodt_nms_candidates", _p(z_img), N, no - 5, C.c_float(conf_thres), int(bool(multi_label)),
            _p(class_allow) if class_allow is not None else None, _p(keys), keys.numel(), _p(count))


def nms_workspace_bytes(n_total: int) -> int:
    b = C.c_size_t(0)
    rc = _lib.sodt_nms_workspace_bytes(int(n_total), C.byref(b))
    if rc != 0:
        raise RuntimeError(f"sodt_nms_workspace_bytes failed with status {rc}")
    return int(b.value)


def nms_select(z_img, keys, n_total, iou_thres, agnostic, ws, out, out_index, out_count):
    """Sort + NMS + max_det + merge-NMS of general.py:485-508 for one image."""
    _launch("sodt_nms_select", _p(z_img), z_img.shape[1] - 5, _p(keys), int(n_total), C.c_float(iou_thres),
            int(bool(agnostic)), _p(ws), ws.numel() * ws.element_size(), _p(out), _p(out_index), _p(out_count))


def prep_weights(table_dev, n, max_elems, dtype_code):
    _launch("sodt_prep_weights", _p(table_dev), n, max_elems, dtype_code)


def transpose_f32(src, dst, rows, cols, accumulate=0):
    """accumulate: 0 store, 1 add, 2 add and clear src."""
    _launch("sodt_transpose_f32", _p(src), _p(dst), rows, cols, int(accumulate))


def zero_(t):
    _launch("sodt_memset_zero", _p(t), t.numel() * t.element_size())


def cast(src, dst, n):
    _launch("sodt_cast", _p(src), _p(dst), n, dt_code(src), dt_code(dst))


def batch_sum(d, out, B, RC):
    _launch("sodt_batch_sum", _p(d), _p(out), B, RC, dt_code(d))


def sgd_ema_step(p, g, mom, ema, p_cast, group_of_chunk, lr, momentum, weight_decay, nesterov, grad_scale=1.0, ema_decay=0.0):
    """One fused SGD(+nesterov, weight decay groups) + EMA + run-dtype cast pass over the flat buffers (csrc/optim.hip)."""
    ng = len(lr)
    arr = lambda v: (C.c_float * ng)(*[float(x) for x in v])
    code = L.F32 if p_cast is None else dt_code(p_cast)
    _launch("sodt_sgd_ema_step", _p(p), _p(g), _p(mom), _p(ema), _p(p_cast), code, _p(group_of_chunk), p.numel(), ng,
            arr(lr), arr(momentum), arr(weight_decay), int(bool(nesterov)), C.c_float(grad_scale), C.c_float(ema_decay))


def maxpool5_fwd(x, y, argmax, B, H, W, Cc, ldx=None, ldy=None, x_off=0, y_off=0):
    """y = MaxPool2d(5, 1, 2)(x), token-major; x / y may be channel slices of wider tensors (ld*, *_off in elements)."""
    es = x.element_size()
    _launch("sodt_maxpool5_fwd", x.data_ptr() + x_off * es, x.shape[-1] if ldx is None else ldx, y.data_ptr() + y_off * es,
            y.shape[-1] if ldy is None else ldy, _p(argmax), B, H, W, Cc, dt_code(x))


def maxpool5_bwd(dy, argmax, dx, B, H, W, Cc, lddy=None, lddx=None, dy_off=0, dx_off=0, accumulate=False):
    es = dy.element_size()
    _launch("sodt_maxpool5_bwd", dy.data_ptr() + dy_off * es, dy.shape[-1] if lddy is None else lddy, _p(argmax),
            dx.data_ptr() + dx_off * es, dx.shape[-1] if lddx is None else lddx, int(bool(accumulate)), B, H, W, Cc, dt_code(dy))


def gemm_set_variant(v) -> None:
    """0/False: automatic; 1/True: force the K-loop tile kernel; 2: force the A-stationary kernel (tests)."""
    _lib.sodt_gemm_set_variant(int(v))


def version() -> str:
    """Version string of the loaded library; a non-default library path (SODT_LIB_PATH) is part of it."""
    v = _lib.sodt_version().decode()
    ov = L.overrides() if hasattr(L, "overrides") else {}
    return v + (" [" + ", ".join(f"{k}={x}" for k, x in sorted(ov.items())) + "]" if ov else "")
