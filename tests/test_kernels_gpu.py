"""Per-kernel parity on the GPU: every HIP kernel (called through the C ABI) against a
plain PyTorch fp32/fp64 statement of the same op, for the f32 parity path (tight) and
the bf16 throughput path (inputs rounded to bf16 first, looser tolerance)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt):
    return (2e-4, 2e-4) if dt == torch.float32 else (3e-2, 3e-2)


def close(a, b, dt, scale=None, what=""):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    rt, at = tol(dt)
    s = float(b.abs().max()) if scale is None else scale
    err = float((a - b).abs().max())
    assert err <= at * max(s, 1e-6) + 1e-7, f"{what}: max err {err:.3e} vs scale {s:.3e} ({dt})"


def rnd(shape, dev, dt, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)


@pytest.fixture(params=["auto", "tiled", "astat"])
def variant(request, ops):
    """NT GEMMs run through all three kernels: auto (weight-stationary persistent for short K), the K-loop
    tile kernel, and the A-stationary kernel."""
    ops.gemm_set_variant({"auto": 0, "tiled": 1, "astat": 2}[request.param])
    yield request.param
    ops.gemm_set_variant(False)


# ------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 576, 192), (130, 39, 128), (512, 192, 768), (128, 256, 48),
                                   (1000, 200, 384), (77, 768, 192)])
def test_gemm_nt_plain_bias_resid(ops, dev, dt, M, N, K, variant):
    A = rnd((M, K), dev, dt, 1)
    W = rnd((N, K), dev, dt, 2, 1 / math.sqrt(K))
    bias = rnd((N,), dev, torch.float32, 3)
    R = rnd((M, N), dev, dt, 4)
    ref = A.float() @ W.float().t() + bias + R.float()
    if N % 8:   # detect-like width: f32 output, scalar tail path
        out = torch.zeros(M, N, device=dev, dtype=torch.float32)
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias, out_f32=True)
        close(out, ref - R.float(), dt, what="nt f32-out tail")
        return
    out = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias, resid=R)
    close(out, ref, dt, what="nt bias+resid")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_gelu_dgelu_rmod(ops, dev, dt, variant):
    M, N, K = 384, 256, 192
    A = rnd((M, K), dev, dt, 1)
    W = rnd((N, K), dev, dt, 2, 1 / math.sqrt(K))
    bias = rnd((N,), dev, torch.float32, 3)
    pre = torch.zeros(M, N, device=dev, dtype=dt)
    act = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm_nt([ops.SegSpec(A)], W, pre, M, N, K, bias=bias, gelu_out=act)
    ref = A.float() @ W.float().t() + bias
    close(pre, ref, dt, what="pre")
    close(act, F.gelu(ref), dt, what="gelu")
    aux = rnd((M, N), dev, dt, 5)
    out = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, dgelu_aux=aux)
    x = aux.float().double().requires_grad_(True)
    F.gelu(x).sum().backward()
    close(out, (A.float() @ W.float().t()).double() * x.grad, dt, what="dgelu")
    # residual with row modulo (pos_embed broadcast over the batch)
    R = rnd((128, N), dev, dt, 6)
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias, resid=R, rmod=128)
    close(out, ref + R.float().repeat(3, 1), dt, what="rmod")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_stats_affine_detect(ops, dev, dt, variant):
    M, N, K = 1000, 128, 256
    A = rnd((M, K), dev, dt, 1)
    W = rnd((N, K), dev, dt, 2, 1 / math.sqrt(K))
    out = torch.zeros(M, N, device=dev, dtype=dt)
    stats = torch.zeros(16, 2, N, device=dev, dtype=torch.float64)     # SODT_STATS_REPL replicas, summed by bn_finalize
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, stats=stats)
    ref = (A.float() @ W.float().t()).double()
    close(stats.sum(0)[0], ref.sum(0), dt, what="sum")
    close(stats.sum(0)[1], (ref * ref).sum(0), dt, what="sumsq")
    sc = rnd((N,), dev, torch.float32, 7).abs() + 0.5
    sh = rnd((N,), dev, torch.float32, 8)
    ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, affine=(sc, sh))
    close(out, F.silu(ref.float() * sc + sh), dt, what="affine silu")
    # Detect store: (B, na, HW, no)
    B, HW, na, no = 2, 500, 3, 13
    Wd = rnd((48, K), dev, dt, 9, 1 / math.sqrt(K))
    Wd[39:] = 0
    bias = rnd((39,), dev, torch.float32, 10)
    pred = torch.zeros(B, na, HW, no, device=dev, dtype=torch.float32)
    ops.gemm_nt([ops.SegSpec(A)], Wd, pred, M, 39, K, bias=bias, detect=(na, no, HW))
    z = A.float() @ Wd[:39].float().t() + bias
    refp = z.view(B, HW, na, no).permute(0, 2, 1, 3)
    close(pred, refp, dt, what="detect")


@pytest.mark.parametrize("M,N,K,taps", [(2048 + 77, 64, 384, 0), (4096, 64, 576, 9), (256 * 300 + 5, 64, 192, 0), (1024, 32, 576, 0)])
def test_gemm_nt_pipelined_thin_with_bn_statistics(ops, dev, M, N, K, taps):
    """the head's BatchNorm convolutions with N <= 64 (C3 at the stride-4 level, common.py:38-50,76-90) on the pipelined kernel's thin
    instantiation: the output and the f64 column sums of v and v^2 (per-lane f32 partial sums over the workgroup's tiles, one f64 atomic
    per wave - round 5) against f64; a 3x3 tap form, ragged M, several tiles per workgroup (> 256 tiles)"""
    dt = torch.bfloat16
    W = rnd((N, K), dev, dt, 2, 1 / math.sqrt(K))
    if taps:
        Cc = K // taps
        H = int(math.isqrt(M // 4))
        B = 4
        M = B * H * H
        x = rnd((M, Cc), dev, dt, 1)
        segs = [ops.SegSpec(x, Cc, 0, dy, dx, 1, 0, H, H) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
        kw = dict(spatial=(H, H))
        xi = x.float().view(B, H, H, Cc).permute(0, 3, 1, 2)
        wconv = W.float().view(N, 9, Cc).permute(0, 2, 1).reshape(N, Cc, 3, 3)
        ref = F.conv2d(xi.double(), wconv.double(), padding=1).permute(0, 2, 3, 1).reshape(M, N)
    else:
        A = rnd((M, K), dev, dt, 1)
        segs, kw = [ops.SegSpec(A)], {}
        ref = A.double() @ W.double().t()
    out = torch.full((M + 2, N), float("nan"), device=dev, dtype=dt)
    stats = torch.zeros(16, 2, N, device=dev, dtype=torch.float64)
    ops.gemm_nt(segs, W, out[:M], M, N, K, stats=stats, **kw)
    torch.cuda.synchronize()
    assert torch.isnan(out[M:]).all()
    close(out[:M], ref, dt, what="z")
    s1, s2 = stats.sum(0)[0].cpu(), stats.sum(0)[1].cpu()
    r1, r2 = ref.sum(0).cpu(), (ref * ref).sum(0).cpu()
    assert float((s2 - r2).abs().max()) <= 1e-3 * float(r2.abs().max()), "sum of squares"
    assert float((s1 - r1).abs().max()) <= 1e-3 * float(r2.abs().max().sqrt() * math.sqrt(M)), "sum"


def _nhwc(x):   # (B,C,H,W) -> token-major (B*H*W, C)
    B, Cc, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, Cc).contiguous()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("N", [8, 40, 64, 256, 320])
def test_gemm_nt_relu_epilogues_and_partial_column_tiles(ops, dev, dt, N):
    """SODT_EPI_RELU / SODT_EPI_DRELU (the SR branch's convolutions, sr.py) and, in bf16 with K >= 512, the pipelined kernel on an N
    that is not a multiple of its 192-column tile (rows >= N of W come from the zero page, their chunks are not stored): 3x3 taps
    over a (2, 24, 20, 64) image (K = 576), a row tail (M = 960), more column tiles than one when N = 256 / 320; the output buffer is
    wider than N and its other columns must stay untouched."""
    B, H, Wd, Cc = 2, 24, 20, 64
    M, K = B * H * Wd, 9 * Cc
    x = rnd((B, Cc, H, Wd), dev, dt, 1)
    w4 = rnd((N, Cc, 3, 3), dev, dt, 2, 1 / math.sqrt(K))
    Wg = w4.permute(0, 2, 3, 1).reshape(N, K).contiguous()                 # [n][tap * Cin + c]
    bias = rnd((N,), dev, torch.float32, 3)
    xt = _nhwc(x)
    segs = [ops.SegSpec(xt, Cc, 0, dy, dx, 1, 0, H, Wd) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    ref = _nhwc(F.conv2d(x.float(), w4.float(), None, padding=1))
    ld = N + 16
    for mode in ("relu", "bias_relu", "drelu", "bias_resid"):
        out = torch.full((M, ld), 7.0, device=dev, dtype=dt)
        aux = rnd((M, N + 8), dev, dt, 5)                                  # (its own leading dimension and a column offset)
        kw = dict(spatial=(H, Wd), ldc=ld, c_off=8)
        if mode == "relu":
            ops.gemm_nt(segs, Wg, out, M, N, K, relu=True, **kw); want = F.relu(ref)
        elif mode == "bias_relu":
            ops.gemm_nt(segs, Wg, out, M, N, K, bias=bias, relu=True, **kw); want = F.relu(ref + bias)
        elif mode == "drelu":
            ops.gemm_nt(segs, Wg, out, M, N, K, drelu_aux=aux, aux_off=8, **kw); want = ref * (aux[:, 8:].float() > 0)
        else:
            ops.gemm_nt(segs, Wg, out, M, N, K, bias=bias, resid=aux, ldr=N + 8, r_off=8, **kw); want = ref + bias + aux[:, 8:].float()
        torch.cuda.synchronize()
        close(out[:, 8:8 + N], want, dt, what=f"{mode} N={N}")
        assert bool((out[:, :8] == 7.0).all()) and bool((out[:, 8 + N:] == 7.0).all()), f"{mode}: columns outside [8, 8 + N) were written"


def test_gemm_nt_pipelined_bf16(ops, dev):
    """The LDS-DMA pipelined bf16 kernel (csrc/gemm3.hip: N % 192 == 0, K >= 384): every epilogue it is built
    for, a row tail, more tiles than workgroups (persistent walk), multi-segment and conv-tap A operands."""
    dt = torch.bfloat16
    ops.gemm_set_variant(0)
    for (M, N, K) in [(256 * 130 + 100, 384, 768), (1000, 192, 384), (4096, 1152, 1536), (70000, 576, 192), (3000, 768, 192)]:
        A = rnd((M, K), dev, dt, 1)
        W = rnd((N, K), dev, dt, 2, 1 / math.sqrt(K))
        bias = rnd((N,), dev, torch.float32, 3)
        R = rnd((M, N), dev, dt, 4)
        aux = rnd((M, N), dev, dt, 5)
        base = A.float() @ W.float().t()
        out = torch.zeros(M, N, device=dev, dtype=dt)
        act = torch.zeros(M, N, device=dev, dtype=dt)
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K)
        close(out, base, dt, what=f"nt3 plain {M}x{N}x{K}")
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias)
        close(out, base + bias, dt, what="nt3 bias")
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, resid=R)
        close(out, base + R.float(), dt, what="nt3 resid")
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias, resid=R)
        close(out, base + bias + R.float(), dt, what="nt3 bias+resid")
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, bias=bias, gelu_out=act)
        close(out, base + bias, dt, what="nt3 pre")
        close(act, F.gelu(base + bias), dt, what="nt3 gelu")
        ops.gemm_nt([ops.SegSpec(A)], W, out, M, N, K, dgelu_aux=aux)
        x = aux.float().double().requires_grad_(True)
        F.gelu(x).sum().backward()
        close(out, base.double() * x.grad, dt, what="nt3 dgelu")
    # two concatenated K-segments with different leading dimensions
    M, N = 2000, 192
    A1f = rnd((M, 320), dev, dt, 6)             # 256 of 320 columns: leading dimension != segment length
    A1 = A1f[:, :256]
    A2 = rnd((M, 128), dev, dt, 7)
    W = rnd((N, 384), dev, dt, 8, 0.05)
    out = torch.zeros(M, N, device=dev, dtype=dt)
    ops.gemm_nt([ops.SegSpec(A1f, 256, 0), ops.SegSpec(A2)], W, out, M, N, 384)
    close(out, torch.cat([A1.float(), A2.float()], 1) @ W.float().t(), dt, what="nt3 concat")
    # 2x2 conv (right/bottom zero pad) and its input gradient, C = 192 (the stage-1 conv MLP shapes)
    B, H, Wd, Ci, Co = 2, 24, 40, 192, 192
    x = rnd((B, Ci, H, Wd), dev, dt, 9)
    xt = _nhwc(x)
    M = B * H * Wd
    w2 = rnd((Co, Ci, 2, 2), dev, dt, 10, 0.05)
    bias = rnd((Co,), dev, torch.float32, 11)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w2.float(), bias)
    wg = w2.permute(0, 2, 3, 1).reshape(Co, 4 * Ci).contiguous()
    segs = [ops.SegSpec(xt, Ci, 0, dy, dx, 1, 0, H, Wd) for dy in (0, 1) for dx in (0, 1)]
    out = torch.zeros(M, Co, device=dev, dtype=dt)
    ops.gemm_nt(segs, wg, out, M, Co, 4 * Ci, spatial=(H, Wd), bias=bias)
    close(out, _nhwc(ref), dt, what="nt3 conv2x2")
    dy_ = rnd((B, Co, H, Wd), dev, dt, 12)
    xr = x.float().clone().requires_grad_(True)
    F.conv2d(F.pad(xr, (0, 1, 0, 1)), w2.float(), bias).backward(dy_.float())
    wgt = w2.permute(1, 2, 3, 0).reshape(Ci, 4 * Co).contiguous()
    dyt = _nhwc(dy_)
    segs = [ops.SegSpec(dyt, Co, 0, -dy, -dx, 1, 0, H, Wd) for dy in (0, 1) for dx in (0, 1)]
    dx_ = torch.zeros(M, Ci, device=dev, dtype=dt)
    ops.gemm_nt(segs, wgt, dx_, M, Ci, 4 * Co, spatial=(H, Wd))
    close(dx_, _nhwc(xr.grad), dt, what="nt3 conv2x2 dx")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_conv_taps(ops, dev, dt, variant):
    """2x2 conv with right/bottom zero pad (Mlp conv variant), 3x3 same conv, and their input gradients."""
    B, H, W, Ci, Co = 2, 12, 20, 64, 128
    x = rnd((B, Ci, H, W), dev, dt, 1)
    xt = _nhwc(x)
    M = B * H * W
    # ---- 2x2, F.pad(0,1,0,1)
    w2 = rnd((Co, Ci, 2, 2), dev, dt, 2, 0.1)
    bias = rnd((Co,), dev, torch.float32, 3)
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w2.float(), bias)
    wg = w2.permute(0, 2, 3, 1).reshape(Co, 4 * Ci).contiguous()
    segs = [ops.SegSpec(xt, Ci, 0, dy, dx, 1, 0, H, W) for dy in (0, 1) for dx in (0, 1)]
    out = torch.zeros(M, Co, device=dev, dtype=dt)
    ops.gemm_nt(segs, wg, out, M, Co, 4 * Ci, spatial=(H, W), bias=bias)
    close(out, _nhwc(ref), dt, what="conv2x2")
    # input gradient of the 2x2 conv: taps negated, weights [ci][tap*Co+co]
    dy_ = rnd((B, Co, H, W), dev, dt, 4)
    xr = x.float().clone().requires_grad_(True)
    F.conv2d(F.pad(xr, (0, 1, 0, 1)), w2.float(), bias).backward(dy_.float())
    wgt = w2.permute(1, 2, 3, 0).reshape(Ci, 4 * Co).contiguous()
    dyt = _nhwc(dy_)
    segs = [ops.SegSpec(dyt, Co, 0, -dy, -dx, 1, 0, H, W) for dy in (0, 1) for dx in (0, 1)]
    dx_ = torch.zeros(M, Ci, device=dev, dtype=dt)
    ops.gemm_nt(segs, wgt, dx_, M, Ci, 4 * Co, spatial=(H, W))
    close(dx_, _nhwc(xr.grad), dt, what="conv2x2 dx")
    # ---- 3x3 pad 1
    w3 = rnd((Co, Ci, 3, 3), dev, dt, 5, 0.05)
    ref3 = F.conv2d(x.float(), w3.float(), None, padding=1)
    wg3 = w3.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous()
    segs = [ops.SegSpec(xt, Ci, 0, dy - 1, dx - 1, 1, 0, H, W) for dy in range(3) for dx in range(3)]
    ops.gemm_nt(segs, wg3, out, M, Co, 9 * Ci, spatial=(H, W))
    close(out, _nhwc(ref3), dt, what="conv3x3")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Cc,H,W", [(192, 32, 32), (384, 16, 32)])
def test_gemm_nt_patch_merge_scatter_pipelined_sizes(ops, dev, dt, Cc, H, W):
    """PatchMerging input gradient (4 GEMMs with output-row scatter, backbone_vit.py:850-857 backward) at sizes the
    pipelined bf16 NT kernel takes (N % 192 == 0, M >= 256): the scatter epilogue of gemm_nt3_kernel<0, OSC>."""
    B = 2
    M2 = B * (H // 2) * (W // 2)
    Wr = rnd((2 * Cc, 4 * Cc), dev, dt, 2, 0.05)
    dz = rnd((M2, 2 * Cc), dev, dt, 3)
    WrT = Wr.t().contiguous()                       # [4C][2C]
    dx = torch.zeros(B * H * W, Cc, device=dev, dtype=dt)
    for tap, (dy, dxx) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        ops.gemm_nt([ops.SegSpec(dz, 2 * Cc, 0, 0, 0, 1, 0, H // 2, W // 2)], WrT, dx, M2, Cc, 2 * Cc,
                    spatial=(H // 2, W // 2), w_off=tap * Cc * 2 * Cc, oscatter=(2, dy, dxx, H, W))
    full = dz.float() @ Wr.float()                  # [M2][4C]
    refdx = torch.zeros(B, H, W, Cc, device=dev)
    f4 = full.view(B, H // 2, W // 2, 4, Cc)
    refdx[:, 0::2, 0::2] = f4[..., 0, :]
    refdx[:, 1::2, 0::2] = f4[..., 1, :]
    refdx[:, 0::2, 1::2] = f4[..., 2, :]
    refdx[:, 1::2, 1::2] = f4[..., 3, :]
    close(dx, refdx.view(-1, Cc), dt, what="patch merge dx (pipelined sizes)")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_nt_merge_upsample_scatter(ops, dev, dt, variant):
    B, H, W, Cc = 2, 8, 12, 64
    x = rnd((B, H, W, Cc), dev, dt, 1)
    xt = x.reshape(-1, Cc)
    # PatchMerging gather (0,0),(1,0),(0,1),(1,1) then Linear(4C -> 2C)
    Wr = rnd((2 * Cc, 4 * Cc), dev, dt, 2, 0.05)
    xf = x.float()
    cat = torch.cat([xf[:, 0::2, 0::2], xf[:, 1::2, 0::2], xf[:, 0::2, 1::2], xf[:, 1::2, 1::2]], -1).reshape(-1, 4 * Cc)
    ref = cat @ Wr.float().t()
    M2 = B * (H // 2) * (W // 2)
    segs = [ops.SegSpec(xt, Cc, 0, dy, dx, 2, 0, H, W) for (dy, dx) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    out = torch.zeros(M2, 2 * Cc, device=dev, dtype=dt)
    ops.gemm_nt(segs, Wr, out, M2, 2 * Cc, 4 * Cc, spatial=(H // 2, W // 2))
    close(out, ref, dt, what="patch merge")
    # its input gradient: 4 GEMMs with output-row scatter
    dz = rnd((M2, 2 * Cc), dev, dt, 3)
    WrT = Wr.t().contiguous()                       # [4C][2C]
    dx = torch.zeros(B * H * W, Cc, device=dev, dtype=dt)
    for tap, (dy, dxx) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        ops.gemm_nt([ops.SegSpec(dz, 2 * Cc, 0, 0, 0, 1, 0, H // 2, W // 2)], WrT, dx, M2, Cc, 2 * Cc,
                    spatial=(H // 2, W // 2), w_off=tap * Cc * 2 * Cc,
                    oscatter=(2, dy, dxx, H, W))
    full = dz.float() @ Wr.float()                  # [M2][4C]
    refdx = torch.zeros(B, H, W, Cc, device=dev)
    f4 = full.view(B, H // 2, W // 2, 4, Cc)
    refdx[:, 0::2, 0::2] = f4[..., 0, :]
    refdx[:, 1::2, 0::2] = f4[..., 1, :]
    refdx[:, 0::2, 1::2] = f4[..., 2, :]
    refdx[:, 1::2, 1::2] = f4[..., 3, :]
    close(dx, refdx.view(-1, Cc), dt, what="patch merge dx")
    # upsample(x2, nearest) + concat as two segments of a 1x1 conv
    lo = rnd((B, H // 2, W // 2, 128), dev, dt, 4)
    Wc = rnd((64, 128 + Cc), dev, dt, 5, 0.05)
    up = lo.float().repeat_interleave(2, 1).repeat_interleave(2, 2)
    ref = torch.cat([up, xf], -1).reshape(-1, 128 + Cc) @ Wc.float().t()
    segs = [ops.SegSpec(lo.reshape(-1, 128), 128, 0, 0, 0, 1, 1, H // 2, W // 2), ops.SegSpec(xt, Cc, 0, 0, 0, 1, 0, H, W)]
    out = torch.zeros(B * H * W, 64, device=dev, dtype=dt)
    ops.gemm_nt(segs, Wc, out, B * H * W, 64, 128 + Cc, spatial=(H, W))
    close(out, ref, dt, what="upsample+concat")


# ------------------------------------------------------------------ GEMM TN
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("M,N,K,splits", [(512, 128, 128, 1), (1000, 192, 192, 3), (4096, 576, 192, None), (700, 48, 128, 2)])
def test_gemm_tn(ops, dev, dt, M, N, K, splits, variant):
    dY = rnd((M, N), dev, dt, 1)
    X = rnd((M, K), dev, dt, 2)
    dW = torch.zeros(N, K, device=dev, dtype=torch.float32)
    db = torch.zeros(N, device=dev, dtype=torch.float32)
    ops.gemm_tn(dY, [ops.SegSpec(X)], dW, M, N, K, dbias=db, splits=splits)
    ref = dY.float().t().double() @ X.float().double()
    close(dW, ref, dt, what="tn dW")
    close(db, dY.float().double().sum(0), dt, what="tn dbias")
    ops.gemm_tn(dY, [ops.SegSpec(X)], dW, M, N, K, splits=splits)   # accumulates
    close(dW, 2 * ref, dt, what="tn accumulate")


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_tn_conv_weight_grad(ops, dev, dt, variant):
    B, H, W, Ci, Co = 2, 12, 20, 64, 128
    x = rnd((B, Ci, H, W), dev, dt, 1)
    dy_ = rnd((B, Co, H, W), dev, dt, 2)
    w2 = torch.zeros(Co, Ci, 2, 2, device=dev, requires_grad=True)
    F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w2).backward(dy_.float())
    xt, dyt = _nhwc(x), _nhwc(dy_)
    segs = [ops.SegSpec(xt, Ci, 0, dy, dx, 1, 0, H, W) for dy in (0, 1) for dx in (0, 1)]
    dW = torch.zeros(Co, Ci, 2, 2, device=dev, dtype=torch.float32)
    ops.gemm_tn(dyt, segs, dW, B * H * W, Co, 4 * Ci, spatial=(H, W), kperm=(Ci, 4))
    close(dW, w2.grad, dt, what="conv2x2 dW (torch layout)")


def test_gemm_nt_gelu_only_and_dgelu_recompute(ops, dev):
    """Linear-GELU-Linear with only the activation kept (bf16 pipelined kernel): SODT_EPI_GELU writes GELU(xn W1^T + b1);
    SODT_EPI_DGELU_RC computes dh = (dy W2) * gelu'(xn W1^T + b1) from A = [xn | dy], W = [W1 | W2^T], recomputing the
    pre-activation in the first K half."""
    dt = torch.bfloat16
    ops.gemm_set_variant(0)
    for (M, Cc) in [(3000, 192), (256 * 9 + 7, 384), (1024, 768)]:
        xn = rnd((M, Cc), dev, dt, 1)
        dy = rnd((M, Cc), dev, dt, 2)
        W1 = rnd((4 * Cc, Cc), dev, dt, 3, 1 / math.sqrt(Cc))
        W2 = rnd((Cc, 4 * Cc), dev, dt, 4, 1 / math.sqrt(4 * Cc))
        b1 = rnd((4 * Cc,), dev, torch.float32, 5)
        assert ops.mlp_recompute_ok(M, Cc, dt)
        act = torch.zeros(M, 4 * Cc, device=dev, dtype=dt)
        ops.gemm_nt([ops.SegSpec(xn)], W1, act, M, 4 * Cc, Cc, bias=b1, gelu_only=True)
        h = (xn.float() @ W1.float().t() + b1).double().requires_grad_(True)
        close(act, F.gelu(h), dt, what=f"gelu only C={Cc}")
        F.gelu(h).sum().backward()
        Wcat = torch.cat([W1, W2.t().contiguous()], 1).contiguous()
        dh = torch.zeros(M, 4 * Cc, device=dev, dtype=dt)
        ops.gemm_nt([ops.SegSpec(xn), ops.SegSpec(dy)], Wcat, dh, M, 4 * Cc, 2 * Cc, bias=b1, dgelu_rc=True)
        close(dh, (dy.float() @ W2.float()).double() * h.grad, dt, what=f"dgelu recompute C={Cc}")
    # a shape the pipelined kernel does not take must be refused, not silently mis-computed
    xn = rnd((512, 128), dev, dt, 1); W = rnd((512, 256), dev, dt, 2); out = torch.zeros(512, 512, device=dev, dtype=dt)
    with pytest.raises(RuntimeError):
        ops.gemm_nt([ops.SegSpec(xn), ops.SegSpec(xn)], W, out, 512, 512, 256, bias=torch.zeros(512, device=dev), dgelu_rc=True)


def test_gemm_tn_pipelined_bf16(ops, dev):
    """The LDS-DMA pipelined bf16 weight-gradient kernel (csrc/gemm3.hip): both tile orientations, N / K tails,
    ragged M slices, dbias through the ones-MFMA, conv taps with the torch-layout K permutation."""
    dt = torch.bfloat16
    ops.gemm_set_variant(0)
    for (M, N, K, splits) in [(4096, 192, 768, None), (5000, 384, 384, 7), (2048 + 17, 576, 192, 3), (3000, 768, 192, None),
                              (4096, 1152, 384, 5), (2000, 136, 200, 2), (8192, 1536, 384, None), (1024, 64, 64, 1)]:
        dY = rnd((M, N), dev, dt, 1)
        X = rnd((M, K), dev, dt, 2)
        ref = dY.float().double().t() @ X.float().double()
        for scratch in (None, torch.full((4 << 20,), float("nan"), device=dev)):   # atomics, then per-slice partial tiles
            ops.set_tn_scratch(scratch)
            dW = torch.ones(N, K, device=dev, dtype=torch.float32)
            db = torch.ones(N, device=dev, dtype=torch.float32)
            ops.gemm_tn(dY, [ops.SegSpec(X)], dW, M, N, K, dbias=db, splits=splits)
            close(dW - 1, ref, dt, what=f"tn3 dW {M}x{N}x{K} scratch={scratch is not None}")
            close(db - 1, dY.float().double().sum(0), dt, what=f"tn3 dbias {M}x{N}x{K}")
        ops.set_tn_scratch(None)
    # 2x2 conv weight gradient, C = 192 (stage-1 conv MLP), torch (Co, Ci, 2, 2) layout via kperm
    B, H, Wd, Ci, Co = 2, 24, 40, 192, 192
    x = rnd((B, Ci, H, Wd), dev, dt, 3)
    dy_ = rnd((B, Co, H, Wd), dev, dt, 4)
    w2 = (rnd((Co, Ci, 2, 2), dev, dt, 5, 0.05)).float().requires_grad_(True)
    F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w2).backward(dy_.float())
    xt, dyt = _nhwc(x), _nhwc(dy_)
    segs = [ops.SegSpec(xt, Ci, 0, dy, dx, 1, 0, H, Wd) for dy in (0, 1) for dx in (0, 1)]
    for scratch in (None, torch.full((4 << 20,), float("nan"), device=dev)):
        ops.set_tn_scratch(scratch)
        dW = torch.zeros(Co, Ci, 2, 2, device=dev, dtype=torch.float32)
        ops.gemm_tn(dyt, segs, dW, B * H * Wd, Co, 4 * Ci, spatial=(H, Wd), kperm=(Ci, 4))
        close(dW, w2.grad, dt, what="tn3 conv2x2 dW")
    ops.set_tn_scratch(None)
    # 3x3 conv (head) with a two-tensor concat input
    Ci1, Ci2, Co = 64, 128, 128
    x1 = rnd((B, Ci1, H, Wd), dev, dt, 6)
    x2 = rnd((B, Ci2, H, Wd), dev, dt, 7)
    dy_ = rnd((B, Co, H, Wd), dev, dt, 8)
    w3 = (rnd((Co, Ci1 + Ci2, 3, 3), dev, dt, 9, 0.05)).float().requires_grad_(True)
    F.conv2d(torch.cat([x1, x2], 1).float(), w3, padding=1).backward(dy_.float())
    x1t, x2t, dyt = _nhwc(x1), _nhwc(x2), _nhwc(dy_)
    segs = []
    for dy in range(3):
        for dx in range(3):
            segs.append(ops.SegSpec(x1t, Ci1, 0, dy - 1, dx - 1, 1, 0, H, Wd))
    Cin = Ci1
    dW = torch.zeros(Co, Cin, 3, 3, device=dev, dtype=torch.float32)
    ops.gemm_tn(dyt, segs, dW, B * H * Wd, Co, 9 * Cin, spatial=(H, Wd), kperm=(Cin, 9))
    w3b = (w3.detach()[:, :Ci1]).clone().requires_grad_(True)
    F.conv2d(x1.float(), w3b, padding=1).backward(dy_.float())
    close(dW, w3b.grad, dt, what="tn3 conv3x3 dW")


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Cc", [64, 128, 192, 384, 768])
def test_layernorm(ops, dev, dt, Cc):
    M = 777
    x = rnd((M, Cc), dev, dt, 1) * 2 + 0.5
    g = rnd((Cc,), dev, torch.float32, 2) * 0.2 + 1
    b = rnd((Cc,), dev, torch.float32, 3) * 0.2
    y = torch.zeros_like(x)
    st = torch.zeros(M, 2, device=dev)
    ops.layernorm_fwd(x, g, b, y, st, M, Cc)
    xr = x.float().double().requires_grad_(True)
    gr = g.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    ref = F.layer_norm(xr, (Cc,), gr, br, 1e-5)
    close(y, ref, dt, what="ln fwd")
    dy = rnd((M, Cc), dev, dt, 4)
    dres = rnd((M, Cc), dev, dt, 5)
    ref.backward(dy.float().double())
    dx = torch.zeros_like(x)
    dg = torch.zeros(Cc, device=dev)
    db = torch.zeros(Cc, device=dev)
    ops.layernorm_bwd(dy, x, st, g, dres, dx, dg, db, M, Cc)
    close(dx, xr.grad + dres.float().double(), dt, what="ln dx")
    close(dg, gr.grad, dt, what="ln dgamma")
    close(db, br.grad, dt, what="ln dbeta")


# ------------------------------------------------------------------ window attention
def _attn_ref(qkv, table, B, H, W, Cc, heads, ws, shift):
    """fp64 torch statement of roll/partition/attention/unpartition/roll (oracle/ref_torch.py:window_attention)."""
    from oracle import ref_torch as R
    hd = Cc // heads
    x = qkv.view(B, H, W, 3 * Cc)
    if shift:
        x = torch.roll(x, (-shift, -shift), (1, 2))
    xw = R.window_partition(x, ws).view(-1, ws * ws, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = xw[0] * hd ** -0.5, xw[1], xw[2]
    attn = q @ k.transpose(-2, -1)
    idx = R.relative_position_index(ws).view(-1).to(qkv.device)
    N = ws * ws
    attn = attn + table[idx].view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    if shift:
        mask = R.shift_mask(H, W, ws, shift, torch.float64).to(qkv.device)
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, ws, ws, Cc)
    o = R.window_unpartition(o, ws, H, W)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    return o.reshape(B * H * W, Cc)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Cc,H,W,ws,shift", [(192, 16, 24, 8, 0), (192, 16, 24, 8, 2), (384, 16, 16, 8, 2),
                                               (768, 32, 32, 32, 0), (768, 16, 32, 16, 0), (768, 32, 64, 32, 16),
                                               (384, 32, 32, 16, 0), (768, 64, 32, 32, 0)])
def test_window_attention(ops, dev, dt, Cc, H, W, ws, shift):
    B, heads = 2, 12
    M = B * H * W
    L2 = 2 * ws - 1
    qkv = rnd((M, 3 * Cc), dev, dt, 1)
    table = rnd((L2 * L2, heads), dev, torch.float32, 2) * 0.5
    bias_t = table.t().contiguous()
    out = torch.zeros(M, Cc, device=dev, dtype=dt)
    lse = torch.zeros(M, heads, device=dev)
    ops.window_attn_fwd(qkv, bias_t, out, lse, B, H, W, Cc, heads, ws, shift)
    qr = qkv.float().double().requires_grad_(True)
    tr = table.double().requires_grad_(True)
    ref = _attn_ref(qr, tr, B, H, W, Cc, heads, ws, shift)
    close(out, ref, dt, what="attn fwd")
    dout = rnd((M, Cc), dev, dt, 3)
    ref.backward(dout.float().double())
    dqkv = torch.zeros_like(qkv)
    dbt = torch.zeros_like(bias_t)
    scratch = torch.zeros(M * Cc + M * heads, device=dev) if ws * ws > 64 else None
    ops.window_attn_bwd(qkv, bias_t, out, dout, lse, dqkv, dbt, scratch, B, H, W, Cc, heads, ws, shift)
    close(dqkv, qr.grad, dt, what="attn dqkv")
    close(dbt.t(), tr.grad, dt, what="attn dbias")


# ------------------------------------------------------------------ front end
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B,S", [(2, 96), (3, 352)])
def test_frontend(ops, dev, dt, B, S):
    from oracle import ref_torch as R
    sd = {k: v.to(dev) for k, v in R.procedural_state_dict(S, 8).items() if "channel_embed" in k or "chan_block" in k}
    x_rgb, x_ir = R.synthetic_inputs(B, S, seed=3)
    x_rgb, x_ir = x_rgb.to(dev), x_ir.to(dev)
    w = torch.stack([sd[f"image_encoder.channel_embed_{c}.proj.weight"].view(48, 16) for c in "rgbi"]).contiguous()
    b = torch.stack([sd[f"image_encoder.channel_embed_{c}.proj.bias"] for c in "rgbi"]).contiguous()
    g = torch.stack([sd[f"image_encoder.chan_block.norm{i}.weight"] for i in range(1, 5)]).contiguous()
    be = torch.stack([sd[f"image_encoder.chan_block.norm{i}.bias"] for i in range(1, 5)]).contiguous()
    t = S // 4
    out = torch.zeros(B * t * t, 192, device=dev, dtype=dt)
    ir_plane = x_ir[:, 0]
    ops.frontend_fwd(x_rgb, ir_plane, 3 * S * S, w, b, g, be, out, B, S)
    sdd = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    x4 = torch.cat([x_rgb, x_ir[:, 0:1]], 1).double()
    r, gg, bb, ii = R.channel_embeds(sdd, x4)
    ref = torch.cat(R.cattention_block(sdd, r, gg, bb, ii), -1).reshape(-1, 192)
    close(out, ref, dt, what="frontend fwd")
    dout = rnd((B * t * t, 192), dev, dt, 5)
    ref.backward(dout.float().double())
    # direct atomics (no workspace) and the two-stage reduction through a workspace (filled with junk: it need not be zeroed)
    for ws in (None, torch.full((ops.frontend_bwd_workspace_bytes(B, S) // 4,), 7.0, device=dev)):
        dw, db, dg, dbe = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(g), torch.zeros_like(be)
        ops.frontend_bwd(x_rgb, ir_plane, 3 * S * S, w, b, g, be, dout, dw, db, dg, dbe, B, S, 1, ws)
        for ci, c in enumerate("rgbi"):
            close(dw[ci], sdd[f"image_encoder.channel_embed_{c}.proj.weight"].grad.view(48, 16), dt, what=f"dw {c}")
            close(db[ci], sdd[f"image_encoder.channel_embed_{c}.proj.bias"].grad, dt, what=f"db {c}")
            close(dg[ci], sdd[f"image_encoder.chan_block.norm{ci + 1}.weight"].grad, dt, what=f"dgamma {ci}")
            close(dbe[ci], sdd[f"image_encoder.chan_block.norm{ci + 1}.bias"].grad, dt, what=f"dbeta {ci}")


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("ws,shift", [(1, 0), (2, 0), (2, 1), (4, 2), (8, 3)])
def test_cross_channel_attention_general(ops, dev, dt, ws, shift):
    """window >= 1 / shifted form of CAttentionBlock (the reference ships window 1) against the oracle."""
    from oracle import ref_torch as R
    B, S = 2, 64
    sd = {k: v.to(dev) for k, v in R.procedural_state_dict(S, 8).items() if "channel_embed" in k or "chan_block" in k}
    x_rgb, x_ir = R.synthetic_inputs(B, S, seed=7)
    x_rgb, x_ir = x_rgb.to(dev), x_ir.to(dev)
    w = torch.stack([sd[f"image_encoder.channel_embed_{c}.proj.weight"].view(48, 16) for c in "rgbi"]).contiguous() * 3
    b = torch.stack([sd[f"image_encoder.channel_embed_{c}.proj.bias"] for c in "rgbi"]).contiguous()
    g = torch.stack([sd[f"image_encoder.chan_block.norm{i}.weight"] for i in range(1, 5)]).contiguous()
    be = torch.stack([sd[f"image_encoder.chan_block.norm{i}.bias"] for i in range(1, 5)]).contiguous()
    t = S // 4
    M = B * t * t
    e = torch.zeros(M, 192, device=dev)
    out = torch.zeros(M, 192, device=dev, dtype=dt)
    ops.patch_embed4_fwd(x_rgb, x_ir, 3 * S * S, w, b, e, B, S)
    ops.cross_attn_ln_fwd(e, g, be, out, B, S, ws, shift)
    sdd = {k: v.double().cpu() for k, v in sd.items()}      # oracle on the CPU (its mask helper builds CPU tensors)
    for ci, c in enumerate("rgbi"):
        sdd[f"image_encoder.channel_embed_{c}.proj.weight"] = (w[ci].double().cpu().view(48, 1, 4, 4))
    sdd = {k: v.requires_grad_(True) for k, v in sdd.items()}
    x4 = torch.cat([x_rgb, x_ir[:, 0:1]], 1).double().cpu()
    planes = R.channel_embeds(sdd, x4)
    close(e, torch.cat(planes, -1).reshape(M, 192), torch.float32, what="embeds")
    ref = torch.cat(R.cattention_block(sdd, *planes, window_size=ws, shift=shift), -1).reshape(M, 192)
    close(out, ref, dt, what="cross attn fwd")
    dout = rnd((M, 192), dev, dt, 9)
    ref.backward(dout.float().double().cpu())
    de = torch.zeros(M, 192, device=dev)
    dw, db, dg, dbe = torch.zeros_like(w), torch.zeros_like(b), torch.zeros_like(g), torch.zeros_like(be)
    ops.cross_attn_ln_bwd(e, g, dout, de, dg, dbe, B, S, ws, shift)
    ops.patch_embed4_bwd(x_rgb, x_ir, 3 * S * S, de, dw, db, B, S)
    for ci, c in enumerate("rgbi"):
        close(dw[ci], sdd[f"image_encoder.channel_embed_{c}.proj.weight"].grad.view(48, 16), dt, what=f"dw {c}")
        close(db[ci], sdd[f"image_encoder.channel_embed_{c}.proj.bias"].grad, dt, what=f"db {c}")
        close(dg[ci], sdd[f"image_encoder.chan_block.norm{ci + 1}.weight"].grad, dt, what=f"dgamma {ci}")
        close(dbe[ci], sdd[f"image_encoder.chan_block.norm{ci + 1}.bias"].grad, dt, what=f"dbeta {ci}")


# ------------------------------------------------------------------ BN + SiLU, copies, detect
@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("Cc", [64, 128, 512])
def test_bn_silu(ops, dev, dt, Cc):
    M = 3000
    z = rnd((M, Cc), dev, dt, 1) * 1.5 + 0.3
    g = rnd((Cc,), dev, torch.float32, 2) * 0.2 + 1
    b = rnd((Cc,), dev, torch.float32, 3) * 0.2
    zd = z.float().double()
    stats = torch.zeros(16, 2, Cc, device=dev, dtype=torch.float64)     # SODT_STATS_REPL replicas: split the sums over two of them
    stats[3] = torch.stack([zd[:1000].sum(0), (zd[:1000] * zd[:1000]).sum(0)])
    stats[11] = torch.stack([zd[1000:].sum(0), (zd[1000:] * zd[1000:]).sum(0)])
    mr = torch.zeros(2, Cc, device=dev)
    rm = torch.zeros(Cc, device=dev)
    rv = torch.ones(Cc, device=dev)
    ops.bn_finalize(stats, mr, rm, rv, M, Cc, 1e-3, 0.03)
    close(mr[0], zd.mean(0), torch.float32, what="mean")
    close(mr[1], 1 / torch.sqrt(zd.var(0, unbiased=False) + 1e-3), torch.float32, what="rstd")
    close(rv, 0.97 + 0.03 * zd.var(0, unbiased=True), torch.float32, what="running var")
    yv = torch.zeros(M, Cc, device=dev, dtype=dt)
    ops.bn_silu_fwd(z, mr, g, b, yv, Cc, M, Cc)
    zr = zd.clone().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.silu(F.batch_norm(zr, None, None, gr, br, True, 0.0, 1e-3))
    close(yv, ref, dt, what="bn silu fwd")
    dy = rnd((M, Cc), dev, dt, 4)
    ref.backward(dy.float().double())
    red = torch.zeros(2, Cc, device=dev, dtype=torch.float64)
    ops.bn_silu_bwd_reduce(dy, Cc, z, mr, g, b, red, M, Cc)
    dz = torch.zeros_like(z)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    ops.bn_silu_bwd_apply(dy, Cc, z, mr, g, b, red, dz, dg, db, M, Cc)
    close(dz, zr.grad, dt, what="bn dz")
    close(dg, gr.grad, dt, what="bn dgamma")
    close(db, br.grad, dt, what="bn dbeta")


@pytest.mark.parametrize("dt", DTYPES)
def test_copy_gather_detect(ops, dev, dt):
    B, H, W, Cc = 2, 6, 10, 64
    src = rnd((B, H, W, Cc), dev, dt, 1)
    dst = torch.zeros(B, 2 * H, 2 * W, Cc + 32, device=dev, dtype=dt)
    ops.copy_rows(src, Cc, dst, Cc + 32, B, 2 * H, 2 * W, 1, Cc, dst_off=32)
    ref = src.repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(dst[..., 32:], ref) and float(dst[..., :32].abs().max()) == 0
    d = rnd((B, 2 * H, 2 * W, Cc + 32), dev, dt, 2)
    ds = torch.zeros(B, H, W, Cc, device=dev, dtype=dt)
    ops.gather_sum_rows(d, Cc + 32, ds, Cc, B, H, W, 1, Cc, d_off=32)
    refs = d[..., 32:].float().view(B, H, 2, W, 2, Cc).sum((2, 4))
    close(ds, refs, dt, what="gather sum")
    HW, na, no = 50, 3, 13
    dpred = rnd((B, na, HW, no), dev, torch.float32, 3)
    dz = torch.ones(B * HW, 48, device=dev, dtype=dt)
    ops.detect_unpermute(dpred, dz, 48, B, HW, na, no)
    refz = dpred.permute(0, 2, 1, 3).reshape(B * HW, 39)
    close(dz[:, :39], refz, dt, what="detect unpermute")
    assert float(dz[:, 39:].abs().max()) == 0
    from oracle import ref_torch as R
    raw = rnd((B, 3, 5, 7, 13), dev, torch.float32, 4)
    ag = torch.tensor(R.ANCHORS_PX, device=dev).view(-1)
    zz = torch.zeros(B, 3 * 5 * 7, 13, device=dev)
    ops.detect_decode(raw, ag, zz, B, 3, 5, 7, 13, 4.0)
    refd = R.detect_decode(raw.cpu(), torch.tensor(R.ANCHORS_PX)).to(dev)
    close(zz, refd, torch.float32, what="decode")
