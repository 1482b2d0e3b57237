"""EDSR's closing convolution (edsr.py:81-84: nn.Conv2d(64, ch, 3, padding=1)) on its own kernels (csrc/conv3.hip, through the C ABI):
forward, input gradient and weight / bias gradient against float64 F.conv2d autograd on the SAME bf16-rounded operands, on grids that
are not multiples of the 16 x 32 / 8 x 32 tiles, with fewer than 8 output channels, and on a grid large enough that every persistent
workgroup walks several tiles."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PKG = "small-object-detection-transformers_amd"


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module(PKG + ".ops")


def _case(B, H, W, cout, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, 64, generator=g).bfloat16()
    w = (torch.randn(cout, 64, 3, 3, generator=g) * 0.05)
    b = torch.randn(cout, generator=g) * 0.1
    dy = torch.randn(B, H, W, cout, generator=g).bfloat16()
    return x, w, b, dy


def _layouts(w, b, dev):
    cout = w.shape[0]
    wq = w.bfloat16()
    wg = torch.zeros(8, 9 * 64, dtype=torch.bfloat16)
    wg[:cout] = wq.permute(0, 2, 3, 1).reshape(cout, 576)            # [n][tap * 64 + c]
    wT = torch.zeros(64, 9 * 8, dtype=torch.bfloat16)
    wT.view(64, 9, 8)[:, :, :cout] = wq.permute(1, 2, 3, 0).reshape(64, 9, cout)      # [c][tap * 8 + n]
    bp = torch.zeros(8)
    bp[:cout] = b
    return wq, wg.to(dev), wT.to(dev), bp.to(dev)


SHAPES = [(1, 16, 32, 4), (2, 20, 40, 4), (1, 50, 70, 3), (3, 8, 8, 8), (1, 5, 130, 4), (2, 96, 160, 4)]


@pytest.mark.parametrize("B,H,W,cout", SHAPES)
def test_forward_and_gradients_vs_float64_conv2d(ops, B, H, W, cout):
    dev = torch.device("cuda:0")
    x, w, b, dy = _case(B, H, W, cout, 7 * H + W)
    wq, wg, wT, bp = _layouts(w, b, dev)
    M = B * H * W
    xd = x.reshape(M, 64).to(dev)
    y = torch.full((M, 8), 7.0, device=dev, dtype=torch.bfloat16)
    ops.conv3_n8_fwd(xd, wg, bp, y, B, H, W)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = wq.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, b64, padding=1)
    got = y.float().cpu().view(B, H, W, 8)
    refr = ref.detach().permute(0, 2, 3, 1)
    scale = refr.abs().max().item()
    assert (got[..., :cout].double() - refr).abs().max().item() <= 6e-3 * scale
    assert (got[..., :cout].double() - refr).norm().item() <= 2.5e-3 * refr.norm().item()          # bf16 output rounding: ~1.1e-3 rms
    assert got[..., cout:].abs().max().item() == 0 if cout < 8 else True

    dyp = torch.zeros(B, H, W, 8, dtype=torch.bfloat16)
    dyp[..., :cout] = dy
    ref.backward(dy.double().permute(0, 3, 1, 2))
    dyd = dyp.reshape(M, 8).to(dev)
    dx = torch.full((M, 64), 3.0, device=dev, dtype=torch.bfloat16)
    ops.conv3_n8_dgrad(dyd, wT, dx, B, H, W)
    rdx = x64.grad.permute(0, 2, 3, 1)
    assert (dx.float().cpu().view(B, H, W, 64).double() - rdx).abs().max().item() <= 6e-3 * rdx.abs().max().item()
    assert (dx.float().cpu().view(B, H, W, 64).double() - rdx).norm().item() <= 2.5e-3 * rdx.norm().item()

    dw = torch.full((cout, 64, 3, 3), 0.5, device=dev)
    db = torch.full((cout,), -0.25, device=dev)
    scr = torch.empty(ops.conv3_n8_wgrad_scratch_floats(), device=dev)
    ops.conv3_n8_wgrad(dyd, xd, dw, db, scr, B, H, W, cout)
    assert ((dw.cpu().double() - 0.5) - w64.grad).abs().max().item() <= 2e-4 * w64.grad.abs().max().item() + 1e-4
    assert ((db.cpu().double() + 0.25) - b64.grad).abs().max().item() <= 2e-4 * b64.grad.abs().max().item() + 1e-4
    # no bias: NULL pointers
    y2 = torch.empty_like(y)
    ops.conv3_n8_fwd(xd, wg, None, y2, B, H, W)
    ref0 = (refr - b.double())
    assert (y2.float().cpu().view(B, H, W, 8)[..., :cout].double() - ref0).abs().max().item() <= 6e-3 * max(ref0.abs().max().item(), 1e-6)
    dw2 = torch.zeros_like(dw)
    ops.conv3_n8_wgrad(dyd, xd, dw2, None, scr, B, H, W, cout)
    assert torch.equal(dw2, dw - 0.5) or (dw2 - (dw - 0.5)).abs().max().item() <= 1e-5 * dw.abs().max().item()
    if cout <= 4:
        # the float32 (B, cout, H, W) boundary forms (deeplabedsr.py:73): output unrounded, gradient rounded to bf16 as it is read
        yn = torch.full((B, cout, H, W), 7.0, device=dev)
        ops.conv3_n8_fwd(xd, wg, bp, None, B, H, W, y_nchw=yn, cout=cout)
        assert (yn.cpu().double() - ref.detach()).abs().max().item() <= 2e-5 * scale + 1e-5
        dyn = dy.float().permute(0, 3, 1, 2).contiguous().to(dev)
        dx3 = torch.empty_like(dx)
        ops.conv3_n8_dgrad(None, wT, dx3, B, H, W, dy_nchw=dyn, cout=cout)
        assert torch.equal(dx3, dx)
        dw3, db3 = torch.zeros_like(dw), torch.zeros_like(db)
        ops.conv3_n8_wgrad(None, xd, dw3, db3, scr, B, H, W, cout, dy_nchw=dyn)
        assert torch.equal(dw3, dw2) and (db3 - (db + 0.25)).abs().max().item() <= 1e-5 * db.abs().max().item() + 1e-6


def test_many_tiles_per_workgroup_bit_identical_to_itself_and_close_to_the_gemm_path(ops):
    """1 x 512 x 1024: 1,024 forward tiles for 512 workgroups (every one walks two tiles: the register prefetch of the next tile, both
    LDS hand-overs); the K-segment GEMM the kernels replace is the second opinion; two runs of the weight gradient are bit-identical
    (fixed-order partial sums)."""
    dev = torch.device("cuda:0")
    B, H, W, cout = 1, 512, 1024, 4
    x, w, b, dy = _case(B, H, W, cout, 3)
    wq, wg, wT, bp = _layouts(w, b, dev)
    M = B * H * W
    xd = x.reshape(M, 64).to(dev)
    y = torch.empty(M, 8, device=dev, dtype=torch.bfloat16)
    ops.conv3_n8_fwd(xd, wg, bp, y, B, H, W)
    taps = [(dy_, dx_) for dy_ in (-1, 0, 1) for dx_ in (-1, 0, 1)]
    segs = [ops.SegSpec(xd, 64, 0, a, c, 1, 0, H, W) for (a, c) in taps]
    y_g = torch.empty_like(y)
    ops.gemm_nt(segs, wg, y_g, M, 8, 576, spatial=(H, W), bias=bp)
    d = (y.float() - y_g.float()).abs().max().item()
    assert d <= 2e-2 * y_g.float().abs().max().item()
    dyp = torch.zeros(M, 8, dtype=torch.bfloat16)
    dyp[:, :cout] = dy.reshape(M, cout)
    dyd = dyp.to(dev)
    dx = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
    ops.conv3_n8_dgrad(dyd, wT, dx, B, H, W)
    bsegs = [ops.SegSpec(dyd, 8, 0, -a, -c, 1, 0, H, W) for (a, c) in taps]
    dx_g = torch.empty_like(dx)
    ops.gemm_nt(bsegs, wT, dx_g, M, 64, 72, spatial=(H, W))
    assert (dx.float() - dx_g.float()).abs().max().item() <= 2e-2 * dx_g.float().abs().max().item()
    scr = torch.empty(ops.conv3_n8_wgrad_scratch_floats(), device=dev)
    dw1, dw2 = torch.zeros(cout, 64, 3, 3, device=dev), torch.zeros(cout, 64, 3, 3, device=dev)
    db1, db2 = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
    ops.conv3_n8_wgrad(dyd, xd, dw1, db1, scr, B, H, W, cout)
    ops.conv3_n8_wgrad(dyd, xd, dw2, db2, scr, B, H, W, cout)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)
    dw_g, db_g = torch.zeros(8, 576, device=dev), torch.zeros(8, device=dev)
    ops.gemm_tn(dyd, segs, dw_g, M, 8, 576, ldy=8, spatial=(H, W), dbias=db_g, kperm=(64, 9))
    assert (dw1 - dw_g[:cout].view(cout, 64, 3, 3)).abs().max().item() <= 1e-3 * dw_g.abs().max().item()
    assert (db1 - dyd[:, :cout].float().sum(0)).abs().max().item() <= 1e-3 * dyd.float().abs().sum(0).max().item()


C64_SHAPES = [(1, 16, 32), (2, 20, 40), (1, 50, 70), (1, 3, 5), (2, 96, 160)]


@pytest.mark.parametrize("B,H,W", C64_SHAPES)
def test_body_convolution_64_to_64_vs_float64_conv2d(ops, B, H, W):
    """sodt_conv3x3_c64_fwd (every epilogue the ResBlocks use, forward and mirrored-tap input gradient) and sodt_conv3x3_c64_wgrad
    (edsr.py:34-53) against float64 F.conv2d autograd on the same bf16 operands."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11 * H + W)
    M = B * H * W
    x = torch.randn(B, H, W, 64, generator=g).bfloat16()
    r = torch.randn(B, H, W, 64, generator=g).bfloat16()
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.04).bfloat16()
    b = torch.randn(64, generator=g) * 0.1
    dy = torch.randn(B, H, W, 64, generator=g).bfloat16()
    wg = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(dev)            # [n][tap * 64 + c]
    wT = w.permute(1, 2, 3, 0).reshape(64, 576).contiguous().to(dev)            # [c][tap * 64 + n]
    xd, rd, dyd, bd = x.reshape(M, 64).to(dev), r.reshape(M, 64).to(dev), dy.reshape(M, 64).to(dev), b.to(dev)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, b64, padding=1)
    refr = ref.detach().permute(0, 2, 3, 1)
    r64 = r.double()

    def check(got, want, what):
        got = got.float().cpu().view(B, H, W, 64).double()
        assert (got - want).norm().item() <= 2.5e-3 * want.norm().item(), what
        assert (got - want).abs().max().item() <= 8e-3 * want.abs().max().item(), what

    y = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
    ops.conv3_c64_fwd(xd, wg, y, B, H, W, bias=bd, relu=True)
    check(y, refr.clamp_min(0), "bias + ReLU")
    ops.conv3_c64_fwd(xd, wg, y, B, H, W, bias=bd, resid=rd)
    check(y, refr + r64, "bias + residual")
    ops.conv3_c64_fwd(xd, wg, y, B, H, W)
    check(y, refr - b.double(), "plain")

    ref.backward(dy.double().permute(0, 3, 1, 2))
    rdx = x64.grad.permute(0, 2, 3, 1)
    dx = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
    ops.conv3_c64_fwd(dyd, wT, dx, B, H, W, flip=True)
    check(dx, rdx, "input gradient")
    ops.conv3_c64_fwd(dyd, wT, dx, B, H, W, flip=True, drelu_aux=rd)
    check(dx, rdx * (r64 > 0), "input gradient x ReLU mask")
    ops.conv3_c64_fwd(dyd, wT, dx, B, H, W, flip=True, resid=rd)
    check(dx, rdx + r64, "input gradient + residual path")
    with pytest.raises(RuntimeError):        # one epilogue operand per launch (no layer of the branch has both)
        ops.conv3_c64_fwd(dyd, wT, dx, B, H, W, flip=True, drelu_aux=xd, resid=rd)

    dw = torch.full((64, 64, 3, 3), 0.5, device=dev)
    db = torch.full((64,), -0.25, device=dev)
    scr = torch.empty(ops.conv3_c64_wgrad_scratch_floats(), device=dev)
    ops.conv3_c64_wgrad(dyd, xd, dw, db, scr, B, H, W)
    assert ((dw.cpu().double() - 0.5) - w64.grad).abs().max().item() <= 2e-4 * w64.grad.abs().max().item() + 1e-4
    assert ((db.cpu().double() + 0.25) - b64.grad).abs().max().item() <= 2e-4 * b64.grad.abs().max().item() + 1e-4
    dw2 = torch.zeros_like(dw)
    ops.conv3_c64_wgrad(dyd, xd, dw2, None, scr, B, H, W)
    dw3 = torch.zeros_like(dw)
    ops.conv3_c64_wgrad(dyd, xd, dw3, None, scr, B, H, W)
    assert torch.equal(dw2, dw3)


def test_body_convolution_many_tiles_vs_gemm_path(ops):
    """1 x 512 x 1024 (1,024 / 2,048 tiles: every persistent workgroup walks several) against the K-segment GEMMs the kernels replace."""
    dev = torch.device("cuda:0")
    B, H, W = 1, 512, 1024
    M = B * H * W
    g = torch.Generator().manual_seed(5)
    xd = torch.randn(M, 64, generator=g).bfloat16().to(dev)
    dyd = torch.randn(M, 64, generator=g).bfloat16().to(dev)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.04).bfloat16()
    wg = w.permute(0, 2, 3, 1).reshape(64, 576).contiguous().to(dev)
    bd = (torch.randn(64, generator=g) * 0.1).to(dev)
    taps = [(a, c) for a in (-1, 0, 1) for c in (-1, 0, 1)]
    segs = [ops.SegSpec(xd, 64, 0, a, c, 1, 0, H, W) for (a, c) in taps]
    y, y_g = torch.empty(M, 64, device=dev, dtype=torch.bfloat16), torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
    ops.conv3_c64_fwd(xd, wg, y, B, H, W, bias=bd, relu=True)
    ops.gemm_nt(segs, wg, y_g, M, 64, 576, spatial=(H, W), bias=bd, relu=True)
    assert (y.float() - y_g.float()).norm().item() <= 3e-3 * y_g.float().norm().item()
    scr = torch.empty(ops.conv3_c64_wgrad_scratch_floats(), device=dev)
    dw, db = torch.zeros(64, 64, 3, 3, device=dev), torch.zeros(64, device=dev)
    ops.conv3_c64_wgrad(dyd, xd, dw, db, scr, B, H, W)
    dw_g, db_g = torch.zeros(64, 576, device=dev), torch.zeros(64, device=dev)
    ops.gemm_tn(dyd, segs, dw_g, M, 64, 576, spatial=(H, W), dbias=db_g, kperm=(64, 9))
    assert (dw - dw_g.view(64, 64, 3, 3)).abs().max().item() <= 1e-3 * dw_g.abs().max().item()
    assert (db - db_g).abs().max().item() <= 1e-3 * db_g.abs().max().item()


@pytest.mark.parametrize("B,H,W", [(2, 16, 32), (1, 12, 40), (1, 40, 72)])
def test_upsampler_stage_conv_256_plus_pixel_shuffle_as_four_plane_launches(ops, B, H, W):
    """edsr.py:14-24: conv(64 -> 256, 3) + nn.PixelShuffle(2).  Plane p = 2 i + j (output channels 4 c + p) is a 64 -> 64 launch that stores
    pixel (y, x) at (2 y + i, 2 x + j); the gradients read the fine gradient through the same map (sodt_conv3_geo).  Against float64
    F.conv2d + F.pixel_shuffle autograd on the same bf16 operands."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H + 3 * W)
    M = B * H * W
    x = torch.randn(B, H, W, 64, generator=g).bfloat16()
    w = (torch.randn(256, 64, 3, 3, generator=g) * 0.04).bfloat16()
    b = torch.randn(256, generator=g) * 0.1
    dfine = torch.randn(B, 2 * H, 2 * W, 64, generator=g).bfloat16()
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    ref = F.pixel_shuffle(F.conv2d(x64, w64, b64, padding=1), 2)                 # (B, 64, 2H, 2W)
    ref.backward(dfine.double().permute(0, 3, 1, 2))
    refr = ref.detach().permute(0, 2, 3, 1)
    xd, bd = x.reshape(M, 64).to(dev), b.to(dev)
    wg = w.permute(0, 2, 3, 1).reshape(256, 576).contiguous().to(dev)            # [n][tap * 64 + c]
    wTp = w.reshape(64, 4 * 576).t().contiguous().view(4, 64, 576).to(dev)      # [p][k][tap * 64 + c] = W[4 c + p][k][tap]
    y = torch.full((4 * M, 64), 9.0, device=dev, dtype=torch.bfloat16)
    for p in range(4):
        ops.conv3_c64_fwd(xd, wg, y, B, H, W, bias=bd, geo=ops.conv3_geo(w_row=(4, p), out=(2, p >> 1, p & 1)))
    got = y.float().cpu().view(B, 2 * H, 2 * W, 64).double()
    assert (got - refr).norm().item() <= 2.5e-3 * refr.norm().item()
    assert (got - refr).abs().max().item() <= 8e-3 * refr.abs().max().item()

    dd = dfine.reshape(4 * M, 64).to(dev)
    dx = torch.full((M, 64), 5.0, device=dev, dtype=torch.bfloat16)
    dw, db = torch.full((256, 64, 3, 3), 0.5, device=dev), torch.full((256,), -0.25, device=dev)
    scr = torch.empty(ops.conv3_c64_wgrad_scratch_floats(), device=dev)
    for p in range(4):
        ops.conv3_c64_wgrad(dd, xd, dw, db, scr, B, H, W, geo=ops.conv3_geo(w_row=(4, p), out=(2, p >> 1, p & 1)))
        ops.conv3_c64_fwd(dd, wTp[p], dx, B, H, W, flip=True, resid=dx if p else None, geo=ops.conv3_geo(inp=(2, p >> 1, p & 1)))
    rdx = x64.grad.permute(0, 2, 3, 1)
    gdx = dx.float().cpu().view(B, H, W, 64).double()
    # (three bf16 roundings of the running sum on top of the output rounding)
    assert (gdx - rdx).norm().item() <= 5e-3 * rdx.norm().item()
    assert ((dw.cpu().double() - 0.5) - w64.grad).abs().max().item() <= 2e-4 * w64.grad.abs().max().item() + 1e-4
    assert ((db.cpu().double() + 0.25) - b64.grad).abs().max().item() <= 2e-4 * b64.grad.abs().max().item() + 1e-4


@pytest.mark.parametrize("M,C,dt", [(1000, 64, torch.bfloat16), (70000, 128, torch.bfloat16), (333, 32, torch.float32)])
def test_col_stats_matches_float64_sums(ops, M, C, dt):
    """sodt_col_stats: the BatchNorm batch statistics (common.py:44-49 in training mode) of a stored convolution output."""
    dev = torch.device("cuda:0")
    L = importlib.import_module(PKG + "._lib")
    z = (torch.randn(M, C, generator=torch.Generator().manual_seed(M)) * 2 + 0.5).to(dt).to(dev)
    stats = torch.zeros(L.STATS_REPL, 2, C, device=dev, dtype=torch.float64)
    ops.col_stats(z, stats, M, C)
    got = stats.sum(0).cpu()
    zd = z.double().cpu()
    assert (got[0] - zd.sum(0)).abs().max().item() <= 1e-5 * zd.abs().sum(0).max().item()
    assert (got[1] - (zd * zd).sum(0)).abs().max().item() <= 1e-5 * (zd * zd).sum(0).max().item()


def test_head_3x3_on_the_direct_kernels_matches_the_gemm_path():
    """Engine.use_direct_conv3: the stride-4 C3's Bottleneck 3x3 (64 -> 64, common.py:38-50, :98-115) forward / backward through
    sodt_conv3x3_c64_* + sodt_col_stats against the nine-segment GEMM with the statistics epilogue: same model, same inputs, bf16 -
    logits and every head gradient agree within bf16 noise (each path is pinned to the oracle by tests/test_model_gpu.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=g).to(dev)
    ir = torch.rand(2, 3, 256, 256, generator=g).to(dev)
    outs = []
    for direct in (True, False):
        m = bench.build_model(256, dev, torch.bfloat16)            # (seeded: both models start from the same parameters)
        eng = m._get_engine()
        eng.use_direct_conv3 = direct
        pred, _ = m(x, ir, "RGB+IR")
        (pred[0].float() * torch.linspace(-1, 1, pred[0].numel(), device=dev).view_as(pred[0])).sum().backward()
        used = [t for t, sv in next(iter(eng.plans.values())).saved.items() if isinstance(sv, dict) and sv.get("direct")]
        assert bool(used) == direct, used
        grads = {n: p.grad.detach().float().clone() for n, p in m.named_parameters() if p.grad is not None}
        outs.append((pred[0].detach().float().clone(), grads))
    (pa, ga), (pb, gb) = outs
    assert (pa - pb).norm().item() <= 2e-2 * pb.norm().item()
    rel = sorted(((ga[n] - gb[n]).norm().item() / (gb[n].norm().item() + 1e-12), n) for n in gb if gb[n].norm().item() > 0)
    head = [r for r in rel if r[1].startswith("detect.")]
    # the head (the convolution's own block and what is downstream of it in the backward): tight; the encoder sees the different bf16
    # rounding of one layer amplified through 40 blocks - noise-level statistics (median), as between any two bf16 paths
    assert len(head) > 30 and head[-1][0] <= 5e-2, head[-3:]
    assert rel[len(rel) // 2][0] <= 8e-2, rel[len(rel) // 2]


def test_rejects_float32_and_misaligned(ops):
    dev = torch.device("cuda:0")
    x = torch.zeros(64, 64, device=dev)
    w = torch.zeros(8, 576, device=dev)
    y = torch.zeros(64, 8, device=dev)
    with pytest.raises(RuntimeError):
        ops.conv3_n8_fwd(x, w, None, y, 1, 8, 8)
    xb = torch.zeros(64 * 64 + 8, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.conv3_n8_fwd(xb[4:4 + 64 * 64].view(64, 64), w.bfloat16(), None, y.bfloat16(), 1, 8, 8)
