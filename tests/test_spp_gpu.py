"""SPP (basics/models/common.py:129-140): the cascaded 5x5 max-pool kernel (csrc/pool.hip) against torch's MaxPool2d 5 / 9 /
13 incl. the gradient routing of ties, and the whole operator (spp.SPPOp: cv1 -> pools -> K-segment concat -> cv2, forward +
hand-written backward) against the outputs of the reference's own common.SPP in tests/golden/spp.pt."""
import importlib
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PKG = "small-object-detection-transformers_amd"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tok(x):          # NCHW -> token-major [B*H*W][C]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous()


def _nchw(t, B, H, W):
    return t.view(B, H, W, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_maxpool5_cascade_matches_torch_5_9_13(ops, dev, dt):
    B, C, H, W = 2, 32, 13, 9
    g = torch.Generator().manual_seed(3)
    x = ((torch.randn(B, C, H, W, generator=g) * 2).round() / 2)          # coarse grid: many ties inside a window
    xt = _tok(x).to(dev).to(dt)
    cur, args, outs = xt, [], []
    for _ in range(3):
        y = torch.empty_like(xt)
        a = torch.zeros(B * H * W, C, device=dev, dtype=torch.uint8)
        ops.maxpool5_fwd(cur, y, a, B, H, W, C)
        outs.append(y); args.append(a); cur = y
    xr = x.clone().requires_grad_(True)
    refs = [F.max_pool2d(xr, k, 1, k // 2) for k in (5, 9, 13)]
    for y, r in zip(outs, refs):
        assert torch.equal(_nchw(y.float().cpu(), B, H, W), r.detach()), "cascade of 5x5 pools != MaxPool2d(5 | 9 | 13)"
    # backward of ONE pool: ties route to the first maximum in scan order, as torch's max_pool2d backward does
    dy = torch.randn(B, C, H, W, generator=g).round()
    refs[0].backward(dy)
    dx = torch.zeros_like(xt)
    ops.maxpool5_bwd(_tok(dy).to(dev).to(dt), args[0], dx, B, H, W, C)
    torch.cuda.synchronize()
    assert float((_nchw(dx.float().cpu(), B, H, W) - xr.grad).abs().max()) <= (1e-6 if dt == torch.float32 else 6e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_spp_operator_vs_reference_golden(dev, dt):
    S = importlib.import_module(PKG + ".spp")
    g = torch.load(os.path.join(GOLD, "spp.pt"))
    B, c1, H, W = g["x"].shape
    c2 = g["y"].shape[1]
    p = {k: v.clone().to(dev) for k, v in g["sd"].items()}
    op = S.SPPOp(c1, c2, B, H, W, dt, dev)
    out = op.forward(p, _tok(g["x"]).to(dev).to(dt), training=True)
    grads = {k: torch.zeros_like(v) for k, v in p.items() if "running" not in k and "num_batches" not in k}
    dx = op.backward(p, _tok(g["gsel"]).to(dev).to(dt), grads)
    torch.cuda.synchronize()
    tol = 2e-4 if dt == torch.float32 else 4e-2

    def chk(a, b, what, grad=False):
        a, b = a.double().cpu(), b.double()
        if grad and dt == torch.bfloat16:
            # max-pooling is not continuous: where two window entries differ by less than a bf16 ulp the gradient takes another
            # route than in f32, a large POINTWISE change - compare in the L2 sense
            e, s = float((a - b).norm()), float(b.norm())
            assert e <= 0.25 * s, f"{what}: relative L2 error {e / s:.3f} (bf16)"
            return
        s = float(b.abs().max())
        e = float((a - b).abs().max())
        assert e <= tol * max(s, 1e-3), f"{what}: {e:.3e} vs scale {s:.3e} ({dt})"
    chk(_nchw(out.float(), B, H, W), g["y"], "SPP output")
    chk(_nchw(dx.float(), B, H, W), g["dx"], "SPP input gradient", grad=True)
    for k, v in g["grads"].items():
        chk(grads[k].view(v.shape), v, f"gradient of {k}", grad=True)
    for k, v in g["stats_after"].items():
        chk(p[k], v, k)


def test_spp_row_parses_with_reference_parameter_names():
    M = importlib.import_module(PKG + ".model")
    m = M.SPP(512, 512)
    keys = set(dict(m.named_parameters()).keys()) | set(dict(m.named_buffers()).keys())
    assert {"cv1.conv.weight", "cv1.bn.weight", "cv2.conv.weight", "cv2.bn.running_var"} <= keys
    assert m.cv1.conv.weight.shape == (256, 512, 1, 1) and m.cv2.conv.weight.shape == (512, 1024, 1, 1)
