"""csrc/convmlp.hip: the 2x2-conv MLP (backbone_vit.py:892-905) with fc1 folded into the convolution.  The four small kernels against
f64 statements of what they compute, and the folded forward / backward against torch autograd on the REFERENCE'S form
(fc1 -> F.pad(0, 1, 0, 1) -> conv1 -> GELU -> fc2) - the algebra the fold relies on."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rnd(shape, dev, seed, scale=1.0, dt=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dt)


def params(dev, Cc, seed=0):
    W1 = rnd((Cc, Cc), dev, seed + 1, 1 / math.sqrt(Cc))
    b1 = rnd((Cc,), dev, seed + 2, 0.5)
    Wc = rnd((Cc, Cc, 2, 2), dev, seed + 3, 1 / math.sqrt(4 * Cc))
    bc = rnd((Cc,), dev, seed + 4, 0.5)
    return W1, b1, Wc, bc


@pytest.mark.parametrize("Cc", [192, 64, 384])
def test_compose_and_border_terms(ops, dev, Cc):
    W1, b1, Wc, bc = params(dev, Cc)
    weff = torch.zeros(Cc, 4 * Cc, device=dev, dtype=torch.bfloat16)
    weffT = torch.zeros_like(weff)
    beff = torch.zeros(Cc, device=dev)
    vtap = torch.zeros(4, Cc, device=dev)
    ops.convmlp_compose(W1, b1, Wc, bc, weff, weffT, beff, vtap, Cc)
    wc = Wc.double().reshape(Cc, Cc, 4)                                   # [co][m][tap]
    ref = torch.einsum("omt,mi->oti", wc, W1.double())                    # [co][tap][ci]
    assert float((weff.double().view(Cc, 4, Cc) - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
    assert torch.equal(weffT.view(Cc, 4, Cc), weff.view(Cc, 4, Cc).permute(2, 1, 0).contiguous())
    vref = torch.einsum("omt,m->to", wc, b1.double())
    assert float((vtap.double() - vref).abs().max()) <= 1e-5 * float(vref.abs().max())
    assert float((beff.double() - (bc.double() + vref.sum(0))).abs().max()) <= 1e-5 * float(vref.abs().max() + bc.abs().max())


@pytest.mark.parametrize("B,H,W,Cc", [(2, 16, 16, 192), (1, 8, 24, 192), (3, 32, 8, 192), (2, 16, 16, 384)])
def test_folded_forward_and_backward_match_the_reference_form(ops, dev, B, H, W, Cc):
    """x -> fc1 -> pad -> conv1 -> GELU -> [a linear read-out] on the CPU in f64 with autograd, against: composed weights + tap GEMM +
    border fix (forward), and tap GEMM with the transposed composed weights + weight-gradient GEMM + border sums + decompose (backward)"""
    M, dt = B * H * W, torch.bfloat16
    W1, b1, Wc, bc = params(dev, Cc, seed=10)
    x = rnd((M, Cc), dev, 20, dt=dt)
    dy = rnd((M, Cc), dev, 21, dt=dt)                     # gradient arriving at the conv's pre-activation
    # ---- reference form (f64, the bf16 values of x / dy)
    W1d, b1d, Wcd, bcd = (t.double().cpu().requires_grad_(True) for t in (W1, b1, Wc, bc))
    xd = x.double().cpu().requires_grad_(True)
    u = xd @ W1d.t() + b1d
    ui = u.view(B, H, W, Cc).permute(0, 3, 1, 2)
    c = F.conv2d(F.pad(ui, (0, 1, 0, 1)), Wcd, bcd).permute(0, 2, 3, 1).reshape(M, Cc)
    (c * dy.double().cpu()).sum().backward()
    # ---- folded forward
    weff = torch.zeros(Cc, 4 * Cc, device=dev, dtype=dt)
    weffT = torch.zeros_like(weff)
    beff = torch.zeros(Cc, device=dev)
    vtap = torch.zeros(4, Cc, device=dev)
    ops.convmlp_compose(W1, b1, Wc, bc, weff, weffT, beff, vtap, Cc)
    taps = ((0, 0), (0, 1), (1, 0), (1, 1))
    cp = torch.zeros(M, Cc, device=dev, dtype=dt)
    ca = torch.zeros(M, Cc, device=dev, dtype=dt)
    segs = [ops.SegSpec(x, Cc, 0, ty, tx, 1, 0, H, W) for (ty, tx) in taps]
    ops.gemm_nt(segs, weff, cp, M, Cc, 4 * Cc, spatial=(H, W), bias=beff, gelu_out=ca)
    ops.convmlp_border_fix(cp, ca, vtap, B, H, W, Cc)
    torch.cuda.synchronize()
    cref = c.detach()
    assert float((cp.double().cpu() - cref).abs().max()) <= 3e-2 * float(cref.abs().max()), "pre-activation"
    # border tokens specifically (a missing / doubled correction would be a bias-sized error there only)
    idx = torch.arange(M).view(B, H, W)
    border = torch.cat([idx[:, -1, :].reshape(-1), idx[:, :, -1].reshape(-1)])
    eb = float((cp.double().cpu()[border] - cref[border]).abs().max())
    assert eb <= 3e-2 * float(cref.abs().max()), f"border tokens: {eb:.3e}"
    assert float((ca.double().cpu() - F.gelu(cref)).abs().max()) <= 3e-2 * float(F.gelu(cref).abs().max())
    # ---- folded backward
    dxn = torch.zeros(M, Cc, device=dev, dtype=dt)
    segs_n = [ops.SegSpec(dy, Cc, 0, -ty, -tx, 1, 0, H, W) for (ty, tx) in taps]
    ops.gemm_nt(segs_n, weffT, dxn, M, Cc, 4 * Cc, spatial=(H, W))
    scr = torch.zeros(Cc * 4 * Cc + 4 * Cc, device=dev)
    dweff, colsum, bs = scr[: Cc * 4 * Cc].view(Cc, 4 * Cc), scr[Cc * 4 * Cc: Cc * 4 * Cc + Cc], scr[Cc * 4 * Cc + Cc:].view(3, Cc)
    ops.gemm_tn(dy, segs, dweff, M, Cc, 4 * Cc, spatial=(H, W), dbias=colsum)
    ops.convmlp_border_sums(dy, bs, B, H, W, Cc)
    gWc, gbc, gW1, gb1 = (torch.ones_like(t) for t in (Wc, bc, W1, b1))       # accumulate INTO existing values
    ops.convmlp_decompose(dweff, colsum, bs, W1, b1, Wc, gWc, gbc, gW1, gb1, Cc)
    torch.cuda.synchronize()
    def close(a, ref, what, tol=2e-2):
        a, ref = a.double().cpu(), ref.double()
        e = float((a - ref).abs().max())
        assert e <= tol * float(ref.abs().max()) + 1e-9, f"{what}: {e:.3e} vs {float(ref.abs().max()):.3e}"
    close(dxn, xd.grad, "d x")
    close(gWc - 1, Wcd.grad, "d conv1.weight")
    close(gbc - 1, bcd.grad, "d conv1.bias")
    close(gW1 - 1, W1d.grad, "d fc1.weight")
    close(gb1 - 1, b1d.grad, "d fc1.bias")


def test_engine_fold_is_as_close_to_the_oracle_as_the_three_gemm_form(dev):
    """whole model, bf16, B = 2 @ 128^2, against the f32 CPU oracle: the shipped folded path (fc1 composed into conv1) must be as
    close as the three-GEMM form it replaces (convmlp_fold_maxc = 0: fc1 GEMM, conv GEMM, du / dxn / dW1 GEMMs).  Two bf16 paths
    differ from EACH OTHER by the bf16 noise of the model (0.19 on |logit| <= 3.9), so the gate is on their distances to the truth."""
    from oracle import ref_torch as R
    from test_model_gpu import build
    S, Bn = 128, 2
    x_rgb, x_ir = R.synthetic_inputs(Bn, S, seed=3)
    errs = []
    osd = None
    for maxc in (0, 384):
        model, sd = build(dev, S)
        model.compute_dtype = torch.bfloat16
        model.train()
        eng = model._get_engine()
        eng.convmlp_fold_maxc = maxc                  # read when the parameter layouts are first prepared (first forward)
        pred, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
        assert (len(eng._prep_for(torch.bfloat16)["cmlp"]) > 0) == (maxc > 0)
        gsel = R._hash01("gsel", pred[0].numel()).view(pred[0].shape).float()
        (pred[0] * gsel.to(dev)).sum().backward()
        if osd is None:
            osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
            opred, _ = R.model_forward(osd, x_rgb, x_ir, True, {})
            (opred[0] * gsel).sum().backward()
        e_logit = float((pred[0].detach().float().cpu() - opred[0].detach()).abs().max())
        rel = []
        for n, p_ in model.named_parameters():
            og = osd[n].grad
            if og is None or n == "image_encoder.stage3.0.mlp.fc2.bias":
                continue
            rel.append(float((p_.grad.float().cpu() - og).norm()) / (float(og.norm()) + 1e-12))
        rel.sort()
        errs.append((e_logit, rel[len(rel) // 2], rel[-1]))
    (l0, m0, w0), (l1, m1, w1) = errs
    assert l1 <= 1.3 * l0 + 0.02, f"logits: folded {l1:.3f} vs three-GEMM {l0:.3f}"
    assert m1 <= 1.3 * m0 + 0.005 and w1 <= 1.3 * w0 + 0.02, f"gradients (median, worst): folded {m1:.4f} {w1:.4f} vs three-GEMM {m0:.4f} {w0:.4f}"
