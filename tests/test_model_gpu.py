"""Whole-model parity on the GPU: Model(cfg) on the HIP engine vs the CPU oracle (same procedural
weights, same synthetic inputs) and vs the golden vectors captured from the real reference."""
import importlib
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def build(dev, img, nc=8):
    from oracle import ref_torch as R
    M = importlib.import_module("small-object-detection-transformers_amd.model")
    cfg = dict(nc=nc, depth_multiple=0.33, width_multiple=0.5, anchors=[[10, 13, 16, 30, 33, 23]],
               backbone=[[-1, 1, "ImageEncoderViT", [img, 6, 192, 4, 256, 4]]],
               head=[[2, 1, "Conv", [512, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 1], 1, "Concat", [1]],
                     [-1, 3, "C3", [512, False]], [-1, 1, "Conv", [256, 1, 1]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
                     [[-1, 0], 1, "Concat", [1]], [-1, 3, "C3", [256, False]], [[10], 1, "Detect", ["nc", "anchors"]]])
    model = M.Model(cfg, input_mode="RGB+IR", ch_steam=3, ch=128, nc=nc)
    sd = R.procedural_state_dict(img, nc)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("relative_position_index" in k or "attn_mask" in k) for k in missing), missing
    return model.to(dev), sd


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max()), float(b.abs().max())


# bf16 bounds: 1.5x / 2x what this test measures (logits 0.197 on |logit| <= 3.25, worst gradient 0.085); the reference's own
# bf16 autocast run differs from its f32 run by 0.29 / 0.25 (tests/golden/autocast_512.pt, tests/test_bf16_parity_gpu.py)
# nc = 12: na * no = 51 > 48, the Detect GEMM pads to 64 columns forward AND backward (advisor r3: the backward had 48 hard-coded)
@pytest.mark.parametrize("dtype,tol_logit,tol_grad,nc", [(torch.float32, 1e-3, 2e-3, 8), (torch.bfloat16, 0.30, 0.17, 8),
                                                         (torch.float32, 1e-3, 2e-3, 12), (torch.bfloat16, 0.30, 0.17, 12)])
def test_train_step_vs_oracle(dev, dtype, tol_logit, tol_grad, nc):
    from oracle import ref_torch as R
    S, B = 128, 2
    model, sd = build(dev, S, nc)
    model.compute_dtype = dtype
    model.train()
    x_rgb, x_ir = R.synthetic_inputs(B, S, seed=1)
    pred, y = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    gsel = R._hash01("gsel", pred[0].numel()).view(pred[0].shape).float()
    (pred[0] * gsel.to(dev)).sum().backward()

    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
    ns = {}
    opred, oy = R.model_forward(osd, x_rgb, x_ir, True, ns)
    (opred[0] * gsel).sum().backward()
    err, scale = rel(pred[0], opred[0])
    assert err <= tol_logit, f"logits max abs err {err:.3e} (|logit| max {scale:.2f})"
    for i in range(3):
        e, s = rel(y[i], oy[i])
        assert e <= tol_logit * max(1.0, s), f"encoder feature {i}: {e:.3e} / {s:.2f}"
    e, s = rel(y[10], oy[10])
    assert e <= tol_logit * max(1.0, s), f"head feature 10: {e:.3e}"
    # gradients of every parameter
    worst = ("", 0.0)
    # stage3.0.mlp.fc2.bias has a mathematically zero gradient (constant in front of a bias-free conv + batch-stat
    # BN), so errors are measured against |g| plus a floor tied to the typical gradient magnitude
    gmed = sorted(float(osd[n].grad.double().norm()) for n, _ in model.named_parameters())[len(osd) // 4]
    allr = []
    for n, p in model.named_parameters():
        assert p.grad is not None, n
        og = osd[n].grad
        gn = float(og.double().norm())
        d = float((p.grad.double().cpu() - og.double()).norm())
        if n == "image_encoder.stage3.0.mlp.fc2.bias":      # exact zero in exact arithmetic: absolute check only
            assert d <= max(tol_grad, 0.05) * gmed, (n, d, gmed)
            continue
        r = d / (gn + 1e-2 * gmed + 1e-12)
        allr.append((r, n))
        if r > worst[1]:
            worst = (n, r)
    allr.sort(reverse=True)
    assert worst[1] <= tol_grad, f"worst relative gradient errors {allr[:6]}"
    # BatchNorm running statistics
    for k, v in ns.items():
        e, s = rel(dict(model.named_buffers())[k], v)
        assert e <= (1e-4 if dtype == torch.float32 else 3e-2) * max(1.0, s), k


def test_golden_512_logits_f32(dev):
    """1e-3 on logits against vectors captured from the reference itself (oracle/gen_golden.py)."""
    from oracle import ref_torch as R
    g = torch.load(os.path.join(GOLD, "full_model_512.pt"))
    model, _ = build(dev, 512)
    model.compute_dtype = torch.float32
    model.train()
    x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=0)
    pred, y = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    e, s = rel(pred[0][:, :, ::8, ::8, :], g["logits_sub"])
    assert e <= 1e-3, f"logits vs reference golden: {e:.3e} (scale {s:.2f})"
    for i in range(3):
        e, s = rel(y[i][..., ::8, ::8], g["feats_sub"][i])
        assert e <= 1e-3 * max(1.0, s), (i, e)
    loss = pred[0].float().square().mean()
    assert abs(float(loss) - g["loss"]) <= 1e-4 * abs(g["loss"])
    loss.backward()
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        gn = g["gnorm"][n]
        r = abs(float(p.grad.double().norm()) - gn) / (gn + 1e-7)
        if r > worst[1] and gn > 1e-6:
            worst = (n, r)
    assert worst[1] <= 5e-3, worst
    # sub-sampled gradient VALUES of the reference (norms alone would pass a permuted / sign-flipped gradient)
    worst = ("", 0.0)
    for n, p in model.named_parameters():
        ref = g["gsub"][n].double()
        got = p.grad.detach().reshape(-1)[::max(1, p.numel() // 64)][:64].double().cpu()
        r = float((got - ref).abs().max()) / (g["gnorm"][n] / max(1.0, p.numel()) ** 0.5 + 1e-9)     # in units of the rms gradient
        if r > worst[1] and g["gnorm"][n] > 1e-6:
            worst = (n, r)
    assert worst[1] <= 2e-2, f"gradient values vs reference golden: {worst}"
    # eval: decode + running stats from the golden
    model.eval()
    with torch.no_grad():
        z, praw, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    e, s = rel(z[:, ::257, :], g["z_sub"])
    assert e <= 2e-3 * max(1.0, s), f"eval decode {e:.3e} / {s:.1f}"


def test_replay_and_accumulation(dev):
    """second step replays the recorded plan; gradients accumulate across backward calls until reset."""
    from oracle import ref_torch as R
    model, _ = build(dev, 128)
    model.compute_dtype = torch.float32
    model.train()
    x_rgb, x_ir = R.synthetic_inputs(2, 128, seed=4)
    xr, xi = x_rgb.to(dev), x_ir.to(dev)
    p1, _ = model(xr, xi, "RGB+IR")
    p1[0].square().mean().backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    rm = {k: v.clone() for k, v in model.named_buffers() if "running_mean" in k}
    p2, _ = model(xr, xi, "RGB+IR")
    assert torch.allclose(p1[0], p2[0], atol=1e-5)
    p2[0].square().mean().backward()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=2e-3, atol=1e-6), n
    for k, v in model.named_buffers():
        if "running_mean" in k:
            assert not torch.equal(v, rm[k])
    for p in model.parameters():
        p.grad = None
    p3, _ = model(xr, xi, "RGB+IR")
    p3[0].square().mean().backward()
    for n, p in model.named_parameters():
        assert torch.allclose(p.grad, g1[n], rtol=2e-3, atol=1e-6), n


def test_general_cross_channel_window(dev):
    """CAttentionBlock with window 2 + shift 1 (the form the reference code expresses but never enables)."""
    from oracle import ref_torch as R
    S, B = 128, 1
    model, sd = build(dev, S)
    model.compute_dtype = torch.float32
    model.image_encoder.chan_block.window_size = 2
    model.image_encoder.chan_block.shift_size = 1
    model.train()
    x_rgb, x_ir = R.synthetic_inputs(B, S, seed=5)
    pred, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    pred[0].square().mean().backward()
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
    opred, _ = R.model_forward(osd, x_rgb, x_ir, True, {}, None, 2, 1)
    opred[0].square().mean().backward()
    e, s = rel(pred[0], opred[0])
    assert e <= 1e-3, e
    for n in ("image_encoder.channel_embed_g.proj.weight", "image_encoder.chan_block.norm3.weight", "image_encoder.channel_embed_i.proj.bias"):
        g, og = dict(model.named_parameters())[n].grad.cpu(), osd[n].grad
        assert float((g - og).norm() / og.norm()) < 5e-3, n


def test_eval_with_nms_stage(dev):
    """Model.nms() (model.py:327-339): eval forward -> decode -> non_max_suppression, all on the GPU; the detections
    equal the oracle's NMS of the same decoded rows (identical candidate ids)."""
    from oracle import ref_torch as R
    model, _ = build(dev, 128)
    model.compute_dtype = torch.float32
    model.eval()
    x_rgb, x_ir = R.synthetic_inputs(2, 128, seed=5)
    with torch.no_grad():
        z, praw, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
        model.nms()
        model._nms.conf = float(z[..., 4].median())        # random-init objectness is ~3e-4: keep about half the rows
        dets, praw2, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    ref = R.non_max_suppression(z.cpu(), model._nms.conf, model._nms.iou)
    assert len(dets) == 2
    for d, r in zip(dets, ref):
        assert d.shape == r.shape
        if r.numel():
            assert torch.equal(d[:, 4:].cpu(), r[:, 4:])
            assert float((d[:, :4].cpu() - r[:, :4]).abs().max()) < 1e-3
    model.nms(False)
    with torch.no_grad():
        z2, _, _ = model(x_rgb.to(dev), x_ir.to(dev), "RGB+IR")
    assert torch.equal(z, z2)
