"""The CPU oracle against the golden vectors captured from the REAL reference (oracle/gen_golden.py)."""
import os

import pytest
import torch

from oracle import ref_torch as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def md(a, b):
    return float((a.double() - b.double()).abs().max())


@pytest.fixture(scope="module")
def per_module():
    return torch.load(os.path.join(GOLD, "per_module.pt"))


@pytest.mark.parametrize("tag", ["swin_lin", "swin_conv", "swin_clamp"])
def test_swin_block(per_module, tag):
    g = per_module[tag]
    c = g["cfg"]
    sd = {"b." + k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    y = R.swin_block(sd, "b.", x, c["H"], c["W"], c["window_size"], c["shift_size"], c["linear_mlp"])
    assert md(y, g["y"]) < 1e-5
    (y * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
    assert md(x.grad, g["dx"]) < 1e-5
    for k, v in g["grads"].items():
        assert md(sd["b." + k].grad, v) < 1e-4, k


@pytest.mark.parametrize("tag", ["pad_lin", "pad_conv_shift"])
def test_swin_block_with_window_padding(tag):
    """Resolutions that are not a multiple of the window: the reference zero-pads AFTER norm1 and crops after the attention
    (backbone_vit.py:619-672); shifted blocks roll the unpadded grid first and mask with the region ids of the unpadded grid.
    tests/golden/pad_block.pt comes from the reference's own SwinTransformerBlock (oracle/gen_golden.py --only-pad)."""
    g = torch.load(os.path.join(GOLD, "pad_block.pt"))[tag]
    c = g["cfg"]
    sd = {"b." + k: v.clone().requires_grad_(True) for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    y = R.swin_block(sd, "b.", x, c["H"], c["W"], c["window_size"], c["shift_size"], c["linear_mlp"])
    assert md(y, g["y"]) < 1e-5
    (y * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
    assert md(x.grad, g["dx"]) < 1e-5
    for k, v in g["grads"].items():
        assert md(sd["b." + k].grad, v) < 1e-4, k


def test_patch_merging_and_embed(per_module):
    g = per_module["pmerge"]
    y = R.patch_merging({"p." + k: v for k, v in g["sd"].items()}, "p.", g["x"], g["H"], g["W"])
    assert md(y, g["y"]) < 1e-5
    for tag, pad in (("pe_pad1", 1), ("pe_pad0", 0)):
        e = per_module[tag]
        y = torch.nn.functional.conv2d(e["x"], e["sd"]["proj.weight"], e["sd"]["proj.bias"], stride=4, padding=pad).permute(0, 2, 3, 1)
        assert md(y, e["y"]) < 1e-6


@pytest.mark.parametrize("tag", ["ca_w1", "ca_w2", "ca_w2s1"])
def test_cross_channel_attention(per_module, tag):
    g = per_module[tag]
    outs = R.cattention_block({"c." + k: v for k, v in g["sd"].items()}, *g["ins"], window_size=g["ws"], shift=g["shift"], pfx="c.")
    for a, b in zip(outs, g["outs"]):
        assert md(a, b) < 1e-5
    if g["ws"] == 1:   # the shipped degenerate form: x_q = LN(q + kv)   
        r, gg, b, i = g["ins"]
        want = R.layer_norm(r + gg, g["sd"]["norm1.weight"], g["sd"]["norm1.bias"])
        assert md(outs[0], want) < 1e-6


@pytest.mark.parametrize("tag", ["conv1", "conv3", "c3"])
def test_head_blocks(per_module, tag):
    g = per_module[tag]
    sd = {"m." + k: v.clone() for k, v in g["sd"].items()}
    for k in sd:
        if "running" not in k:
            sd[k].requires_grad_(True)
    x = g["x"].clone().requires_grad_(True)
    ns = {}
    fn = R.c3 if tag == "c3" else R.conv_bn_silu
    y = fn(sd, "m.", x, True, ns)
    assert md(y, g["y"]) < 1e-5
    (y * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
    assert md(x.grad, g["dx"]) < 1e-4
    for k, v in g["grads"].items():
        assert md(sd["m." + k].grad, v) < 2e-4, k
    for k, v in g["stats_after"].items():
        assert md(ns["m." + k], v) < 1e-6
    sd2 = {k: v.detach() for k, v in sd.items()}
    sd2.update(ns)
    assert md(fn(sd2, "m.", g["x"], False, None), g["y_eval"]) < 1e-5


def test_full_model_512_against_reference_golden():
    g = torch.load(os.path.join(GOLD, "full_model_512.pt"))
    sd = R.procedural_state_dict(512, 8)
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
    x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=0)
    ns, taps = {}, {}
    pred, feats = R.model_forward(osd, x_rgb, x_ir, True, ns, taps)
    assert md(pred[0][:, :, ::8, ::8, :], g["logits_sub"]) < 1e-4
    for i in range(3):
        assert md(feats[i][..., ::8, ::8], g["feats_sub"][i]) < 1e-4
    for k, v in g["taps_sub"].items():
        t = taps[k]
        assert md(t.reshape(1, -1, t.shape[-1])[:, ::97, ::7], v) < 1e-4, k
    loss = pred[0].square().mean()
    assert abs(float(loss) - g["loss"]) < 1e-5 * abs(g["loss"])
    loss.backward()
    for k, gn in g["gnorm"].items():
        assert abs(float(osd[k].grad.double().norm()) - gn) <= 1e-4 * gn + 1e-7, k
    for k, v in g["stats_after_sub"].items():
        assert md(ns[k][::8], v) < 1e-5
    osd2 = {k: v.detach() for k, v in osd.items()}
    osd2.update(ns)
    with torch.no_grad():
        z, _, _ = R.model_forward(osd2, x_rgb, x_ir, False)
    assert md(z[:, ::257, :], g["z_sub"]) < 1e-3


def test_greedy_nms_spec():
    boxes = torch.tensor([[0., 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5], [21, 21, 29, 29]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95, 0.6])
    keep = R.greedy_nms(boxes, scores, 0.5)
    assert keep.tolist() == [3, 2]          # 0,1 suppressed by 3 (IoU > 0.5); 4 suppressed by 2
    keep = R.greedy_nms(boxes, scores, 0.99)
    assert keep.tolist() == [3, 0, 1, 2, 4]


def test_nms_oracle_matches_reference_golden():
    """oracle non_max_suppression vs the outputs of the reference's own function (general.py:425), nms.pt."""
    cases = torch.load(os.path.join(GOLD, "nms.pt"))
    assert len(cases) >= 8 and sum("labels" in c for c in cases) >= 2
    for c in cases:
        z = R.synthetic_predictions(c["B"], c["N"], c["nc"], seed=c["seed"])
        out, idx = R.non_max_suppression(z, c["conf"], c["iou"], classes=c["classes"], agnostic=c["agnostic"],
                                         multi_label=c["multi_label"], return_index=True, labels=c.get("labels", ()))
        for o, i, ro, ri in zip(out, idx, c["out"], c["index"]):
            assert o.shape == ro.shape
            assert torch.equal(i, ri)
            if o.numel():
                assert float((o - ro).abs().max()) < 1e-3


def test_compute_loss_matches_reference_goldens():
    """oracle compute_loss / build_targets / CIoU against the reference's own ComputeLoss outputs (tests/golden/loss.pt,
    written by oracle/gen_golden.py from basics/utils/loss.py): values and d(loss)/d(pred), incl. duplicate cells and no targets."""
    cases = torch.load(os.path.join(GOLD, "loss.pt"))
    for c in cases:
        pred = c["pred"].clone().requires_grad_(True)
        out = R.compute_loss(pred, c["targets"], c["anchors"], c["hyp"], c["gr"])
        out[0].backward()
        for a, b in zip(out, c["out"]):
            assert float((a.detach().reshape(-1) - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
        assert float((pred.grad - c["dpred"]).abs().max()) <= 1e-7


def test_spp_matches_reference_golden():
    """oracle spp() against the reference's common.SPP (tests/golden/spp.pt): output and input gradient."""
    g = torch.load(os.path.join(GOLD, "spp.pt"))
    sd = {k: v.clone() for k, v in g["sd"].items()}
    x = g["x"].clone().requires_grad_(True)
    ns = {}
    y = R.spp(sd, "", x, True, ns)
    (y * g["gsel"]).sum().backward()
    assert float((y - g["y"]).abs().max()) < 1e-5 and float((x.grad - g["dx"]).abs().max()) < 1e-5
    for k, v in g["stats_after"].items():
        assert float((ns[k] - v).abs().max()) < 1e-5


@pytest.mark.parametrize("name", ["decoder", "edsr", "deeplab"])
def test_sr_branch_matches_reference_golden(name):
    """oracle sr_decoder / edsr / deeplab_sr against the reference's own Decoder / EDSR / DeepLab classes (tests/golden/sr.pt,
    oracle/gen_golden.py --only-sr; sr_decoder_noBN_noD.py:6-45, edsr.py:55-102, deeplabedsr.py:35-73): output, input
    gradients, every parameter's gradient norm and 64 strided gradient values, on the procedural weights."""
    g = torch.load(os.path.join(GOLD, "sr.pt"))[name]
    sd = {k: v.requires_grad_(True) for k, v in R.procedural_from_shapes(g["shapes"]).items()}
    ins = [t.clone().requires_grad_(True) for t in g["inputs"]]
    if name == "decoder":
        y = R.sr_decoder(sd, "sr_decoder.", ins[1], ins[0], 2)
    elif name == "edsr":
        y = R.edsr(sd, "edsr.", ins[0])
    else:
        y = R.deeplab_sr(sd, "model_up.", ins[0], ins[1], 2)
    gsel = R._hash01("sr:" + name, y.numel()).view(y.shape).float()
    (y * gsel).sum().backward()
    st = g["y_step"]
    assert float((y.detach()[..., ::st, ::st] - g["y_sub"]).abs().max()) <= 1e-4 * max(1.0, g["y_absmax"])
    for a, b in zip(ins, g["dinputs"]):
        assert float((a.grad - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    for k, v in g["gnorm"].items():
        assert abs(float(sd[k].grad.double().norm()) - v) <= 1e-4 * (v + 1e-9), k
        got = sd[k].grad.reshape(-1)[::max(1, sd[k].numel() // 64)][:64]
        assert float((got - g["gsub"][k]).abs().max()) <= 1e-4 * (float(g["gsub"][k].abs().max()) + 1e-6), k
