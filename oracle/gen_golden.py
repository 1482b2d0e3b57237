"""Pin the oracle (oracle/ref_torch.py) against the REAL reference and emit golden
fixtures under tests/golden/.  Runs only in the build container, where
/root/reference exists; nothing here travels to the GPU box except the small .pt
fixtures it writes.  The reference is imported on CPU with stub modules for the
third-party packages that are absent (timm, cv2, torchvision, seaborn, numba) --
the recipe recorded in SURVEY.md section 8(c).

usage: python oracle/gen_golden.py [--skip-full]
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_torch as R  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x
    stub("timm")
    stub("timm.models")
    stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: (x, x) if not isinstance(x, tuple) else x,
         trunc_normal_=torch.nn.init.trunc_normal_)
    stub("cv2", setNumThreads=lambda n: None, ocl=types.SimpleNamespace(setUseOpenCL=lambda b: None))
    tv = stub("torchvision")
    tv.ops = stub("torchvision.ops")
    tv.transforms = stub("torchvision.transforms")
    tv.models = stub("torchvision.models")
    stub("seaborn")
    stub("numba", jit=lambda *a, **k: (lambda f: f))
    if "/root" not in sys.path:
        sys.path.insert(0, "/root")
    return importlib.import_module("reference.basics.models.model"), \
        importlib.import_module("reference.basics.models.backbone_vit"), \
        importlib.import_module("reference.basics.models.common")


def load_into(module: torch.nn.Module, sd, prefix=""):
    own = module.state_dict()
    for k, v in own.items():
        if v.dtype.is_floating_point and (prefix + k) in sd:
            v.copy_(sd[prefix + k].to(v.dtype))


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def sub(t, step):
    return t[..., ::step, ::step].contiguous() if t.dim() == 4 else t


def tiny_sd(mod: torch.nn.Module, seed: int):
    """Fill a tiny reference module with closed-form values and return its float state."""
    out = {}
    with torch.no_grad():
        for k, v in mod.state_dict().items():
            if not v.dtype.is_floating_point:
                continue
            if k.endswith("attn_mask"):
                continue
            u = R._hash01(f"{seed}:{k}", v.numel()).view(v.shape).to(v.dtype)
            if k.endswith("running_var"):
                val = 1.0 + 0.25 * u
            elif ("norm" in k or "bn" in k) and k.endswith("weight"):
                val = 1.0 + 0.2 * u
            elif v.dim() >= 2:
                fan = v[0].numel()
                val = u * (1.5 / fan ** 0.5)
            else:
                val = 0.2 * u
            v.copy_(val)
            out[k] = val.clone()
    return out


def per_module_goldens(ref_vit, ref_common, out):
    """Tiny instantiations of the reference's own classes (KB-sized fixtures)."""
    torch.manual_seed(0)
    g = {}
    # ---- SwinTransformerBlock: (shift 0, linear), (shift 2, conv), window-clamped
    for tag, kw, (H, W) in (
        ("swin_lin", dict(dim=24, num_heads=12, window_size=8, shift_size=0, linear_mlp=True), (16, 16)),
        ("swin_conv", dict(dim=24, num_heads=12, window_size=8, shift_size=2, linear_mlp=False), (16, 24)),
        ("swin_clamp", dict(dim=24, num_heads=12, window_size=32, shift_size=0, linear_mlp=True), (8, 8)),
    ):
        blk = ref_vit.SwinTransformerBlock(input_resolution=(H, W), **kw)
        sd = tiny_sd(blk, 1)
        x = R._hash01(tag, 2 * H * W * 24).view(2, H * W, 24).float().requires_grad_(True)
        y = blk(x)
        (y * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
        grads = {k: p.grad.clone() for k, p in blk.named_parameters()}
        # oracle check
        osd = {"b." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.detach().clone().requires_grad_(True)
        yo = R.swin_block(osd, "b.", xo, H, W, kw["window_size"], kw["shift_size"], kw["linear_mlp"])
        (yo * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
        d = maxdiff(y, yo)
        dg = max(maxdiff(grads[k], osd["b." + k].grad) for k in grads)
        dx = maxdiff(x.grad, xo.grad)
        print(f"[pin] {tag}: fwd {d:.2e} dparam {dg:.2e} dx {dx:.2e}")
        assert d < 1e-5 and dg < 1e-4 and dx < 1e-5, tag
        g[tag] = dict(cfg=dict(H=H, W=W, **kw), sd=sd, x=x.detach(), y=y.detach(), dx=x.grad.clone(), grads=grads)
    # ---- PatchMerging
    pm = ref_vit.PatchMerging((8, 12), 16)
    sd = tiny_sd(pm, 2)
    x = R._hash01("pm", 2 * 96 * 16).view(2, 96, 16).float()
    y = pm(x, (8, 12))
    yo = R.patch_merging({"p." + k: v for k, v in sd.items()}, "p.", x, 8, 12)
    print(f"[pin] patch_merging {maxdiff(y, yo):.2e}")
    assert maxdiff(y, yo) < 1e-5
    g["pmerge"] = dict(sd=sd, x=x, y=y.detach(), H=8, W=12)
    # ---- CAttentionBlock at window 1 (shipped) and general (window 2, shift 0 / 1)
    for tag, ws, shift in (("ca_w1", 1, 0), ("ca_w2", 2, 0), ("ca_w2s1", 2, 1)):
        blk = ref_vit.CAttentionBlock(embedding_dim=48, num_heads=12, shift_size=shift)
        blk.window_size = ws
        blk.input_resolution = (8, 8)
        if shift > 0:   # rebuild the mask for the overridden window / resolution (ctor lines :441-459)
            blk.attn_mask = R.shift_mask(8, 8, ws, shift)
        sd = tiny_sd(blk, 3)
        ins = [R._hash01(tag + str(i), 2 * 64 * 48).view(2, 8, 8, 48).float() for i in range(4)]
        outs = blk(*ins)
        oo = R.cattention_block({"c." + k: v for k, v in sd.items()}, *ins, window_size=ws, shift=shift, pfx="c.")
        d = max(maxdiff(a, b) for a, b in zip(outs, oo))
        print(f"[pin] {tag} {d:.2e}")
        assert d < 1e-5
        g[tag] = dict(sd=sd, ins=ins, outs=[o.detach() for o in outs], ws=ws, shift=shift)
    # ---- PatchEmbed pad 1 vs 0
    for tag, pad in (("pe_pad1", (1, 1)), ("pe_pad0", (0, 0))):
        pe = ref_vit.PatchEmbed(kernel_size=(4, 4), stride=(4, 4), padding=pad, in_chans=1, embed_dim=48)
        sd = tiny_sd(pe, 4)
        x = R._hash01(tag, 2 * 32 * 32).view(2, 1, 32, 32).float()
        g[tag] = dict(sd=sd, x=x, y=pe(x).detach())
    # ---- Conv / C3 train (batch-stat BN) + eval
    for tag, mk in (("conv1", lambda: ref_common.Conv(16, 24, 1)), ("conv3", lambda: ref_common.Conv(16, 24, 3)),
                    ("c3", lambda: ref_common.C3(24, 16, 1, False))):
        m = mk()
        for mm in m.modules():
            if isinstance(mm, torch.nn.BatchNorm2d):
                mm.eps, mm.momentum = R.BN_EPS, R.BN_MOMENTUM
        sd = tiny_sd(m, 5)
        cin = 16 if tag != "c3" else 24
        x = R._hash01(tag, 2 * cin * 8 * 12).view(2, cin, 8, 12).float().requires_grad_(True)
        m.train()
        y = m(x)
        (y * R._hash01(tag + "g", y.numel()).view(y.shape).float()).sum().backward()
        grads = {k: p.grad.clone() for k, p in m.named_parameters()}
        after = {k: v.clone() for k, v in m.state_dict().items() if "running" in k}
        ns = {}
        osd = {"m." + k: v for k, v in sd.items()}
        yo = R.conv_bn_silu(osd, "m.", x.detach(), True, ns) if tag != "c3" else R.c3(osd, "m.", x.detach(), True, ns)
        d = maxdiff(y, yo)
        ds = max(maxdiff(after[k], ns["m." + k]) for k in after)
        m.eval()
        ye = m(x.detach())
        osd2 = dict(osd)
        osd2.update(ns)
        yeo = R.conv_bn_silu(osd2, "m.", x.detach(), False, None) if tag != "c3" else R.c3(osd2, "m.", x.detach(), False, None)
        print(f"[pin] {tag}: train {d:.2e} stats {ds:.2e} eval {maxdiff(ye, yeo):.2e}")
        assert d < 1e-5 and ds < 1e-6 and maxdiff(ye, yeo) < 1e-5
        g[tag] = dict(sd=sd, x=x.detach(), y=y.detach(), dx=x.grad.clone(), grads=grads, stats_after=after, y_eval=ye.detach())
    torch.save(g, os.path.join(out, "per_module.pt"))


def full_model_goldens(ref_model, out):
    """Whole model @512^2 with procedural weights."""
    Model = ref_model.Model
    torch.manual_seed(0)
    m = Model("/root/reference/models/model.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
    sd = R.procedural_state_dict(512, 8)
    keys_ref = {k for k, v in m.state_dict().items() if v.dtype.is_floating_point and not k.endswith("attn_mask")}
    assert keys_ref == set(sd.keys()), (sorted(keys_ref - set(sd))[:5], sorted(set(sd) - keys_ref)[:5])
    nparams = sum(p.numel() for p in m.parameters())
    assert nparams == 22007851, nparams
    with torch.no_grad():
        load_into(m, sd)
    x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=0)
    m.train()
    pred, feats = m(x_rgb, x_ir, "RGB+IR")
    loss = pred[0].float().square().mean()
    loss.backward()
    gnorm = {k: float(p.grad.double().norm()) for k, p in m.named_parameters()}
    # sub-sampled gradient VALUES of the reference (a permuted or sign-flipped gradient of equal norm must not pass): up to
    # 64 elements per parameter at a stride that covers the whole tensor
    gsub = {k: p.grad.detach().reshape(-1)[::max(1, p.numel() // 64)][:64].clone() for k, p in m.named_parameters()}
    stats_after = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k}

    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k)
           for k, v in sd.items()}
    ns, taps = {}, {}
    opred, ofeats = R.model_forward(osd, x_rgb, x_ir, True, ns, taps)
    oloss = opred[0].square().mean()
    oloss.backward()
    d = maxdiff(pred[0], opred[0])
    print(f"[pin] full model train logits maxdiff {d:.3e} (|logit| max {float(pred[0].abs().max()):.2f})")
    assert d < 1e-4
    for i in range(3):
        di = maxdiff(feats[i], ofeats[i])
        print(f"[pin] encoder out {i}: {di:.3e}")
        assert di < 1e-4
    # stage3.0.mlp.fc2.bias has a mathematically zero gradient (a per-channel constant in
    # front of a bias-free 1x1 conv followed by batch-stat BN), so compare with an absolute floor.
    dgn = max((abs(gnorm[k] - float(osd[k].grad.double().norm())) - 1e-7) / (gnorm[k] + 1e-12) for k in gnorm)
    print(f"[pin] grad-norm rel diff max {dgn:.3e} (after 1e-7 abs floor)")
    assert dgn < 1e-4
    dst = max(maxdiff(stats_after[k], ns[k]) for k in stats_after)
    print(f"[pin] BN running stats {dst:.2e}")
    assert dst < 1e-5
    m.eval()
    with torch.no_grad():
        z, praw, _ = m(x_rgb, x_ir, "RGB+IR")
        osd2 = {k: v.detach() for k, v in osd.items()}
        osd2.update(ns)
        oz, opraw, _ = R.model_forward(osd2, x_rgb, x_ir, False)
    print(f"[pin] eval decode maxdiff {maxdiff(z, oz):.3e}")
    assert maxdiff(z, oz) < 1e-3
    gold = dict(
        img_size=512, B=1, seed=0,
        logits_sub=pred[0].detach()[:, :, ::8, ::8, :].contiguous(),
        feats_sub=[sub(f.detach(), 8) for f in feats[:3]],
        taps_sub={k: v.detach().view(1, -1, v.shape[-1])[:, ::97, ::7].contiguous() for k, v in taps.items()},
        loss=float(loss), gnorm=gnorm, gsub=gsub,
        stats_after_sub={k: v[::8].clone() for k, v in stats_after.items()},
        z_sub=z[:, ::257, :].contiguous(),
    )
    torch.save(gold, os.path.join(out, "full_model_512.pt"))
    print("[pin] wrote full_model_512.pt")


def autocast_goldens(ref_model, out):
    """What bf16 costs the REFERENCE ITSELF: the real Model @512^2 (procedural weights, the inputs of full_model_512.pt)
    under torch.autocast(cpu, bfloat16) - the reference trains under amp.autocast (Train.py:405) - against its own f32 run:
    logits and per-parameter gradient errors.  The bf16 throughput path of the build is gated at a multiple of these."""
    Model = ref_model.Model
    torch.manual_seed(0)
    m = Model("/root/reference/models/model.yaml", input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
    sd = R.procedural_state_dict(512, 8)
    with torch.no_grad():
        load_into(m, sd)
    x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=0)
    m.train()
    res = {}
    for name, ctx in (("f32", None), ("bf16", torch.autocast("cpu", dtype=torch.bfloat16))):
        for p in m.parameters():
            p.grad = None
        with torch.no_grad():
            load_into(m, sd)                      # BN running stats back to the start
        if ctx is None:
            pred, _ = m(x_rgb, x_ir, "RGB+IR")
        else:
            with ctx:
                pred, _ = m(x_rgb, x_ir, "RGB+IR")
        pred[0].float().square().mean().backward()
        res[name] = (pred[0].detach().float().clone(), {k: p.grad.detach().double().clone() for k, p in m.named_parameters()})
        print(f"[pin] reference {name} forward+backward done")
    (lf, gf), (lb, gb) = res["f32"], res["bf16"]
    gmed = sorted(float(g.norm()) for g in gf.values())[len(gf) // 4]
    grel = {k: float((gb[k] - gf[k]).norm() / (gf[k].norm() + 1e-2 * gmed + 1e-12)) for k in gf}
    gold = dict(img_size=512, B=1, seed=0, logit_max=float(lf.abs().max()), logit_maxdiff=float((lb - lf).abs().max()),
                logit_meandiff=float((lb - lf).abs().mean()), grad_rel=grel, grad_floor=1e-2 * gmed)
    worst = sorted(grel.items(), key=lambda kv: -kv[1])[:5]
    print(f"[pin] reference autocast(bf16) vs f32 @512^2: logits max|d| {gold['logit_maxdiff']:.3f} (|logit| max {gold['logit_max']:.2f}), "
          f"mean|d| {gold['logit_meandiff']:.4f}; worst gradient relative errors {[(k, round(v, 3)) for k, v in worst]}")
    torch.save(gold, os.path.join(out, "autocast_512.pt"))
    print("[pin] wrote autocast_512.pt")


def spp_goldens(ref_common, out):
    """common.SPP (c1 64 -> c2 96, train-mode BN) on a 2 x 64 x 12 x 10 map: output, input gradient, parameter gradients and
    the oracle's restatement against them.  Values are rounded to a coarse grid so that window maxima TIE often: the
    gradient routing of ties (first maximum in scan order) is part of what is pinned."""
    torch.manual_seed(7)
    m = ref_common.SPP(64, 96)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps, mod.momentum = 1e-3, 0.03
    sd = tiny_sd(m, 31)
    with torch.no_grad():
        load_into(m, sd)
    x = (torch.randn(2, 64, 12, 10) * 2).round() / 2
    x.requires_grad_(True)
    m.train()
    y = m(x)
    gsel = R._hash01("spp", y.numel()).view(y.shape).float()
    (y * gsel).sum().backward()
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    xo = x.detach().clone().requires_grad_(True)
    yo = R.spp(osd, "", xo, True, {})
    (yo * gsel).sum().backward()
    print(f"[pin] SPP out {maxdiff(y, yo):.2e} dx {maxdiff(x.grad, xo.grad):.2e}")
    assert maxdiff(y, yo) < 1e-5 and maxdiff(x.grad, xo.grad) < 1e-5
    gold = dict(sd={k: v.clone() for k, v in sd.items()}, x=x.detach().clone(), y=y.detach().clone(), gsel=gsel, dx=x.grad.clone(),
                grads={k: p.grad.clone() for k, p in m.named_parameters()},
                stats_after={k: v.clone() for k, v in m.state_dict().items() if "running_" in k})
    torch.save(gold, os.path.join(out, "spp.pt"))
    print("[pin] wrote spp.pt")


def nms_goldens(out):
    """Run the reference's own non_max_suppression (general.py:425) - with torchvision.ops.nms, absent from
    this image, bound to the published greedy algorithm (R.greedy_nms) - and pin the oracle's restatement."""
    sys.modules["torchvision.ops"].nms = R.greedy_nms
    sys.modules["torchvision"].ops.nms = R.greedy_nms
    G = importlib.import_module("reference.basics.utils.general")
    cases = []
    for (B, N, nc, conf, iou, ml, agn, classes, seed) in [
            (2, 1200, 8, 0.25, 0.45, True, False, None, 1),      # n < 3000: merge-NMS + redundancy filter
            (1, 6000, 8, 0.001, 0.6, True, False, None, 2),      # test.py:145 settings, n > 30000: truncation
            (2, 1500, 8, 0.25, 0.45, False, False, None, 3),     # best-class path
            (1, 1500, 8, 0.25, 0.45, True, True, [1, 5], 4),     # agnostic + class filter
            (1, 64, 1, 0.25, 0.45, True, False, None, 5),        # nc == 1 (multi_label forced off)
            (1, 50, 8, 0.999, 0.45, True, False, None, 6)]:      # nothing passes
        z = R.synthetic_predictions(B, N, nc, seed=seed)
        ref = G.non_max_suppression(z.clone(), conf, iou, classes=classes, agnostic=agn, multi_label=ml)
        mine, idx = R.non_max_suppression(z.clone(), conf, iou, classes=classes, agnostic=agn, multi_label=ml, return_index=True)
        for a, b in zip(ref, mine):
            assert a.shape == b.shape, (a.shape, b.shape)
            if a.numel():
                assert maxdiff(a, b) < 1e-3, maxdiff(a, b)
        print(f"[pin] NMS B={B} N={N} nc={nc} conf={conf} multi_label={ml}: kept {[int(a.shape[0]) for a in ref]} identical")
        cases.append(dict(B=B, N=N, nc=nc, conf=conf, iou=iou, multi_label=ml, agnostic=agn, classes=classes, seed=seed,
                          out=[a.clone() for a in ref], index=[i.clone() for i in idx]))
    # autolabelling rows (general.py:451-458, `save_hybrid` in test.py:143-145): labels appended after the confidence filter
    for (B, N, nc, conf, iou, ml, seed) in [(2, 1200, 8, 0.25, 0.45, True, 7), (2, 300, 8, 0.5, 0.45, False, 8)]:
        z = R.synthetic_predictions(B, N, nc, seed=seed)
        gl = torch.Generator().manual_seed(seed)
        lab = []
        for b in range(B):
            nl = 6 if b == 0 else 0                      # second image: no labels
            l = torch.zeros(nl, 5)
            if nl:
                l[:, 0] = torch.randint(0, nc, (nl,), generator=gl).float()
                l[:3, 1:5] = z[b, :3, :4]                # three labels on top of predictions (they suppress / merge)
                l[3:, 1:3] = torch.rand(nl - 3, 2, generator=gl) * 900 + 50
                l[3:, 3:5] = torch.rand(nl - 3, 2, generator=gl) * 40 + 10
            lab.append(l)
        ref = G.non_max_suppression(z.clone(), conf, iou, multi_label=ml, labels=lab)
        mine, idx = R.non_max_suppression(z.clone(), conf, iou, multi_label=ml, return_index=True, labels=lab)
        for a, b in zip(ref, mine):
            assert a.shape == b.shape, (a.shape, b.shape)
            if a.numel():
                assert maxdiff(a, b) < 1e-3, maxdiff(a, b)
        print(f"[pin] NMS with labels B={B} N={N} conf={conf} multi_label={ml}: kept {[int(a.shape[0]) for a in ref]} identical")
        cases.append(dict(B=B, N=N, nc=nc, conf=conf, iou=iou, multi_label=ml, agnostic=False, classes=None, seed=seed,
                          labels=[l.clone() for l in lab], out=[a.clone() for a in ref], index=[i.clone() for i in idx]))
    torch.save(cases, os.path.join(out, "nms.pt"))
    print("[pin] wrote nms.pt")


def loss_goldens(out):
    """Run the reference's own ComputeLoss (basics/utils/loss.py:90-224, float32 as in training) on fixed head outputs and
    targets, pin the oracle's compute_loss against it and store inputs + outputs + d(loss)/d(pred)."""
    LM = importlib.import_module("reference.basics.utils.loss")

    class _Det:
        pass

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
    m, det = _M(), _Det()
    det.nl, det.na, det.nc, det.stride = 1, 3, 8, torch.tensor([4.])
    det.anchors = torch.tensor([[[10., 13.], [16., 30.], [33., 23.]]]) / 4          # models/model.yaml:8, model.py:131
    m.detect, m.hyp, m.gr = [det], dict(R.LOSS_HYP), 1.0
    cl = LM.ComputeLoss(m)
    cases = []
    # 30 boxes on 8x8: duplicate cells; last case: targets on / over the right and bottom border (x or y == 1.0): the
    # reference clamps the cell index IN PLACE before tbox = gxy - gij (loss.py:219-221), so tbox uses the clamped cell
    for seed, (B, t, per) in enumerate([(2, 16, 12), (1, 32, 40), (2, 16, 0), (2, 8, 30), (2, 16, 10)]):
        torch.manual_seed(100 + seed)
        pred = torch.randn(B, 3, t, t, 13, requires_grad=True)
        tg = R.synthetic_targets(B, per, 8, seed) if per else torch.zeros(0, 6)
        if (B, t) == (2, 8):
            tg[:, 4:6] *= 8.0                                                      # boxes large enough for the 8x8 grid's anchors
        if seed == 4:
            tg[:, 4:6] = tg[:, 4:6] * 4.0 + 0.15                                   # wide enough to pass anchor_t on a 16-grid
            tg[0::3, 2] = 1.0                                                      # x on the right border
            tg[1::3, 3] = 1.0                                                      # y on the bottom border
            tg[2, 2:4] = 1.0
        ref = cl([pred], tg)
        ref[0].backward()
        dref = pred.grad.clone()
        pred.grad = None
        mine = R.compute_loss(pred, tg, det.anchors[0])
        mine[0].backward()
        dl = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(ref, mine))
        dg = maxdiff(dref, pred.grad)
        print(f"[pin] ComputeLoss B={B} grid {t} targets {tg.shape[0]}: loss {float(ref[0]):.5f} value diff {dl:.2e} grad diff {dg:.2e}")
        assert dl < 1e-5 and dg < 1e-6
        cases.append(dict(pred=pred.detach().clone(), targets=tg.clone(), anchors=det.anchors[0].clone(), hyp=dict(R.LOSS_HYP), gr=1.0,
                          out=[x.detach().clone().reshape(-1) for x in ref], dpred=dref))
    torch.save(cases, os.path.join(out, "loss.pt"))
    print("[pin] wrote loss.pt")


SPP_HEAD = [[2, 1, "Conv", [512, 1, 1]], [-1, 1, "SPP", [512, [5, 9, 13]]], [-1, 1, "nn.Upsample", [None, 2, "nearest"]],
            [[-1, 1], 1, "Concat", [1]], [-1, 3, "C3", [512, False]], [-1, 1, "Conv", [256, 1, 1]],
            [-1, 1, "nn.Upsample", [None, 2, "nearest"]], [[-1, 0], 1, "Concat", [1]], [-1, 3, "C3", [256, False]],
            [[11], 1, "Detect", ["nc", "anchors"]]]


def spp_head_goldens(ref_model, out):
    """A head yaml with an SPP row (common.py:129-140; the row SRyolo_MF.yaml:46 uses) after detect.0: the reference's own
    parse_model / forward_once build and run it (model.py:350-435, :268-281).  Pins the oracle's generic head walk
    (ref_torch.head_graph) against the real reference at 512^2 and stores sub-sampled logits, the loss, gradient norms and
    sub-sampled gradient values.  Weights: ref_torch.procedural_from_shapes over the reference model's own float
    state_dict shapes (reproducible on the GPU box from the build's Model with the same cfg)."""
    import yaml
    Model = ref_model.Model
    cfg = yaml.safe_load(open("/root/reference/models/model.yaml"))
    cfg["head"] = [list(r) for r in SPP_HEAD]
    m = Model(cfg, input_mode="RGB+IR", ch_steam=3, ch=128, nc=8)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()
              if v.dtype.is_floating_point and not k.endswith("attn_mask")}
    sd = R.procedural_from_shapes(shapes)
    with torch.no_grad():
        load_into(m, sd)
    x_rgb, x_ir = R.synthetic_inputs(1, 512, seed=9)
    m.train()
    pred, feats = m(x_rgb, x_ir, "RGB+IR")
    loss = pred[0].float().square().mean()
    loss.backward()
    gnorm = {k: float(p.grad.double().norm()) for k, p in m.named_parameters()}
    gsub = {k: p.grad.detach().reshape(-1)[::max(1, p.numel() // 64)][:64].clone() for k, p in m.named_parameters()
            if k.startswith("detect.") or "neck" in k or "stage3.0.attn.qkv" in k}
    osd = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and "anchor" not in k) for k, v in sd.items()}
    opred, ofeats = R.model_forward(osd, x_rgb, x_ir, True, {}, head_rows=SPP_HEAD)
    opred[0].square().mean().backward()
    d = maxdiff(pred[0], opred[0])
    dgn = max((abs(gnorm[k] - float(osd[k].grad.double().norm())) - 1e-7) / (gnorm[k] + 1e-12) for k in gnorm)
    print(f"[pin] SPP head @512^2: logits maxdiff {d:.3e} (|logit| max {float(pred[0].abs().max()):.2f}); y[4] (SPP out) "
          f"{maxdiff(feats[4], ofeats[4]):.3e}; grad-norm rel diff {dgn:.3e}")
    assert d < 1e-4 and dgn < 1e-4 and maxdiff(feats[4], ofeats[4]) < 1e-4
    # the model.yaml head through the generic walk equals the hard-wired statement
    rows0 = yaml.safe_load(open("/root/reference/models/model.yaml"))["head"]
    sd0 = R.procedural_state_dict(128, 8)
    a0, b0 = R.synthetic_inputs(1, 128, seed=2)
    p1, _ = R.model_forward(sd0, a0, b0, True, {})
    p2, _ = R.model_forward(sd0, a0, b0, True, {}, head_rows=rows0)
    assert torch.equal(p1[0], p2[0])
    torch.save(dict(img_size=512, seed=9, head=SPP_HEAD, logits_sub=pred[0].detach()[:, :, ::8, ::8, :].contiguous(),
                    spp_out_sub=sub(feats[4].detach(), 4), loss=float(loss), gnorm=gnorm, gsub=gsub,
                    nkeys=len(shapes)), os.path.join(out, "spp_head_512.pt"))
    print("[pin] wrote spp_head_512.pt")


def sr_goldens(out):
    """The super-resolution auxiliary branch from the reference's OWN classes (basics/models/sr_decoder_noBN_noD.py:6-45,
    edsr.py:55-102, deeplabedsr.py:35-73; importable although Model(sr=True) itself cannot reach them, SURVEY.md section 8
    config reality row 5): Decoder(c1 16, c2 32), EDSR(4, 64, factor 8, depth 2) and the full DeepLab(4, 128, 512, factor 2) -
    EDSR depth 16, 2.9 M parameters - on small maps.  Weights are the PROCEDURAL ones of ref_torch.procedural_from_shapes
    (re-creatable anywhere from names and shapes), so only inputs, outputs, input gradients, gradient norms and 64 strided
    gradient values per parameter are stored.  Each case pins the oracle restatement (0.0 difference)."""
    dec_m = importlib.import_module("reference.basics.models.sr_decoder_noBN_noD")
    edsr_m = importlib.import_module("reference.basics.models.edsr")
    dl_m = importlib.import_module("reference.basics.models.deeplabedsr")
    gold = {}

    def case(name, m, pfx, inputs, run_ref, run_oracle):
        shapes = {pfx + k: tuple(v.shape) for k, v in m.state_dict().items() if v.dtype.is_floating_point}
        psd = R.procedural_from_shapes(shapes)
        with torch.no_grad():
            load_into(m, psd, prefix=pfx)
        ins = [t.clone().requires_grad_(True) for t in inputs]
        y = run_ref(m, *ins)
        gsel = R._hash01("sr:" + name, y.numel()).view(y.shape).float()
        (y * gsel).sum().backward()
        osd = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        ins2 = [t.clone().requires_grad_(True) for t in inputs]
        yo = run_oracle(osd, *ins2)
        (yo * gsel).sum().backward()
        gn = {pfx + k: float(p.grad.double().norm()) for k, p in m.named_parameters()}
        gd = max(abs(float(osd[k].grad.double().norm()) - v) / (v + 1e-9) for k, v in gn.items())
        di = max(maxdiff(a.grad, b.grad) for a, b in zip(ins, ins2))
        print(f"[pin] SR {name}: out {tuple(y.shape)} diff {maxdiff(y, yo):.2e} (|y| max {float(y.abs().max()):.2f}), input grads {di:.2e}, "
              f"grad norms (rel) {gd:.2e}; {sum(p.numel() for p in m.parameters())} parameters")
        assert maxdiff(y, yo) <= 1e-4 * max(1.0, float(y.abs().max())) and gd <= 1e-4
        gsub = {pfx + k: p.grad.detach().reshape(-1)[::max(1, p.numel() // 64)][:64].clone() for k, p in m.named_parameters()}
        step = 4 if y.shape[-1] > 64 else 1
        gold[name] = dict(shapes=shapes, inputs=[t.detach().clone() for t in inputs], y_sub=y.detach()[..., ::step, ::step].contiguous().clone(),
                          y_step=step, y_absmax=float(y.abs().max()), dinputs=[t.grad.clone() for t in ins], gnorm=gn, gsub=gsub)

    h = lambda tag, *shape: (R._hash01("srin:" + tag, int(torch.tensor(shape).prod())).view(*shape).float() * 2)
    case("decoder", dec_m.Decoder(16, 32), "sr_decoder.", [h("dl", 2, 16, 12, 10), h("dx", 2, 32, 6, 5)],
         lambda m, low, x: m(x, low, 2), lambda sd, low, x: R.sr_decoder(sd, "sr_decoder.", x, low, 2))
    case("edsr", edsr_m.EDSR(num_channels=4, input_channel=64, factor=8, depth=2), "edsr.", [h("ex", 1, 64, 7, 6)],
         lambda m, x: m(x), lambda sd, x: R.edsr(sd, "edsr.", x))
    # as Model would build it (model.py:113-115 with c1 = 128, c2 = 512): low-level feature 128 @ t, x 512 @ t / 2
    case("deeplab", dl_m.DeepLab(4, 128, 512, factor=2), "model_up.", [h("low", 1, 128, 16, 16), h("x", 1, 512, 8, 8)],
         lambda m, low, x: m(low, x), lambda sd, low, x: R.deeplab_sr(sd, "model_up.", low, x, 2))
    torch.save(gold, os.path.join(out, "sr.pt"))
    print("[pin] wrote sr.pt")


def pad_block_goldens(ref_vit, out):
    """SwinTransformerBlock at resolutions that are NOT a multiple of the window (backbone_vit.py:619-643: zero padding after norm1,
    cropped after the attention): unshifted / linear MLP and shifted / conv MLP.  Pins the oracle's padded window_partition; the
    engine runs the unshifted case (stage 3 at S = 640, 768, ...), tests/test_pad_gpu.py."""
    g = {}
    for tag, kw, (H, W) in (
        ("pad_lin", dict(dim=24, num_heads=12, window_size=8, shift_size=0, linear_mlp=True), (12, 20)),
        ("pad_conv_shift", dict(dim=24, num_heads=12, window_size=8, shift_size=2, linear_mlp=False), (12, 12)),
    ):
        blk = ref_vit.SwinTransformerBlock(input_resolution=(H, W), **kw)
        sd = tiny_sd(blk, 7)
        x = R._hash01(tag, 2 * H * W * 24).view(2, H * W, 24).float().requires_grad_(True)
        y = blk(x)
        gw = R._hash01(tag + "g", y.numel()).view(y.shape).float()
        (y * gw).sum().backward()
        grads = {k: p.grad.clone() for k, p in blk.named_parameters()}
        osd = {"b." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xo = x.detach().clone().requires_grad_(True)
        yo = R.swin_block(osd, "b.", xo, H, W, kw["window_size"], kw["shift_size"], kw["linear_mlp"])
        (yo * gw).sum().backward()
        d = maxdiff(y, yo)
        dg = max(maxdiff(grads[k], osd["b." + k].grad) for k in grads)
        dx = maxdiff(x.grad, xo.grad)
        print(f"[pin] {tag}: fwd {d:.2e} dparam {dg:.2e} dx {dx:.2e}")
        assert d < 1e-5 and dg < 1e-4 and dx < 1e-5, tag
        g[tag] = dict(cfg=dict(H=H, W=W, **kw), sd=sd, x=x.detach(), y=y.detach(), dx=x.grad.clone(), grads=grads)
    torch.save(g, os.path.join(out, "pad_block.pt"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only-nms", action="store_true")
    ap.add_argument("--only-loss", action="store_true")
    ap.add_argument("--only-autocast", action="store_true")
    ap.add_argument("--only-spp", action="store_true")
    ap.add_argument("--only-spp-head", action="store_true")
    ap.add_argument("--only-sr", action="store_true")
    ap.add_argument("--only-pad", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 8)
    ref_model, ref_vit, ref_common = import_reference()
    if a.only_loss:
        loss_goldens(GOLD)
        return
    if a.only_pad:
        pad_block_goldens(ref_vit, GOLD)
        return
    if a.only_autocast:
        autocast_goldens(ref_model, GOLD)
        return
    if a.only_spp:
        spp_goldens(ref_common, GOLD)
        return
    if a.only_spp_head:
        spp_head_goldens(ref_model, GOLD)
        return
    if a.only_sr:
        sr_goldens(GOLD)
        return
    loss_goldens(GOLD)
    nms_goldens(GOLD)
    if a.only_nms:
        return
    per_module_goldens(ref_vit, ref_common, GOLD)
    pad_block_goldens(ref_vit, GOLD)
    spp_goldens(ref_common, GOLD)
    sr_goldens(GOLD)
    if not a.skip_full:
        full_model_goldens(ref_model, GOLD)
        autocast_goldens(ref_model, GOLD)
        spp_head_goldens(ref_model, GOLD)


if __name__ == "__main__":
    main()
