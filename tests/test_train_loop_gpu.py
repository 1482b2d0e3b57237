"""The training step bench.py times, end to end on a small input: forward, YOLOv5 ComputeLoss (device kernels), hand-written
backward, FusedSGD with the weight-decay groups (basics/optimizer.py:35-49), ModelEMA - the loss must fall on a fixed batch and
the averaged model must evaluate (Train.py:405-453 in miniature)."""
import importlib

import pytest
import torch

from test_model_gpu import build

pytestmark = pytest.mark.gpu
PKG = "small-object-detection-transformers_amd"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_loss_falls_with_fused_optimizer_and_ema(dev, dt):
    O = importlib.import_module(PKG + ".optim")
    LS = importlib.import_module(PKG + ".loss")
    S, B = 256, 2
    model, _ = build(dev, S)
    model.compute_dtype = dt
    model.train()
    model.hyp, model.gr, model.nc = dict(LS.DEFAULT_HYP), 1.0, 8
    ema = O.ModelEMA(model)
    opt = O.FusedSGD(O.set_weight_decay(model), model=model, lr=0.01, momentum=0.937, nesterov=True, ema=ema)
    compute_loss = LS.ComputeLoss(model)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, 3, S, S, generator=g).to(dev)
    ir = torch.rand(B, 3, S, S, generator=g).to(dev)
    targets = LS.synthetic_targets(B, 16, 8, seed=1).to(dev)
    ls = []
    for _ in range(16):
        pred, _ = model(x, ir, "RGB+IR")
        loss = compute_loss(pred, targets)[0]
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        ema.update(model)
        ls.append(float(loss.detach()) / B)
    assert all(v == v for v in ls), ls
    assert ls[-1] < 0.9 * ls[0], ls
    assert ema.updates == 16
    ema.ema.eval()
    with torch.no_grad():
        z = ema.ema(x, ir, "RGB+IR")[0]
    assert torch.isfinite(z).all()


def test_hipgraph_replay_matches_plain_replay(dev, monkeypatch):
    """SODT_HIPGRAPH=1: the recorded forward / backward launch lists captured into hipGraphs give the same logits and gradients
    as the plain replay (three steps each: record, capture + first graph launch, graph launch)."""
    eng_mod = importlib.import_module(PKG + ".engine")
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 256, 256, generator=g).to(dev)
    ir = torch.rand(2, 3, 256, 256, generator=g).to(dev)
    res = {}
    for use in (False, True):
        monkeypatch.setattr(eng_mod, "USE_HIPGRAPH", use)
        model, _ = build(dev, 256)
        model.compute_dtype = torch.float32
        model.train()
        for _ in range(3):
            for p in model.parameters():
                p.grad = None
            pred, _ = model(x, ir, "RGB+IR")
            pred[0].float().square().mean().backward()
        torch.cuda.synchronize()
        eng = model._get_engine()
        plan = next(iter(eng.plans.values()))
        assert bool(plan.graphs) == use
        res[use] = (pred[0].detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5
    for k, gv in res[False][1].items():
        d = float((res[True][1][k] - gv).abs().max())
        assert d <= 2e-5 * max(1.0, float(gv.abs().max())), (k, d)
