"""non_max_suppression through the C ABI (sodt_nms_candidates / sodt_nms_select) against the oracle restatement of
general.py:425-512 and the golden outputs of the reference's own function (tests/golden/nms.pt, written by
oracle/gen_golden.py).  Kept candidate ids must be IDENTICAL; merged box coordinates (an f32 matmul in the
reference) are compared to 1e-3 px."""
import importlib
import os

import pytest
import torch

from oracle import ref_torch as R

GOLD = os.path.join(os.path.dirname(__file__), "golden", "nms.pt")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nms(pkg):
    return importlib.import_module(pkg.__name__ + ".nms")


def _check(out, idx, ref_out, ref_idx):
    assert len(out) == len(ref_out)
    for o, i, ro, ri in zip(out, idx, ref_out, ref_idx):
        assert o.shape == ro.shape, (o.shape, ro.shape)
        assert torch.equal(i.cpu(), ri.long()), "kept candidate ids differ"
        if o.numel():
            assert torch.equal(o[:, 4:].cpu(), ro[:, 4:]), "conf / cls differ"
            assert float((o[:, :4].cpu() - ro[:, :4]).abs().max()) < 1e-3


def test_golden_cases(nms, dev):
    cases = torch.load(GOLD)
    for c in cases:
        z = R.synthetic_predictions(c["B"], c["N"], c["nc"], seed=c["seed"])
        out, idx = nms.non_max_suppression(z.to(dev), c["conf"], c["iou"], classes=c["classes"], agnostic=c["agnostic"],
                                           multi_label=c["multi_label"], return_index=True, labels=c.get("labels", ()))
        _check(out, idx, c["out"], c["index"])
    assert sum("labels" in c for c in cases) >= 2        # autolabelling rows (general.py:451-458) are pinned too


@pytest.mark.parametrize("N,conf,iou,ml", [(196608, 0.001, 0.6, True), (196608, 0.25, 0.45, False), (2500, 0.3, 0.45, True),
                                           (65, 0.01, 0.3, True), (1, 0.0, 0.5, True)])
def test_vs_oracle(nms, dev, N, conf, iou, ml):
    # 196608 = 3 anchors x 256 x 256 cells: the eval output of one 1024x1024 image (model.py:55-64, Detect stride 4)
    z = R.synthetic_predictions(1, N, 8, seed=N % 97, clusters=max(1, min(400, N // 20)))
    ro, ri = R.non_max_suppression(z.clone(), conf, iou, multi_label=ml, return_index=True)
    out, idx = nms.non_max_suppression(z.to(dev), conf, iou, multi_label=ml, return_index=True)
    _check(out, idx, ro, ri)


def test_ties_and_duplicates(nms, dev):
    """Identical boxes with identical scores: the first in candidate order survives (stable order)."""
    z = torch.zeros(1, 200, 13)
    z[0, :, :4] = torch.tensor([100.0, 100.0, 20.0, 30.0])
    z[0, :, 4] = 0.9
    z[0, :, 5 + 3] = 0.8
    z[0, 100:, :2] += 500.0
    ro, ri = R.non_max_suppression(z.clone(), 0.25, 0.45, multi_label=True, return_index=True)
    out, idx = nms.non_max_suppression(z.to(dev), 0.25, 0.45, multi_label=True, return_index=True)
    _check(out, idx, ro, ri)
    assert idx[0].tolist() == [0 * 8 + 3, 100 * 8 + 3]


def test_empty_and_errors(nms, dev):
    z = R.synthetic_predictions(2, 100, 8, seed=3)
    out = nms.non_max_suppression(z.to(dev), 0.9999, 0.45)
    assert all(o.shape == (0, 6) for o in out)
    with pytest.raises(RuntimeError):
        nms.non_max_suppression(z, 0.25, 0.45)                      # CPU tensor: no fallback
    with pytest.raises(ValueError):
        nms.non_max_suppression(z.to(dev), 0.25, 0.45, labels=[torch.zeros(1, 5)])      # one label tensor per image


def _peaked_logits(B, t, nc, seed, nobj=48):
    """Raw Detect logits (B, 3, t, t, 5 + nc) of a 'trained-like' detector: `nobj` objects per image, each a peak of the
    objectness / class logits on one anchor that falls off by 1.75 per cell of Chebyshev distance (so neighbouring cells of
    a peak, whose decoded boxes overlap, are separated by far more than a bf16 ulp: 2^-8 relative on |logit| <= 7 is 0.03),
    distinct peak heights 0.11 apart, background at -9.  Every decision non_max_suppression takes (confidence cut,
    suppression order, IoU test) has a margin >> the rounding of the logits to bf16."""
    g = torch.Generator().manual_seed(seed)
    raw = torch.full((B, 3, t, t, 5 + nc), -9.0)
    raw[..., 0:4] = 0.0
    yy, xx = torch.meshgrid(torch.arange(t), torch.arange(t), indexing="ij")
    for b in range(B):
        cells = torch.randperm((t // 8) * (t // 8), generator=g)[:nobj]          # one object per 8 x 8 cell block at most
        for k, c in enumerate(cells.tolist()):
            cy, cx = 8 * (c // (t // 8)) + 3, 8 * (c % (t // 8)) + 4
            a, cls = k % 3, (k * 5) % nc
            d = torch.maximum((yy - cy).abs(), (xx - cx).abs()).float()
            peak = 6.5 - 0.11 * (k % 24)
            obj = peak - 1.75 * d
            m = d <= 2
            raw[b, a][m, 4] = obj[m]
            raw[b, a][m, 5 + cls] = 4.0 - 0.5 * d[m]
            raw[b, a][m, 5 + (cls + 3) % nc] = -2.0 - 0.25 * (k % 5)
            # box offsets / sizes: coarse procedural values (exact in bf16), different per object
            raw[b, a][m, 0] = 0.25 * ((k % 7) - 3)
            raw[b, a][m, 1] = 0.25 * ((k % 5) - 2)
            raw[b, a][m, 2] = 0.5 * ((k % 4) - 1)
            raw[b, a][m, 3] = 0.5 * ((k % 3) - 1)
    return raw


@pytest.mark.parametrize("multi_label", [False, True])
def test_bf16_rounded_logits_give_identical_ids(nms, ops, dev, multi_label):
    """north_star: 'identical NMS indices'.  With scores separated by margins >> a bf16 ulp (a trained-like score landscape, not
    the near-tied scores of a random-init model), decode + non_max_suppression on the logits ROUNDED TO bf16 (the storage
    precision of the throughput path) keeps exactly the candidate ids of the f32 logits and of the oracle's restatement of
    Detect's eval branch + general.py:425-512 on the CPU."""
    B, t, nc = 2, 64, 8
    raw = _peaked_logits(B, t, nc, seed=11)
    ag = torch.tensor(R.ANCHORS_PX, dtype=torch.float32).view(-1)
    zo = R.detect_decode(raw, ag.view(3, 2))
    ro, ri = R.non_max_suppression(zo.clone(), 0.25, 0.45, multi_label=multi_label, return_index=True)
    assert all(len(i) >= 40 for i in ri)                       # the peaks survive, their shoulders are suppressed or cut
    ids = {}
    for name, r in (("f32", raw), ("bf16", raw.to(torch.bfloat16).float())):
        r = r.to(dev).contiguous()
        z = torch.empty(B, 3 * t * t, 5 + nc, device=dev)
        ops.detect_decode(r, ag.to(dev), z, B, 3, t, t, 5 + nc, 4.0)
        out, idx = nms.non_max_suppression(z, 0.25, 0.45, multi_label=multi_label, return_index=True)
        ids[name] = [i.cpu() for i in idx]
        for o, i, ro_, ri_ in zip(out, idx, ro, ri):
            assert torch.equal(i.cpu(), ri_.long()), f"{name}: kept candidate ids differ from the oracle's"
            assert float((o[:, :4].cpu() - ro_[:, :4]).abs().max()) < (1e-3 if name == "f32" else 0.5)
    assert all(torch.equal(a, b) for a, b in zip(ids["f32"], ids["bf16"]))
