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
