"""Two data-parallel ranks through the real engine on ONE GPU (gloo moves the CUDA buffers; RCCL refuses two ranks on a
device): after ddp.attach, every rank's gradients equal the sum of the two ranks' single-process gradients - on the
recorded first step (one all-reduce) and on replayed steps (tail of the buffer all-reduced under the stage-1 backward)."""
import contextlib
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
PKG = "small-object-detection-transformers_amd"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(dev):
    import yaml
    M = importlib.import_module(PKG + ".model")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, PKG, "configs", "SRyolo_MF.yaml")))
    cfg["backbone"][0][3][0] = 128
    torch.manual_seed(0)
    m = M.Model(cfg, input_mode="RGB+IR", ch_steam=3, ch=128, nc=8).to(dev).train()
    m.compute_dtype = torch.float32
    return m


def _inputs(rank, dev):
    g = torch.Generator().manual_seed(100 + rank)
    return torch.rand(2, 3, 128, 128, generator=g).to(dev), torch.rand(2, 3, 128, 128, generator=g).to(dev)


def _grads(model, x, ir, steps):
    out = []
    for _ in range(steps):
        for p in model.parameters():
            p.grad = None
        pred, _ = model(x, ir, "RGB+IR")
        pred[0].float().square().mean().backward()
        torch.cuda.synchronize()
        out.append(model._get_engine().flat_grad.clone())
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ddp = importlib.import_module(PKG + ".ddp")
    # single-process references for both ranks' inputs (same initial weights everywhere)
    ref = _build(dev)
    want = sum(_grads(ref, *_inputs(r, dev), 1)[0] for r in range(world))
    model = _build(dev)
    ddp.attach(model, average=False)
    got = _grads(model, *_inputs(rank, dev), 3)          # step 1 records the plan, steps 2-3 replay with the overlap
    eng = model._get_engine()
    plan = next(iter(eng.plans.values()))
    scale = float(want.abs().max())
    errs = [float((g - want).abs().max()) / scale for g in got]
    ok = len(plan.bwd_marks) == 2 and 0 < eng.ddp_split < eng.ddp_split3 < eng.flat_grad.numel() and all(e < 2e-4 for e in errs)
    # gradient accumulation without zeroing in between (the reference accumulates 4 micro-steps, Train.py:125,448):
    # two backwards, reduced every time and with no_sync() on the first, SUM and MEAN - always (sum over ranks of 2 g) [/ world]
    for average in (False, True):
        for use_no_sync in (False, True):
            eng.ddp = ddp.GradReducer(average=average)
            for p in model.parameters():
                p.grad = None
            x, ir = _inputs(rank, dev)
            for k in range(2):
                ctx = eng.ddp.no_sync() if (use_no_sync and k == 0) else contextlib.nullcontext()
                with ctx:
                    pred, _ = model(x, ir, "RGB+IR")
                    pred[0].float().square().mean().backward()
            torch.cuda.synchronize()
            w2 = want * (2.0 / (world if average else 1))
            e = float((eng.flat_grad - w2).abs().max()) / float(w2.abs().max())
            errs.append(e)
            ok = ok and e < 2e-4
    # a forward that overwrites the saved activations before backward must raise, not give silent garbage
    for p in model.parameters():
        p.grad = None
    pred1, _ = model(*_inputs(rank, dev), "RGB+IR")
    model(*_inputs(rank, dev), "RGB+IR")
    try:
        pred1[0].float().square().mean().backward()
        ok = False
    except RuntimeError as ex:
        ok = ok and "overwritten" in str(ex)
    q.put((rank, ok, errs))
    dist.destroy_process_group()


def test_two_ranks_engine_overlap():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=600) for _ in range(2)]
    for p in ps:
        p.join(120)
    assert all(r[1] for r in res), res


def _nccl_worker(port, q):
    """RCCL itself on the hardware: init_process_group('nccl') with world_size 1 on the one GPU, real all-reduce calls
    through GradReducer's three entry points (world is forced to 2 so that the collectives are issued, the sum over one
    rank being the identity), then one engine backward with the overlapped replay."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ddp = importlib.import_module(PKG + ".ddp")
    red = ddp.GradReducer(average=False)
    red.world = 2                                   # issue the collectives although the group has one rank
    flat = torch.arange(1000, dtype=torch.float32, device=dev)
    red.reduce_async(flat[400:])
    red.finish(flat[:400])
    red.reduce(flat)
    torch.cuda.synchronize()
    ok = bool(torch.equal(flat.cpu(), torch.arange(1000, dtype=torch.float32))) and dist.get_backend() == "nccl"
    ref = _build(dev)
    want = _grads(ref, *_inputs(0, dev), 1)[0]
    model = _build(dev)
    ddp.attach(model, average=False)
    model._get_engine().ddp.world = 2
    got = _grads(model, *_inputs(0, dev), 3)       # step 1 records, steps 2-3 replay with RCCL all-reduces between segments
    errs = [float((g - want).abs().max()) / float(want.abs().max()) for g in got]
    q.put((ok and all(e < 2e-4 for e in errs), errs))
    dist.destroy_process_group()


def test_rccl_single_rank_on_hardware():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert res[0], res
