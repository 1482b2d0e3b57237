"""N > 1 path on CPU: two gloo ranks run the gradient reducer (the same code drives RCCL on the GPUs)."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ddp = importlib.import_module("small-object-detection-transformers_amd.ddp")
    flat = torch.full((1000,), float(rank + 1))
    ddp.GradReducer(average=False).reduce(flat)            # reference semantics: SUM (Train.py:439-440)
    ok1 = bool(torch.all(flat == 3.0))
    flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
    ddp.GradReducer(average=True).reduce(flat)
    ok2 = bool(torch.allclose(flat, torch.arange(10, dtype=torch.float32) * 1.5))
    # bucketed form: tail asynchronously, head at the end (the engine's overlap with the stage-1 backward)
    flat = torch.arange(100, dtype=torch.float32) * (rank + 1)
    red = ddp.GradReducer(average=True)
    red.reduce_async(flat[40:])
    flat[:40] += 1.0                      # "later kernels" still write the head
    red.finish(flat[:40])
    ok2 = ok2 and bool(torch.allclose(flat[40:], torch.arange(100, dtype=torch.float32)[40:] * 1.5)) \
        and bool(torch.allclose(flat[:40], torch.arange(40, dtype=torch.float32) * 1.5 + 1.0)) and not red._pending
    start, per = ddp.shard_batch(16, rank, world)
    ok3 = (start, per) == (rank * 8, 8)
    # attach(): parameters are broadcast from rank 0
    lin = torch.nn.Linear(4, 4)
    with torch.no_grad():
        lin.weight.fill_(float(rank))
    lin._get_engine = lambda: None
    ddp.attach(lin, average=True)
    ok4 = bool(torch.all(lin.weight == 0.0)) and lin._pending_ddp.world == 2
    # gradient accumulation (Train.py:125,448: 4 micro-steps per optimizer step), the engine's protocol: the kernels ADD
    # each backward's local gradient into the flat buffer, begin_backward() runs first, reduce() last
    def micro(red, flat, k, fresh):
        red.begin_backward(flat, fresh)
        flat += float((rank + 1) * k)                     # this rank's gradient of micro-step k
        red.reduce(flat)
    ok5 = True
    for average in (False, True):
        want = sum((r + 1) * k for r in range(world) for k in (1, 2, 3)) / (world if average else 1)
        # (a) reduce on every micro-step, as the reference's DDP does (no no_sync in Train.py)
        red, flat = ddp.GradReducer(average=average), torch.zeros(8)
        for k in (1, 2, 3):
            micro(red, flat, k, fresh=(k == 1))
        ok5 = ok5 and bool(torch.allclose(flat, torch.full((8,), want)))
        # (b) torch DDP's idiom: no_sync() around all but the last micro-step
        red, flat = ddp.GradReducer(average=average), torch.zeros(8)
        with red.no_sync():
            micro(red, flat, 1, fresh=True)
            micro(red, flat, 2, fresh=False)
        micro(red, flat, 3, fresh=False)
        ok5 = ok5 and bool(torch.allclose(flat, torch.full((8,), want)))
        # (c) mixed: reduced, then an unsynchronised step, then a synchronised one
        red, flat = ddp.GradReducer(average=average), torch.zeros(8)
        micro(red, flat, 1, fresh=True)
        with red.no_sync():
            micro(red, flat, 2, fresh=False)
        micro(red, flat, 3, fresh=False)
        ok5 = ok5 and bool(torch.allclose(flat, torch.full((8,), want)))
    ok6 = ddp.GradReducer().average is True               # torch DDP's mean is the default (Train.py keeps loss *= world_size)
    q.put((rank, ok1 and ok2 and ok3 and ok4 and ok5 and ok6))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
    assert res == {0: True, 1: True}
