"""world_size-2 gloo test of the gradient reducer (segments, overlap hooks, averaging, buffer broadcast)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "dune-transformercvn_amd"))
    from transformercvn.hip.distributed import GradReducer, broadcast_buffers, segment_plan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 1000
    spans = {"prong": (100, 400), "event": (400, 900)}
    plan = segment_plan(total, spans)
    assert plan["head"] == [(0, 100), (900, 1000)] and plan["event"] == [(400, 900)]
    g = torch.arange(total, dtype=torch.float32) * (rank + 1)
    red = GradReducer(g, spans)
    for tag in ("head", "event", "prong"):          # order in which backward completes the segments
        red.on_ready(tag)
    red.finish()
    expect = torch.arange(total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok = torch.allclose(g, expect)
    buf = torch.full((16,), float(rank))
    broadcast_buffers(buf)
    ok = ok and bool((buf == 0).all())
    ret[rank] = ok
    dist.destroy_process_group()


def test_grad_reducer_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
