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


# ---------------------------------------------------------------------------------------------------------------------
# The runtime's own backward schedule + the Lightning-module hooks, end to end on two gloo ranks.  CPU tensors stand in for
# the gradient arena and fake engines write rank-dependent "gradients" into it; everything else is the product code path:
# NeutrinoFullBaseTrainer.enable_data_parallel() -> HipRuntime._backward() -> grad_ready_hook -> GradReducer -> on_after_backward().
# ---------------------------------------------------------------------------------------------------------------------
class _FakeEngine:
    def __init__(self, arena, span, value):
        self.arena, self.span, self.value = arena, span, value

    def backward(self, d):
        lo, hi = self.span
        self.arena[lo:hi] += self.value * torch.arange(hi - lo, dtype=torch.float32)


class _FakeHead(_FakeEngine):
    def __init__(self, arena, spans, value, rows, width):
        self.arena, self.spans, self.value, self.rows, self.width = arena, spans, value, rows, width

    def backward(self, rows, tok_row, d_ev, d_pr):
        for lo, hi in self.spans:
            self.arena[lo:hi] += self.value
        return torch.full((self.rows, self.width), self.value)


def _runtime_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "dune-transformercvn_amd"), os.path.join(root, "tests")]
    from oracle import tcvn_oracle as O
    from model_utils import build_trainer
    from transformercvn.hip.distributed import segment_plan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = O.tutorial_config(densenet_structure=[1, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=1)
    model = build_trainer(cfg, None, device=None)
    rt = model.network.hip_runtime()
    # the arenas ensure_bound() would build on the GPU, on the CPU
    names = [n for n, _ in model.network.named_parameters()]
    sizes = [p.numel() for _, p in model.network.named_parameters()]
    offs = dict(zip(names, zip([sum(sizes[:i]) for i in range(len(sizes))], sizes)))

    def span(prefix):
        ks = [k for k in offs if k.startswith(prefix)]
        return min(offs[k][0] for k in ks), max(offs[k][0] + offs[k][1] for k in ks)
    total = sum(sizes)
    rt.flat_grad = torch.zeros(total)
    torch.manual_seed(100 + rank)                           # every rank starts from DIFFERENT random weights (the reference's train.py
    rt.flat_param = torch.randn(total)                      # sets no seed and relies on DDP's construction-time broadcast)
    rt.flat_nbt = torch.full((7,), rank, dtype=torch.int64)
    rt.flat_buf = torch.full((64,), float(rank))
    rt.segments = {"prong": span("prong_embedding.prong_pixel_embedding."), "event": span("prong_embedding.event_pixel_embedding.")}
    rt._needs_rebind = lambda: False
    rt._reattach_grads = lambda: False
    rt._pos_grad = torch.zeros(1, 32)
    plan = segment_plan(total, rt.segments)
    pe = model.network.prong_embedding
    feat, pix = pe.feature_embedding_dim, pe.pixel_embedding_dim
    rt.head = _FakeHead(rt.flat_grad, plan["head"], float(rank + 1), 5, feat + pix + 32)
    rt.ev_engine = _FakeEngine(rt.flat_grad, rt.segments["event"], float(rank + 1))
    rt.pr_engine = _FakeEngine(rt.flat_grad, rt.segments["prong"], float(2 * rank + 1))
    order = []
    assert model.enable_data_parallel() is not None and rt.grad_ready_hook is not None

    def same_on_all_ranks(t):
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        return all(torch.equal(got[0], x) for x in got)
    # on_fit_start's state sync: rank 0's parameters, buffers and counters everywhere
    torch.manual_seed(100)
    synced = torch.equal(rt.flat_param, torch.randn(total)) and same_on_all_ranks(rt.flat_param)
    synced = synced and bool((rt.flat_nbt == 0).all()) and bool((rt.flat_buf == 0).all())
    inner = rt.grad_ready_hook
    rt.grad_ready_hook = lambda tag: (order.append(tag), inner(tag))
    model.on_train_batch_start(None, 0)                     # buffer broadcast from rank 0
    st = dict(B=2, feat=feat, pix=pix, rows=None, tok_row=None)
    rt._backward(st, torch.zeros(2, 4), torch.zeros(2, 3, 8))
    model.on_after_backward()                               # waits for the segment exchanges
    mean_scale = sum(r + 1 for r in range(world)) / world
    mean_scale_pr = sum(2 * r + 1 for r in range(world)) / world
    ok = synced and order == ["head", "event", "prong"] and bool((rt.flat_buf == 0).all())
    rt.flat_param -= 0.1 * rt.flat_grad                     # one optimizer step on the averaged gradients: the ranks stay identical
    ok = ok and same_on_all_ranks(rt.flat_param)
    for lo, hi in plan["head"]:
        ok = ok and torch.allclose(rt.flat_grad[lo:hi], torch.full((hi - lo,), mean_scale))
    lo, hi = rt.segments["event"]
    ok = ok and torch.allclose(rt.flat_grad[lo:hi], mean_scale * torch.arange(hi - lo, dtype=torch.float32))
    lo, hi = rt.segments["prong"]
    ok = ok and torch.allclose(rt.flat_grad[lo:hi], mean_scale_pr * torch.arange(hi - lo, dtype=torch.float32))
    # block-wise prong backward: one exchange segment per dense block, issued last block first
    class _Parts:
        n_parts = 2

        def __init__(self, arena, spans, value):
            self.arena, self.spans, self.value = arena, spans, value

        def backward_part(self, d, part):
            lo, hi = self.spans[part]
            self.arena[lo:hi] += self.value * (part + 1)
    plo, phi = rt.segments.pop("prong")
    mid = (plo + phi) // 2
    rt.segments["prong0"], rt.segments["prong1"] = (plo, mid), (mid, phi)
    rt.flat_grad.zero_()
    rt.pr_engine = _Parts(rt.flat_grad, [(plo, mid), (mid, phi)], float(rank + 1))
    assert model.enable_data_parallel() is not None
    order.clear()
    inner2 = rt.grad_ready_hook
    rt.grad_ready_hook = lambda tag: (order.append(tag), inner2(tag))
    rt._backward(st, torch.zeros(2, 4), torch.zeros(2, 3, 8))
    model.on_after_backward()
    ok = ok and order == ["head", "event", "prong1", "prong0"]
    ok = ok and torch.allclose(rt.flat_grad[plo:mid], torch.full((mid - plo,), mean_scale))
    ok = ok and torch.allclose(rt.flat_grad[mid:phi], torch.full((phi - mid,), 2 * mean_scale))
    # torch DDP accepts the module: everything but the hidden anchor is on the ignore list, checkpoints keep the reference keys
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    managed = [n for n, p in ddp.module.named_parameters() if n not in model._ddp_params_and_buffers_to_ignore]
    ok = ok and managed == ["_ddp_anchor"] and "_ddp_anchor" not in model.state_dict()
    ret[rank] = ok
    dist.destroy_process_group()


def test_runtime_backward_hooks_average_gradients_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_runtime_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)
