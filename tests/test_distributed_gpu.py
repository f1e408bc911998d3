"""The RCCL branch of the data-parallel exchange on real hardware (reference: train.py:123-127, DDPStrategy over NCCL).

A one-GPU box cannot form a multi-rank RCCL group, but a ONE-rank "nccl" group runs the same code: ReduceOp.AVG all-reduces of arena
slices issued asynchronously from the backward's streams (token path on the main stream, event embedder on the side stream, the
prong embedder block by block), the buffer-arena broadcast and the state broadcast of enable_data_parallel().  At one rank every
collective is the identity, so the step must reproduce the step without the exchange.  Runs in a child process (its own RCCL
communicator, torn down with the process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch
import torch.distributed as dist
from oracle import tcvn_oracle as O
from golden_utils import load_case
from model_utils import build_trainer, to_device
from transformercvn.hip import distributed as hd

cfg, over, batch, g = load_case(CASE)
sd = O.fill_state(cfg, int(g["weight_seed"]))
dbatch = to_device(batch)

def run(exchange):
    model = build_trainer(cfg, sd, precision=PRECISION)
    model.train()
    rt = model.network.hip_runtime()
    order = []
    if exchange:
        red = model.enable_data_parallel()
        assert red is not None and red.avg is not None, "nccl backend must select ReduceOp.AVG"
        inner = rt.grad_ready_hook
        rt.grad_ready_hook = lambda tag: (order.append(tag), inner(tag))
    model.on_train_batch_start(dbatch, 0)
    rt.step = 3
    rt.zero_grad()
    loss = model.training_step(dbatch, 0)
    loss.backward()
    model.on_after_backward()
    torch.cuda.synchronize()
    return loss.item(), rt.flat_grad.clone(), rt.flat_buf.clone(), order

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
base = run(False)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
hd.SINGLE_RANK_COLLECTIVES = True
got = run(True)
dist.destroy_process_group()
n_blocks = len(cfg.densenet_structure)
assert got[3] == ["head", "event"] + [f"prong{i}" for i in range(n_blocks - 1, -1, -1)], got[3]
assert got[0] == base[0], (got[0], base[0])
err = ((got[1] - base[1]).norm() / base[1].norm()).item()
assert torch.isfinite(got[1]).all() and err < 1e-4, err
assert torch.equal(got[2], base[2])
print("RCCL one-rank exchange ok: segments", got[3], "gradient arena rel L2 vs no exchange", err)
"""


@pytest.mark.parametrize("case,precision", [("small_b3", "fp32"), ("tutorial_b2p4", "bf16")])
def test_rccl_exchange_runs_on_hardware_at_world_size_one(case, precision):
    code = (f"import sys\nsys.path[:0] = {sys.path!r}\nCASE, PRECISION = {case!r}, {precision!r}\n" + CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    print(p.stdout[-2000:], p.stderr[-3000:])
    assert p.returncode == 0
