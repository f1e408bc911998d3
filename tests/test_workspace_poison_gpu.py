"""No kernel may read workspace bytes it (or an earlier kernel of the same step) did not write: the workspaces come from
torch.empty, their offsets move with the ragged prong count, and since round 2 the gradient concat buffers G[b] are no longer
zeroed in the bf16 path (csrc/densenet_bwd.hip: the first contribution writes).  Every workspace is filled with 0xFF bytes -- NaN as
bf16, fp32 and fp64 -- before a step: a single poisoned operand would turn the loss or a gradient into NaN.  The forward has no
atomics, so the loss must equal the clean run's bit for bit; backward sums a few partials with LDS / fp32 atomics, so its gradients
are compared to summation-order noise."""
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg
from model_utils import build_trainer, to_device

pytestmark = pytest.mark.gpu


def _step(model, rt, dbatch, step_no):
    rt.step = step_no                       # same step number -> same seeds -> same dropout masks
    rt.zero_grad()
    loss = model.training_step(dbatch, 0)
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), rt.flat_grad.clone()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
@pytest.mark.parametrize("sliced", [False, True])
@pytest.mark.parametrize("case", ["tutorial_b2p4", "tutorial_ragged"])
def test_poisoned_workspace_gives_identical_step(precision, sliced, case):
    cfg, over, batch, g = load_case(case)           # 400x280 maps: odd 99x69 / 49x34 -> 24x17 pooling remainders; ragged: 1/16/5 prongs
    sd = O.fill_state(cfg, int(g["weight_seed"]))   # dropout 0.1 + noise: the path bench.py times
    model = build_trainer(cfg, sd, precision=precision)
    model.train()
    rt = model.network.hip_runtime()
    if sliced:                                      # block-by-block prong backward (the data-parallel schedule)
        rt.grad_ready_hook = lambda tag: None
    dbatch = to_device(batch)
    clean_loss, clean_grad = _step(model, rt, dbatch, 5)
    assert torch.isfinite(clean_grad).all()
    for eng in (rt.ev_engine, rt.pr_engine, rt.head):
        assert eng._ws is not None
        eng._ws.fill_(0xFF)
    loss, grad = _step(model, rt, dbatch, 5)
    assert torch.isfinite(grad).all(), "a kernel read workspace bytes nobody wrote"
    assert loss == clean_loss
    err = ((grad - clean_grad).norm() / clean_grad.norm()).item()
    print(f"{case} {precision} sliced={sliced}: gradient arena vs clean run, relative L2 {err:.2e}")
    assert err < 1e-4


def test_poisoned_workspace_eval_forward():
    cfg, over, batch, g = load_case("tutorial_ragged")
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), precision="bf16")
    model.eval()
    rt = model.network.hip_runtime()
    dbatch = to_device(batch)
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(dbatch)
        for eng in (rt.ev_engine, rt.pr_engine, rt.head):
            eng._ws.fill_(0xFF)
        _, _, ev2, pr2 = model.shared_step(dbatch)
    assert torch.equal(ev, ev2) and torch.equal(pr, pr2)
