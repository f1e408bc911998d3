"""BASELINE config 2 at its own shape -- 32 events x 8 prongs/event (256 prong maps + 32 event maps, S = 9 tokens per event) in one
step -- against the reference's golden vectors.

A batch made of k copies of a golden batch has, per copy, the golden's eval logits (events are independent in eval mode) and
-- because duplicating every sample leaves each BatchNorm's batch mean / biased variance unchanged -- also the golden's
train-mode logits, losses and parameter gradients (the loss is a mean over k times as many identical rows).  Unbiased
running variances differ by n/(n-1) and are not compared.  That pins the full-size forward AND backward launches (grids of
256-512 workgroups, 13.7 k tiles per 3x3 launch) to reference numbers without a CPU run at that size."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg, rel_err
from model_utils import build_trainer, to_device
from test_oracle_golden import grad_close, is_noise_grad

pytestmark = pytest.mark.gpu


def tile_batch(batch, k):
    f, x, ec, ev, em, pc, pv, pm, et, pt = batch
    B = f.shape[0]
    n_pr = int(pm.sum())

    def rep_coords(c, per):
        out = c.repeat(k, 1)
        out[:, 0] += (torch.arange(k).repeat_interleave(c.shape[0]) * per).to(out.dtype)
        return out
    return (f.repeat(k, 1, 1), x.repeat(k, 1), rep_coords(ec, B), ev.repeat(k, 1), em.repeat(k, 1), rep_coords(pc, n_pr),
            pv.repeat(k, 1), pm.repeat(k, 1), et.repeat(k), pt.repeat(k, 1))


CASE = "tutorial_b2p8"      # 2 events x 8 prongs, hidden 128, 6-layer encoder (made from the reference by oracle/make_golden.py)
K = 16                      # 16 copies -> exactly BASELINE config 2: 32 events, 256 prong maps, 8 prongs/event


@pytest.mark.parametrize("precision,gate", [("fp32", 1e-3), ("bf16", 2e-2)])
def test_config2_sized_eval_logits_equal_golden_per_copy(precision, gate):
    cfg, over, batch, g = load_case(CASE)
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), precision=precision)
    model.eval()
    big = tile_batch(batch, K)
    assert int(big[7].sum()) == 256 and big[0].shape[0] == 32 and int(big[7].sum(1).max()) == 8
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(to_device(big))
    ev, pr = ev.cpu(), pr.cpu()
    B = batch[0].shape[0]
    worst = 0.0
    for c in range(K):
        worst = max(worst, rel_err(ev[c * B:(c + 1) * B], g["eval_event_logits"]), rel_err(pr[c * B:(c + 1) * B], g["eval_prong_logits"]))
    print(f"{precision}: 32 events x 8 prongs (256 prong maps), worst per-copy eval logit error vs reference golden {worst:.3e}")
    assert worst < gate
    assert rel_err(ev[:B], ev[-B:]) < (1e-6 if precision == "fp32" else 1e-2)      # copies agree with each other


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_config2_sized_train_step_equals_golden(precision):
    cfg, over, batch, g = load_case(CASE)
    cfgt = train_cfg(over)
    sd = O.fill_state(cfgt, int(g["weight_seed"]))
    model = build_trainer(cfgt, sd, precision=precision)
    model.train()
    rt = model.network.hip_runtime()
    rt.zero_grad()
    big = to_device(tile_batch(batch, K))
    loss = model.training_step(big, 0)
    loss.backward()
    torch.cuda.synchronize()
    ref = float(g["train_total_loss"])
    fp32 = precision == "fp32"
    print(f"{precision}: full-size train loss {loss.item():.6f} (golden {ref:.6f})")
    assert abs(loss.item() - ref) < (1e-4 if fp32 else 3e-2) * abs(ref)
    named = dict(model.named_parameters())
    worst = 0.0
    for k in [k for k in g if k.startswith("grad:")]:
        mine, r = named[k[5:]].grad.cpu().numpy(), g[k]
        if fp32:
            assert grad_close(k[5:], mine, r, rtol=5e-2), k
        elif np.abs(r).max() > 1e-6 and "bias" not in k and "event_pixel_embedding" not in k:
            # (the golden batch holds TWO distinct event maps: BatchNorm1d over two values maps them to -1 / +1 whatever their
            #  size, so the event embedder's true gradient is a cancellation residue of order eps: noise in bf16)
            cos = float((mine.ravel() * r.ravel()).sum() / (np.linalg.norm(mine) * np.linalg.norm(r) + 1e-30))
            worst = max(worst, 1 - cos)
            assert cos > 0.9, (k, cos)
    # all 782 gradient norms of the reference step
    if fp32:
        for k, n_ref in zip(g["grad_keys"], g["grad_norms"]):
            n_mine = named[str(k)].grad.norm().item()
            if n_ref > 1e-4 and not is_noise_grad(str(k)):
                assert abs(n_mine - n_ref) < 5e-2 * n_ref, (k, n_mine, n_ref)
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(big)
    B = batch[0].shape[0]
    e = max(rel_err(ev[:B].cpu(), g["train_event_logits"]), rel_err(pr[-B:].cpu(), g["train_prong_logits"]))
    print(f"{precision}: full-size train-mode logit error vs golden {e:.3e}; worst 1-cos of sentinel grads {worst:.3e}")
    # bf16: the batch has 2 distinct events, so BatchNorm1d of the event rows is the degenerate -1/+1 case (see
    # test_full_model_gpu.py) -- the loss above is the bf16 gate, the logits are reported only
    assert e < 1e-3 or not fp32
