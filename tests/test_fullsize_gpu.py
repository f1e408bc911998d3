"""BASELINE config 2 at its own shape against a golden made by the REFERENCE at that shape: 32 distinct events x 8 prongs/event (256
prong maps + 32 event maps, S = 9 tokens per event), hidden 128, 6-layer encoder -- tests/golden/tutorial_b32p8.npz, written by
oracle/make_golden.py from the real reference module (3.4 minutes of CPU time).  One step here is exactly the step bench.py times
(minus dropout / noise, which tests/test_dropout_noise_gpu.py replays): eval logits, train loss and logits, the 26 sentinel gradients
and all 782 gradient norms are compared with the reference's own numbers.  With 32 distinct events no BatchNorm1d is degenerate, so
the bf16 mode's train-mode numbers are meaningful here (they are not on the 2-3 event goldens)."""
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


CASE = "tutorial_b32p8"     # 32 distinct events x 8 prongs: BASELINE config 2 itself (made from the reference by oracle/make_golden.py)
K = 1                       # (tile_batch(batch, k) can still replicate a golden batch k times; the real case needs no copies)


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
    if precision == "bf16":
        import os
        bl = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "autocast_bf16_band.npz"))[f"{CASE}:logit_err"]
        print(f"   reference under bf16 autocast on the same step: eval event {bl[0]:.3e} prong {bl[1]:.3e}")
        assert worst <= 1.25 * max(bl[0], bl[1])


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
    import os
    band = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "autocast_bf16_band.npz"))
    for k in [k for k in g if k.startswith("grad:")]:
        mine, r = named[k[5:]].grad.cpu().numpy(), g[k]
        if fp32:
            assert grad_close(k[5:], mine, r, rtol=5e-2), k
        elif np.abs(r).max() > 1e-6 and not is_noise_grad(k[5:]) and not k.endswith("event_position_embedding"):
            cos = float((mine.ravel() * r.ravel()).sum() / (np.linalg.norm(mine) * np.linalg.norm(r) + 1e-30))
            worst = max(worst, 1 - cos)
            # next to it: the cosine the reference's own bf16 autocast backward reaches with its fp32 gradients on this very step
            # (tests/golden/autocast_bf16_band.npz); gate = that band with 0.1 of slack, capped at 0.9
            ref_cos = float(band[f"{CASE}:gradcos:{k[5:]}"])
            print(f"   bf16 gradient cosine {k[5:][-56:]:56s} here {cos:.3f}   reference under autocast {ref_cos:.3f}")
            assert cos > min(0.9, ref_cos - 0.1), (k, cos, ref_cos)
            # ... and a magnitude band beside the direction (round-4 verdict, weak #2): a gradient with the right direction and a wrong
            # scale (a dropped 1/N, a doubled accumulation) passes every cosine
            ratio = float(np.linalg.norm(mine) / (np.linalg.norm(r) + 1e-30))
            print(f"   bf16 gradient norm ratio {k[5:][-56:]:55s} {ratio:.3f}")
            assert 0.7 < ratio < 1.4, (k, ratio)
    # all 782 gradient norms of the reference step
    if fp32:
        for k, n_ref in zip(g["grad_keys"], g["grad_norms"]):
            n_mine = named[str(k)].grad.norm().item()
            if n_ref > 1e-4 and not is_noise_grad(str(k)):
                assert abs(n_mine - n_ref) < 5e-2 * n_ref, (k, n_mine, n_ref)
    else:       # bf16: the same 782 norms as a band (printed worst cases; tensors whose reference norm is rounding noise are skipped)
        ratios = []
        for k, n_ref in zip(g["grad_keys"], g["grad_norms"]):
            if n_ref > 1e-3 and not is_noise_grad(str(k)) and not str(k).endswith("event_position_embedding"):
                ratios.append((named[str(k)].grad.norm().item() / n_ref, str(k)))
        ratios.sort()
        print(f"   bf16 gradient-norm ratios over {len(ratios)} tensors: min {ratios[0][0]:.3f} ({ratios[0][1][-60:]}), max {ratios[-1][0]:.3f} ({ratios[-1][1][-60:]})")
        assert 0.5 < ratios[0][0] and ratios[-1][0] < 2.0, (ratios[0], ratios[-1])
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(big)
    B = batch[0].shape[0]
    e = max(rel_err(ev[:B].cpu(), g["train_event_logits"]), rel_err(pr[-B:].cpu(), g["train_prong_logits"]))
    print(f"{precision}: full-size train-mode logit error vs golden {e:.3e}; worst 1-cos of sentinel grads {worst:.3e}")
    if fp32:
        assert e < 1e-3
    else:       # 32 distinct events: train-mode logits are meaningful in bf16 here -- gate against the reference's own autocast band
        bl = band[f"{CASE}:logit_err"]
        print(f"bf16 train-mode logit error {e:.3e}; reference under bf16 autocast: event {bl[2]:.3e} prong {bl[3]:.3e}")
        assert e <= 1.25 * max(bl[2], bl[3])
