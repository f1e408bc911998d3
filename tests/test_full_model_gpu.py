"""Drop-in Lightning module on the MI355X (HIP path through the C ABI) against the reference's golden vectors and the
fp64 CPU oracle.  Gate of BASELINE.json: event and prong logits within 1e-3 relative (fp32 mode)."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg, rel_err
from model_utils import build_trainer, to_device
from test_oracle_golden import grad_close, is_noise_grad

pytestmark = pytest.mark.gpu

LOGIT_GATE = 1e-3          # BASELINE.json north_star
# b2p8 / b2p12: hidden 128, 6-layer encoder, S = 9 / 13 -> the <16,5> and <16,8> row buckets of csrc/encoder_fused.hip (the ones
# BASELINE config 2 and the middle of config 5's ragged range run); ragged: S = 17 -> <16,11>; b2p4: S = 5 -> <16,3>
CASES = ["small_b3", "tutorial_b2p4", "tutorial_ragged", "tutorial_b2p8", "tutorial_b2p12"]      # (tutorial_b32p8: tests/test_fullsize_gpu.py)


def _loaded_so():
    with open("/proc/self/maps") as f:
        return any("libtcvn_hip.so" in line for line in f)


@pytest.mark.parametrize("name", CASES)
def test_eval_logits_match_reference(name):
    cfg, over, batch, g = load_case(name)
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])))
    model.eval()
    with torch.no_grad():
        et, pt, ev, pr = model.shared_step(to_device(batch))
    assert _loaded_so()
    e1, e2 = rel_err(ev.cpu(), g["eval_event_logits"]), rel_err(pr.cpu(), g["eval_prong_logits"])
    print(name, "eval logit rel err", e1, e2)
    assert e1 < LOGIT_GATE and e2 < LOGIT_GATE
    assert e1 < 5e-5 and e2 < 5e-5           # what fp32 actually achieves


@pytest.mark.parametrize("name", CASES)
def test_train_step_matches_reference(name):
    cfg, over, batch, g = load_case(name)
    cfgt = train_cfg(over)
    sd = O.fill_state(cfgt, int(g["weight_seed"]))
    model = build_trainer(cfgt, sd)
    model.train()
    rt = model.network.hip_runtime()
    rt.zero_grad()
    dbatch = to_device(batch)
    loss = model.training_step(dbatch, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["train_total_loss"])) < 1e-4 * abs(float(g["train_total_loss"]))
    assert abs(model.logged["event_loss"].item() - float(g["train_event_loss"])) < 2e-4
    assert abs(model.logged["prong_loss"].item() - float(g["train_prong_loss"])) < 2e-4
    # fp64 oracle gradients are the yardstick; the reference's own fp32 gradients sit 1e-3..1e-2 away from them
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    b64 = tuple(t.double() if t.is_floating_point() else t for t in batch)
    _, (ev64, pr64), g64, ctx64 = O.train_step(sd64, cfgt, b64)
    named = dict(model.named_parameters())
    deep = "tutorial" in name
    worst = 0.0
    for k, ref in g64.items():
        mine = named[k].grad
        assert mine is not None, k
        mine = mine.detach().cpu().double()
        if is_noise_grad(k) or ref.abs().max() < 1e-6:
            assert mine.abs().max().item() < 5e-3, k
            continue
        l2 = ((mine - ref).norm() / ref.norm()).item()
        worst = max(worst, l2)
        assert l2 < (1e-2 if deep else 2e-3), (k, l2)
    print(name, "worst relative L2 gradient error vs fp64 oracle", worst)
    # golden (reference fp32) sentinels with the fp32 noise band
    for k in [k for k in g if k.startswith("grad:")]:
        assert grad_close(k[5:], named[k[5:]].grad.cpu().numpy(), g[k], rtol=2e-2 if deep else 6e-3), k
    # BatchNorm running statistics after the step
    msd = model.state_dict()
    for k in [k for k in g if k.startswith("newstat:")]:
        if k.endswith("num_batches_tracked"):
            assert int(msd[k[8:]]) == int(g[k])
        else:
            assert rel_err(msd[k[8:]].cpu(), g[k]) < 1e-4, k
    # train-mode logits of the same step
    model.zero_grad()
    with torch.no_grad():
        et, pt, ev, pr = model.shared_step(dbatch)
    assert rel_err(ev.cpu(), g["train_event_logits"]) < LOGIT_GATE
    assert rel_err(pr.cpu(), g["train_prong_logits"]) < LOGIT_GATE


# The bf16 throughput mode is gated against the REFERENCE'S OWN bf16 behaviour on the same inputs and weights:
# tests/golden/autocast_bf16_band.npz (oracle/make_autocast_band.py) holds the logits of the real reference module run under
# torch.autocast(bfloat16) -- what `train.py -fp16` selects up to the half type -- on every golden case.  Its max-norm relative
# logit error against the fp32 goldens is 0.9-2.7e-2 in eval mode and 0.02-0.74 in train mode (train mode on these batches is
# degenerate for any 16-bit format: they hold 2-3 events, and BatchNorm1d over 2-3 rows maps the event embeddings to about -1/+1
# whatever their size, so roundings of the embeddings move the logits a lot).  Measured here on MI355X: eval 1.3e-3 (small net) ...
# 1.3e-2 (tutorial nets), train 0.02-0.34 -- inside the reference's band on every case.  Gates: eval <= 1.25 x the band and <= 2e-2
# absolute; train <= 2 x the band: on these 2-3-event batches the train-mode figure is one draw of a chaotic quantity, for the reference
# and for us -- a change of the conv0 statistics' summation ORDER (fp32 partial sums folded differently, last-bit differences) moved
# tutorial_b2p4's prong figure from 0.17 to 0.39 with the eval figures unchanged to three digits; the band is a scale, not a bound.
# The 32-event train step of test_fullsize_gpu.py (gate 1.25 x its band) is the meaningful train-mode bf16 check.
BF16_EVAL_GATE, BAND_SLACK, TRAIN_BAND_SLACK = 2e-2, 1.25, 2.0


def autocast_band(name):
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "autocast_bf16_band.npz"))
    return d, d[f"{name}:logit_err"]


@pytest.mark.parametrize("name", CASES)
def test_bf16_full_model_logit_error_vs_reference(name):
    """The throughput mode end to end (precision="bf16": bf16 DenseNets, fp32 token path) against the reference's fp32 golden
    logits, next to the error of the reference's own bf16 autocast run on the same case (dropout = noise = 0)."""
    cfg, over, batch, g = load_case(name)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    model = build_trainer(cfg, sd, precision="bf16")
    model.eval()
    dbatch = to_device(batch)
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(dbatch)
    e_ev, e_pr = rel_err(ev.cpu(), g["eval_event_logits"]), rel_err(pr.cpu(), g["eval_prong_logits"])
    cfgt = train_cfg(over)
    model = build_trainer(cfgt, O.fill_state(cfgt, int(g["weight_seed"])), precision="bf16")
    model.train()
    with torch.no_grad():
        _, _, ev, pr = model.shared_step(dbatch)
    t_ev, t_pr = rel_err(ev.cpu(), g["train_event_logits"]), rel_err(pr.cpu(), g["train_prong_logits"])
    _, band = autocast_band(name)
    print(f"BF16 LOGIT ERROR {name}: eval event {e_ev:.3e} prong {e_pr:.3e}; train event {t_ev:.3e} prong {t_pr:.3e}   "
          f"[reference under bf16 autocast: eval {band[0]:.3e} {band[1]:.3e}; train {band[2]:.3e} {band[3]:.3e}]")
    assert max(e_ev, e_pr) < BF16_EVAL_GATE and max(e_ev, e_pr) <= BAND_SLACK * max(band[0], band[1])
    assert max(t_ev, t_pr) <= TRAIN_BAND_SLACK * max(band[2], band[3])


def test_cpu_tensors_fail_loudly():
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, None, device=None)
    with pytest.raises(RuntimeError):
        model.shared_step(batch)


@pytest.mark.gpu
def test_device_feeder_feeds_training_steps():
    """batches staged by the feeder (pinned, copy stream, host-side prong counts) give the same loss as batches
    moved synchronously."""
    from transformercvn.hip.feeder import DeviceFeeder
    cfg, over, batch, g = load_case("small_b3")
    cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    batches = [O.synthetic_batch([2, 3, 1], 21 + i, cfg) for i in range(3)]
    model = build_trainer(cfg, sd)
    model.train()
    rt = model.network.hip_runtime()
    direct = []
    for b in batches:
        rt.zero_grad()
        direct.append(model.training_step(to_device(b), 0).item())
    model2 = build_trainer(cfg, sd)
    model2.train()
    rt2 = model2.network.hip_runtime()
    fed = []
    for b in DeviceFeeder(batches, "cuda", depth=2):
        assert len(b) == 11 and b[2].is_cuda and b[2].dtype == torch.int32
        rt2.zero_grad()
        fed.append(model2.training_step(b, 0).item())
    assert len(fed) == 3
    for a, b in zip(direct, fed):
        assert abs(a - b) <= 1e-5 * abs(a), (direct, fed)
