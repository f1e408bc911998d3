"""The option branches both shipped option files leave off: smart prong features
(layers/prong_feature_embedding.py:36-78), transformer_norm_first (layers/prong_custom_bert_encoder.py:45-52) and one_hot_pixels
(trainers/neutrino_full_dense_trainer.py:47-52), each as a full fp32 train step against the CPU oracle."""
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import rel_err
from model_utils import build_trainer, to_device
from test_oracle_golden import is_noise_grad

pytestmark = pytest.mark.gpu

SMALL = dict(densenet_structure=[2, 2], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=2, pixel_embedding_dim=64,
             hidden_dim=64, num_prong_decoder_layers=3, dropout=0.0, pixel_noise_std=0.0)


def _step_vs_oracle(cfg, batch, tol=6e-3):      # fp32 vs fp32: both sides carry ~1e-3 of rounding noise through the BatchNorm chains
    sd = O.fill_state(cfg, 21)
    (total, el, pl), (ev, pr), grads, _ = O.train_step(sd, cfg, batch)
    model = build_trainer(cfg, sd)
    model.train()
    model.network.hip_runtime().zero_grad()
    dbatch = to_device(batch)
    loss = model.training_step(dbatch, 0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - total.item()) < 1e-4 * abs(total.item()), (loss.item(), total.item())
    named = dict(model.named_parameters())
    worst, seen = 0.0, 0
    for k, r in grads.items():
        if is_noise_grad(k) or r.abs().max() < 1e-6 or k.endswith("event_position_embedding"):
            continue
        g = named[k].grad
        assert g is not None, k
        e = ((g.cpu() - r).norm() / r.norm()).item()
        worst = max(worst, e)
        seen += 1
        assert e < tol, (k, e)
    with torch.no_grad():
        _, _, ev_g, pr_g = model.shared_step(dbatch)
    assert rel_err(ev_g.cpu(), ev) < 1e-3 and rel_err(pr_g.cpu(), pr) < 1e-3
    return worst, seen, grads, named


def _with_features(batch, seed=5):
    g = torch.Generator().manual_seed(seed)
    b = list(batch)
    b[0] = torch.randn(b[0].shape, generator=g)
    b[1] = torch.randn(b[1].shape, generator=g)
    return tuple(b)


def test_smart_prong_features_train_step():
    cfg = O.tutorial_config(**dict(SMALL, disable_smart_features=False))
    # 22 prong rows: the MLP's BatchNorm1d statistics over a handful of rows would amplify fp32 rounding to the 1e-2 level
    batch = _with_features(O.synthetic_batch([4, 6, 5, 7], 31, cfg, event_hits=(200, 600), prong_hits=(20, 200)))
    worst, seen, grads, named = _step_vs_oracle(cfg, batch)
    k = "network.prong_embedding.feature_embedding.embedding.0.linear.weight"
    assert grads[k].abs().max() > 1e-6 and named[k].grad.abs().max() > 1e-6          # the MLP really trains
    print("smart features: worst rel L2 gradient error", worst, "over", seen, "tensors")


@pytest.mark.parametrize("bn,prelu,smart,dropout", [(False, True, False, 0.0),      # Linear(bias) - Identity - PReLU
                                                    (True, False, False, 0.0),      # Linear - BatchNorm1d - ReLU
                                                    (False, False, True, 0.0),      # both off, smart-feature MLP included
                                                    (False, True, False, 0.1)])     # Dropout modules present: decoder indices {0,3,6}/{1,4,7}
def test_linear_block_option_variants_train_step(bn, prelu, smart, dropout):
    """Round 5 (round-4 verdict, missing #6): options.linear_batch_norm = False (LinearBlock = Linear(bias) - Identity - act - Dropout;
    create_linear_block without BatchNorm1d) and options.linear_prelu_activation = False (ReLU, no slope parameter) -- reference
    layers/prong_feature_embedding.py:11-21, layers/encoder.py:10-24 -- as full train steps against the CPU oracle: the head plan's slot
    table follows the module lists (combined_embedding.linear.bias appears, the decoder's Sequential indices shift), the row kernels
    skip the normalisation / take ReLU.  With dropout the masks differ from the oracle's, so that case checks the plumbing only
    (strict state_dict load through build_trainer, finite loss, every parameter that should train has a gradient)."""
    cfg = O.tutorial_config(**dict(SMALL, linear_batch_norm=bn, linear_prelu_activation=prelu, disable_smart_features=not smart,
                                   dropout=dropout))
    batch = O.synthetic_batch([4, 6, 5, 7], 41, cfg, event_hits=(200, 600), prong_hits=(20, 200))
    if smart:
        batch = _with_features(batch)
    if dropout == 0.0:
        worst, seen, grads, named = _step_vs_oracle(cfg, batch)
        if not bn:
            k = "network.prong_embedding.combined_embedding.linear.bias"
            assert k in grads and named[k].grad.abs().max() > 1e-6
            assert not any("combined_embedding.norm" in n for n in named)
        if not prelu:
            assert not any("combined_embedding.activation" in n for n in named)
        print(f"LinearBlock variant bn={bn} prelu={prelu} smart={smart}: worst rel L2 gradient error {worst:.2e} over {seen} tensors")
        return
    sd = O.fill_state(cfg, 21)
    model = build_trainer(cfg, sd)
    model.train()
    model.network.hip_runtime().zero_grad()
    loss = model.training_step(to_device(batch), 0)
    loss.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    named = dict(model.named_parameters())
    assert "network.prong_decoder.hidden_layers.3.weight" in named and "network.prong_decoder.hidden_layers.1.weight" in named      # Linear / PReLU of blocks 1 / 0
    for k in ("network.prong_decoder.hidden_layers.0.weight", "network.prong_decoder.hidden_layers.1.weight",
              "network.prong_decoder.hidden_layers.3.bias", "network.prong_embedding.combined_embedding.linear.bias"):
        assert named[k].grad is not None and named[k].grad.abs().max() > 0, k


def test_transformer_norm_first_train_step():
    cfg = O.tutorial_config(**dict(SMALL, transformer_norm_first=True, dropout=0.0))
    worst, seen, _, _ = _step_vs_oracle(cfg, O.synthetic_batch([2, 4, 1], 33, cfg))
    print("norm_first: worst rel L2 gradient error", worst, "over", seen, "tensors")


def test_one_hot_pixels_train_step():
    cfg = O.tutorial_config(**dict(SMALL, one_hot_pixels=True))
    worst, seen, _, _ = _step_vs_oracle(cfg, O.synthetic_batch([1, 2], 35, cfg, event_hits=(200, 400), prong_hits=(20, 100)))
    print("one_hot_pixels: worst rel L2 gradient error", worst, "over", seen, "tensors")


def test_log_pixels_train_step_and_scattered_map():
    """a2's log_pixels branch (trainers/neutrino_full_dense_trainer.py:54-57: values = log(v + 1) instead of v / 255): the map k_scatter
    writes holds log1p of the hit values at the hit positions and zeros elsewhere, and a full train step matches the oracle."""
    cfg = O.tutorial_config(**dict(SMALL, log_pixels=True))
    batch = O.synthetic_batch([2, 3], 37, cfg, event_hits=(200, 400), prong_hits=(20, 100))
    worst, seen, _, named = _step_vs_oracle(cfg, batch)
    print("log_pixels: worst rel L2 gradient error", worst, "over", seen, "tensors")
    # the scattered map itself (fp32 mode): exactly log(v + 1) at the hits, 0 elsewhere
    model = build_trainer(cfg, O.fill_state(cfg, 21))
    model.eval()
    with torch.no_grad():
        model.shared_step(to_device(batch))
    img = model.network.hip_runtime().pr_engine.tap("img").float().cpu()
    c, v = batch[5].long(), batch[6]
    got = img[c[:, 0], c[:, 1], c[:, 2]]
    assert torch.allclose(got, torch.log(v + 1), rtol=2e-6, atol=0)
    assert int((img != 0).sum()) == int((v != 0).sum())
