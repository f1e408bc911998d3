"""The oracle against the LIVE reference (only where /root/reference exists, i.e. in the build container -- never on the GPU box):
a fresh seeded case that is not among the committed goldens, eval logits and one training step (loss + a few gradients).
The reference is imported read-only with the stub modules of oracle/make_golden.py."""
import os

import pytest
import torch

from oracle import tcvn_oracle as O

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "transformercvn")), reason="reference tree not present")


def test_oracle_matches_live_reference_on_a_fresh_case():
    from oracle import make_golden as MG
    over = dict(densenet_structure=[2, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=2,
                pixel_embedding_dim=64, hidden_dim=64, num_prong_decoder_layers=3, dropout=0.0, pixel_noise_std=0.0)
    cfg = O.tutorial_config(**over)
    batch = O.synthetic_batch([3, 1, 2], 4242, cfg)
    sd = O.fill_state(cfg, 77)
    torch.manual_seed(0)
    ref = MG.build_reference(cfg)
    res = ref.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # eval logits
    ref.eval()
    with torch.no_grad():
        _, _, ev_ref, pr_ref = ref.shared_step(batch)
    with torch.no_grad():
        ev, pr = O.shared_step(sd, cfg, batch, training=False)[2:4]
    assert (ev - ev_ref).abs().max() <= 2e-5 * ev_ref.abs().max().clamp_min(1.0)
    assert (pr - pr_ref).abs().max() <= 2e-5 * pr_ref.abs().max().clamp_min(1.0)
    # one training step
    ref.train()
    loss_ref = ref.training_step(batch, 0)
    loss_ref.backward()
    (total, el, pl), _, grads, _ = O.train_step(sd, cfg, batch)
    assert abs(total.item() - loss_ref.item()) <= 2e-5 * abs(loss_ref.item())
    named = dict(ref.named_parameters())
    for k in ("network.encoder.encoder.layers.0.self_attn.in_proj_weight", "network.prong_decoder.output_layer.weight",
              "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.output_block.conv2.weight"):
        g_ref = named[k].grad
        err = ((grads[k] - g_ref).norm() / g_ref.norm().clamp_min(1e-30)).item()
        assert err < 5e-3, (k, err)
