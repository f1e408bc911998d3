"""Run by tests/test_oracle_vs_reference.py in a fresh interpreter (the product package must not be importable here: the module
name `transformercvn` belongs to the reference in this process).  Prints one JSON line with the oracle-vs-reference errors."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:] = [p for p in sys.path if "dune-transformercvn_amd" not in p]
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from oracle import make_golden as MG  # noqa: E402
from oracle import tcvn_oracle as O  # noqa: E402

over = dict(densenet_structure=[2, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=2,
            pixel_embedding_dim=64, hidden_dim=64, num_prong_decoder_layers=3, dropout=0.0, pixel_noise_std=0.0)
cfg = O.tutorial_config(**over)
batch = O.synthetic_batch([3, 1, 2], 4242, cfg)
sd = O.fill_state(cfg, 77)
torch.manual_seed(0)
ref = MG.build_reference(cfg)
res = ref.load_state_dict(sd, strict=True)
assert not res.missing_keys and not res.unexpected_keys
ref.eval()
with torch.no_grad():
    _, _, ev_ref, pr_ref = ref.shared_step(batch)
    ev, pr = O.shared_step(sd, cfg, batch, training=False)[2:4]
out = {"event_logits": ((ev - ev_ref).abs().max() / ev_ref.abs().max().clamp_min(1.0)).item(),
       "prong_logits": ((pr - pr_ref).abs().max() / pr_ref.abs().max().clamp_min(1.0)).item()}
ref.train()
loss_ref = ref.training_step(batch, 0)
loss_ref.backward()
(total, el, pl), _, grads, _ = O.train_step(sd, cfg, batch)
out["loss"] = abs(total.item() - loss_ref.item()) / abs(loss_ref.item())
named = dict(ref.named_parameters())
for k in ("network.encoder.encoder.layers.0.self_attn.in_proj_weight", "network.prong_decoder.output_layer.weight",
          "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.output_block.conv2.weight"):
    g_ref = named[k].grad
    out["grad:" + k] = ((grads[k] - g_ref).norm() / g_ref.norm().clamp_min(1e-30)).item()
print("RESULT " + json.dumps(out))
