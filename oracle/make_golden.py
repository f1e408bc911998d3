"""Generate golden vectors from the REAL reference (run in the build container only).

    python oracle/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference, read-only) is imported unmodified; the third-party
packages it imports but that are absent here (MinkowskiEngine, h5py, numba,
pytorch_lightning, torchmetrics) are replaced by empty stub modules -- they are only
used for type annotations / the training harness, never for arithmetic on this path
(SURVEY.md 8(c), Appendix C).  Weights come from the closed-form generator
``tcvn_oracle.fill_state`` (keyed by state_dict key name), inputs from
``tcvn_oracle.synthetic_batch``; both are stored/reproducible, so fixtures stay small.

Nothing here travels to the GPU box except the resulting .npz data files.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tcvn_oracle as O  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

SENTINELS = [
    "network.prong_embedding.prong_pixel_embedding.features.conv0.weight",
    "network.prong_embedding.prong_pixel_embedding.features.conv0.bias",
    "network.prong_embedding.prong_pixel_embedding.features.norm0.weight",
    "network.prong_embedding.prong_pixel_embedding.features.relu0.weight",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.bottleneck_block.norm1.weight",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.bottleneck_block.norm1.bias",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.bottleneck_block.relu1.weight",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.bottleneck_block.conv1.weight",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.output_block.conv2.weight",
    "network.prong_embedding.prong_pixel_embedding.features.dense1.layers.0.output_block.conv2.bias",
    "network.prong_embedding.prong_pixel_embedding.features.transition1.norm.weight",
    "network.prong_embedding.prong_pixel_embedding.features.transition1.conv.weight",
    "network.prong_embedding.prong_pixel_embedding.features.final_norm.weight",
    "network.prong_embedding.prong_pixel_embedding.output_block.linear.weight",
    "network.prong_embedding.event_pixel_embedding.features.conv0.weight",
    "network.prong_embedding.event_pixel_embedding.features.dense1.layers.0.output_block.norm2.weight",
    "network.prong_embedding.event_pixel_embedding.output_block.norm.weight",
    "network.prong_embedding.event_position_embedding",
    "network.prong_embedding.combined_embedding.linear.weight",
    "network.prong_embedding.combined_embedding.activation.weight",
    "network.encoder.encoder.layers.0.self_attn.in_proj_weight",
    "network.encoder.encoder.layers.0.self_attn.out_proj.bias",
    "network.encoder.encoder.layers.0.norm1.weight",
    "network.encoder.encoder.layers.1.linear1.weight",
    "network.event_decoder.hidden_layer.weight",
    "network.prong_decoder.output_layer.weight",
]


def install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    stub("MinkowskiEngine", SparseTensor=type("SparseTensor", (), {}))
    stub("h5py")
    stub("numba")

    class _LM(nn.Module):
        def log(self, *a, **k):
            pass

    stub("pytorch_lightning", LightningModule=_LM)

    class _Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    stub("torchmetrics", Accuracy=_Metric, AUROC=_Metric)
    if REF not in sys.path:
        sys.path.insert(0, REF)


def build_reference(cfg):
    """Instantiate the reference Lightning module with a synthetic dataset description."""
    install_stubs()
    from transformercvn.options import Options
    from transformercvn.network.trainers.neutrino_full_dense_trainer import NeutrinoFullDenseTrainer

    class _DS:
        num_features, num_extra, pixel_features = cfg.features_dim, cfg.extra_dim, cfg.pixel_dim
        num_prong_classes, num_event_classes = cfg.num_prong_classes, cfg.num_event_classes
        pixel_shape, pixels = tuple(cfg.pixel_shape), None

        def __len__(self):
            return 1000

        def compute_statistics(self):
            return (torch.zeros(cfg.features_dim), torch.ones(cfg.features_dim),
                    torch.tensor(0.), torch.tensor(1.), None, None)

    class Ref(NeutrinoFullDenseTrainer):
        def create_datasets(self):
            d = _DS()
            return d, d, None

    opt = Options.load(os.path.join(REF, "option_files", "fdhd_beam_2018prod_aiml_tutorial_2025_04_21.json"))
    opt.update_options({k: v for k, v in vars(cfg).items()
                        if k in vars(opt) and k not in ("training_file",)})
    torch.manual_seed(0)
    return Ref(opt)


def tap_summary(t: torch.Tensor):
    f = t.detach().double().reshape(-1)
    step = max(1, f.numel() // 256)
    return np.array([f.mean().item(), f.std().item(), f.abs().max().item()]), f[::step][:256].float().numpy().copy()


def reference_taps(model, prefix_map):
    """Forward hooks on the reference modules matching the oracle's tap names."""
    taps, handles = {}, []
    for name, mod in model.named_modules():
        if name in prefix_map:
            key = prefix_map[name]
            handles.append(mod.register_forward_hook(lambda m, i, o, key=key: taps.__setitem__(key, o if torch.is_tensor(o) else o[0])))
    return taps, handles


def tap_names(cfg):
    m = {}
    pe = "network.prong_embedding"
    for emb in ("prong_pixel_embedding", "event_pixel_embedding"):
        p = f"{pe}.{emb}"
        m[p + ".features.conv0"] = p + ":conv0"
        m[p + ".features.pooling0"] = p + ":pool0"
        for b in range(len(cfg.densenet_structure)):
            m[f"{p}.features.dense{b + 1}"] = f"{p}:dense{b + 1}"
            m[f"{p}.features.dense{b + 1}.layers.0.bottleneck_block"] = f"{p}:dense{b + 1}.bottleneck0"
            if b != len(cfg.densenet_structure) - 1:
                m[f"{p}.features.transition{b + 1}"] = f"{p}:transition{b + 1}"
        m[p + ".condense"] = p + ":condense"
        m[p + ".output_block"] = p + ":out"
    m[pe + ".combined_embedding"] = "combined"
    m[pe] = "tokens"
    m["network.encoder"] = "hidden"
    return m


def run_case(name, cfg_over, prongs, batch_seed, weight_seed, with_train=True, max_prongs=None):
    cfg = O.tutorial_config(**cfg_over)
    batch = O.synthetic_batch(prongs, batch_seed, cfg, max_prongs=max_prongs)
    out = {
        "cfg_keys": np.array(list(cfg_over.keys())), "cfg_vals": np.array([repr(v) for v in cfg_over.values()]),
        "prongs": np.array(prongs), "batch_seed": batch_seed, "weight_seed": weight_seed,
        "features": batch[0].numpy(), "extra": batch[1].numpy(),
        "event_coords": batch[2].numpy().astype(np.int16), "event_values": batch[3].numpy().astype(np.uint8),
        "event_mask": batch[4].numpy(), "prong_coords": batch[5].numpy().astype(np.int16),
        "prong_values": batch[6].numpy().astype(np.uint8), "prong_mask": batch[7].numpy(),
        "event_targets": batch[8].numpy(), "prong_targets": batch[9].numpy(),
    }
    # ---- eval mode on the stock configuration (dropout modules present, inactive) ----
    sd = O.fill_state(cfg, weight_seed)
    ref = build_reference(cfg)
    missing = ref.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ref.eval()
    taps, handles = reference_taps(ref, tap_names(cfg))
    with torch.no_grad():
        et, pt, ev, pr = ref.shared_step(batch)
    for h in handles:
        h.remove()
    out["eval_event_logits"], out["eval_prong_logits"] = ev.numpy(), pr.numpy()
    for k, t in taps.items():
        s, smp = tap_summary(t)
        out["evaltap_stat:" + k], out["evaltap_samp:" + k] = s, smp
    print(f"[{name}] eval logits", ev.flatten()[:4].tolist())

    if with_train:
        # ---- training mode, RNG-free: dropout = 0, pixel noise = 0 (SURVEY.md 7 'RNG') ----
        over = dict(cfg_over, dropout=0.0, pixel_noise_std=0.0)
        cfgt = O.tutorial_config(**over)
        sdt = O.fill_state(cfgt, weight_seed)
        reft = build_reference(cfgt)
        reft.load_state_dict(sdt, strict=True)
        reft.train()
        taps, handles = reference_taps(reft, tap_names(cfgt))
        loss = reft.training_step(batch, 0)
        for h in handles:
            h.remove()
        # recompute the two loss terms exactly as training_step does
        with torch.no_grad():
            pass
        loss.backward()
        et, pt = batch[8], batch[9][:, :int(batch[7].sum(1).max())]
        out["train_total_loss"] = np.array(loss.item())
        for k, t in taps.items():
            s, smp = tap_summary(t)
            out["traintap_stat:" + k], out["traintap_samp:" + k] = s, smp
        named = dict(reft.named_parameters())
        norms, keys = [], []
        for k, p in named.items():
            if not p.requires_grad:
                continue
            keys.append(k)
            norms.append(0.0 if p.grad is None else p.grad.double().norm().item())
        out["grad_keys"], out["grad_norms"] = np.array(keys), np.array(norms)
        for k in SENTINELS:
            if k in named and named[k].grad is not None:
                out["grad:" + k] = named[k].grad.numpy().copy()
        new_sd = reft.state_dict()
        for k in ("network.prong_embedding.prong_pixel_embedding.features.norm0.running_mean",
                  "network.prong_embedding.prong_pixel_embedding.features.norm0.running_var",
                  "network.prong_embedding.prong_pixel_embedding.features.dense2.layers.3.bottleneck_block.norm1.running_var",
                  "network.prong_embedding.event_pixel_embedding.output_block.norm.running_mean",
                  "network.prong_embedding.combined_embedding.norm.running_var",
                  "network.prong_embedding.prong_pixel_embedding.features.norm0.num_batches_tracked"):
            if k in new_sd:
                out["newstat:" + k] = new_sd[k].numpy().copy()
        # train-mode logits + individual losses (second, identical forward: BN batch stats do not depend on running stats)
        reft.zero_grad()
        with torch.no_grad():
            et2, pt2, ev2, pr2 = reft.shared_step(batch)
            el = reft.loss(ev2, et2)
            valid = pt2 >= 0
            pl = reft.loss(pr2[valid], pt2[valid].long())
        out["train_event_logits"], out["train_prong_logits"] = ev2.numpy(), pr2.numpy()
        out["train_event_loss"], out["train_prong_loss"] = np.array(el.item()), np.array(pl.item())
        print(f"[{name}] train loss {loss.item():.6f} event {el.item():.6f} prong {pl.item():.6f}")
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.set_num_threads(8)
    only = sys.argv[1:]
    global run_case
    _run = run_case

    def run_case(name, *a, **k):                      # noqa: F811  (optional case filter from argv)
        if not only or name in only:
            _run(name, *a, **k)
    # BASELINE config 1: tutorial DenseNet, 2-layer encoder, B=2, 4 prongs/event
    run_case("tutorial_b2p4", dict(num_encoder_layers=2), [4, 4], batch_seed=11, weight_seed=1)
    # ragged batch, full 6-layer encoder (config 5 shape): 1..16 prongs
    run_case("tutorial_ragged", dict(), [1, 16, 5], batch_seed=12, weight_seed=2)
    # BASELINE config 2's own token shape: 8 prongs/event (S = 9: the <16,5> instantiation of the fused encoder), 6-layer encoder
    run_case("tutorial_b2p8", dict(), [8, 8], batch_seed=14, weight_seed=4)
    # 12 prongs/event (S = 13: the <16,8> instantiation; the middle of config 5's ragged range)
    run_case("tutorial_b2p12", dict(), [12, 12], batch_seed=15, weight_seed=5)
    # BASELINE config 2 itself: 32 DISTINCT events x 8 prongs (288 maps), full model -- minutes of reference CPU time, ~3 MB fixture
    run_case("tutorial_b32p8", dict(), [8] * 32, batch_seed=16, weight_seed=6)
    # reduced network for fast layer-by-layer debugging
    run_case("small_b3", dict(densenet_structure=[2, 2], densenet_growth_rate=8, initial_pixel_dim=16,
                              num_encoder_layers=2, pixel_embedding_dim=64, hidden_dim=64,
                              num_prong_decoder_layers=3),
             [2, 3, 1], batch_seed=13, weight_seed=3)


if __name__ == "__main__":
    main()
