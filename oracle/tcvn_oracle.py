"""CPU oracle for the TransformerCVN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

This is a plain-PyTorch fp32 (or fp64), functional restatement of the reference's
algorithm for the path named in BASELINE.json: sparse pixel maps -> DenseNet
embedders -> combined embedding -> transformer encoder -> decoders -> softmax focal
loss.  It works on a flat ``state_dict``-style mapping (reference key names) and a
config namespace, so that the very same tensors can be loaded into the reference, the
oracle and the HIP product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product package never does.

Parity pinning: ``oracle/make_golden.py`` imports the real reference (read-only, with
stub modules for absent third-party packages) in the build container, runs it on
seeded inputs/weights and commits the results under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors, and
``tests/test_oracle_vs_reference.py`` checks it against the live reference whenever
``/root/reference`` exists.

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
import zlib
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5


# ----------------------------------------------------------------------------------------------
# Configuration
# ----------------------------------------------------------------------------------------------
def tutorial_config(**overrides) -> SimpleNamespace:
    """Hyper-parameters of option_files/fdhd_beam_2018prod_aiml_tutorial_2025_04_21.json
    merged over the defaults of transformercvn/options.py:21-162 (only the keys the hot path reads)."""
    cfg = dict(
        hidden_dim=128, initial_feature_dim=8, initial_pixel_dim=64,
        feature_embedding_dim=32, pixel_embedding_dim=256, position_embedding_dim=32,
        num_embedding_layers=100, num_encoder_layers=6, num_prong_decoder_layers=4,
        num_attention_heads=8, transformer_activation="gelu", transformer_norm_first=False,
        linear_prelu_activation=True, linear_batch_norm=True,
        disable_smart_features=True, normalize_features=True, one_hot_pixels=False, log_pixels=False,
        densenet_structure=[3, 6, 12, 6, 3], densenet_growth_rate=32, densenet_batch_norm_size=4,
        pixel_noise_std=0.001, dropout=0.1, event_prong_loss_proportion=0.9, loss_gamma=1.0,
        # dataset-derived dimensions (neutrino_full_base_trainer.py:55-62)
        features_dim=4, extra_dim=2, pixel_dim=3, pixel_shape=(400, 280),
        num_prong_classes=8, num_event_classes=4,
    )
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


def make_divisible_channel_count(v: float, divisor: int, min_value: Optional[int] = None) -> int:
    """layers/prong_masked_mobilenet_embedding.py:10-23."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


# ----------------------------------------------------------------------------------------------
# state_dict layout (names + shapes) -- mirrors the module tree of the reference
# ----------------------------------------------------------------------------------------------
def _bn(prefix: str, c: int, out: Dict[str, Tuple[int, ...]]):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def densenet_layout(prefix: str, cfg, in_ch: int, out_dim: int, out: Dict[str, Tuple[int, ...]]):
    """layers/dense_net.py:97-162 (constructor order == state_dict order)."""
    g, bs = cfg.densenet_growth_rate, cfg.densenet_batch_norm_size
    c = cfg.initial_pixel_dim
    f = prefix + ".features"
    out[f + ".conv0.weight"] = (c, in_ch, 7, 7)
    out[f + ".conv0.bias"] = (c,)
    _bn(f + ".norm0", c, out)
    out[f + ".relu0.weight"] = (c,)
    nblocks = len(cfg.densenet_structure)
    for b, nl in enumerate(cfg.densenet_structure):
        for i in range(nl):
            cin = c + i * g
            p = f"{f}.dense{b + 1}.layers.{i}"
            _bn(p + ".bottleneck_block.norm1", cin, out)
            out[p + ".bottleneck_block.relu1.weight"] = (cin,)
            out[p + ".bottleneck_block.conv1.weight"] = (bs * g, cin, 1, 1)
            out[p + ".bottleneck_block.conv1.bias"] = (bs * g,)
            _bn(p + ".output_block.norm2", bs * g, out)
            out[p + ".output_block.relu2.weight"] = (bs * g,)
            out[p + ".output_block.conv2.weight"] = (g, bs * g, 3, 3)
            out[p + ".output_block.conv2.bias"] = (g,)
        c = c + nl * g
        if b != nblocks - 1:
            p = f"{f}.transition{b + 1}"
            _bn(p + ".norm", c, out)
            out[p + ".relu.weight"] = (c,)
            out[p + ".conv.weight"] = (c // 2, c, 1, 1)
            out[p + ".conv.bias"] = (c // 2,)
            c = c // 2
    _bn(f + ".final_norm", c, out)
    out[f + ".final_relu.weight"] = (c,)
    o = prefix + ".output_block"
    out[o + ".linear.weight"] = (out_dim, c)
    _bn(o + ".norm", out_dim, out)
    out[o + ".relu.weight"] = (out_dim,)
    return c


def _linear_block_layout(prefix: str, cfg, i: int, o: int, out):
    """layers/prong_feature_embedding.py:7-23."""
    out[prefix + ".linear.weight"] = (o, i)
    if not cfg.linear_batch_norm:
        out[prefix + ".linear.bias"] = (o,)
    if cfg.linear_batch_norm:
        _bn(prefix + ".norm", o, out)
    if cfg.linear_prelu_activation:
        out[prefix + ".activation.weight"] = (o,)


def feature_embedding_dims(cfg, out_dim: int) -> List[Tuple[int, int]]:
    """layers/prong_feature_embedding.py:56-70."""
    dims = [(cfg.features_dim + cfg.extra_dim, cfg.initial_feature_dim)]
    cur = cfg.initial_feature_dim
    for _ in range(cfg.num_embedding_layers):
        nxt = 2 * cur
        if nxt >= out_dim:
            break
        dims.append((cur, nxt))
        cur = nxt
    dims.append((cur, out_dim))
    return dims


def prong_decoder_dims(cfg) -> Tuple[List[Tuple[int, int]], int]:
    """layers/prong_target_decoder.py:19-32 (incl. the `return next_hidden_dim` quirk)."""
    cur = cfg.hidden_dim
    dims = []
    nxt = cur
    for _ in range(cfg.num_prong_decoder_layers):
        nxt = cur // 2
        if nxt < 8:
            break
        dims.append((cur, nxt))
        cur = nxt
    return dims, nxt


def embed_dims(cfg) -> Tuple[int, int, int]:
    """networks/neutrino_full_base_network.py:51-53."""
    return (make_divisible_channel_count(cfg.pixel_embedding_dim, 8),
            make_divisible_channel_count(cfg.feature_embedding_dim, 8),
            make_divisible_channel_count(cfg.position_embedding_dim, 8))


def state_layout(cfg) -> Dict[str, Tuple[int, ...]]:
    """Ordered name -> shape map of the reference Lightning module's state_dict
    (trainers/neutrino_base.py:37-41 + networks/neutrino_full_base_network.py:38-85,133-164)."""
    out: Dict[str, Tuple[int, ...]] = {}
    if cfg.normalize_features:
        out["mean"] = (cfg.features_dim,)
        out["std"] = (cfg.features_dim,)
        out["extra_mean"] = ()
        out["extra_std"] = ()
    pix, feat, pos = embed_dims(cfg)
    pe = "network.prong_embedding"
    out[pe + ".event_position_embedding"] = (1, pos)
    out[pe + ".prong_position_embedding"] = (1, pos)
    for j, (i, o) in enumerate(feature_embedding_dims(cfg, feat)):
        _linear_block_layout(f"{pe}.feature_embedding.embedding.{j}", cfg, i, o, out)
    in_ch = cfg.pixel_dim * 256 if cfg.one_hot_pixels else cfg.pixel_dim
    if getattr(cfg, "embedder", "dense") == "sdxl":              # networks/neutrino_full_sdxl_network.py:6-15 (parity unpinned)
        from oracle.sdxl_oracle import sdxl_layout
        sdxl_layout(pe + ".prong_pixel_embedding", in_ch, pix, cfg.initial_pixel_dim, out)
        sdxl_layout(pe + ".event_pixel_embedding", in_ch, pix + feat, cfg.initial_pixel_dim, out)
    else:
        densenet_layout(pe + ".prong_pixel_embedding", cfg, in_ch, pix, out)
        densenet_layout(pe + ".event_pixel_embedding", cfg, in_ch, pix + feat, out)
    _linear_block_layout(pe + ".combined_embedding", cfg, feat + pix + pos, cfg.hidden_dim, out)
    d = cfg.hidden_dim
    for l in range(cfg.num_encoder_layers):
        p = f"network.encoder.encoder.layers.{l}"
        out[p + ".self_attn.in_proj_weight"] = (3 * d, d)
        out[p + ".self_attn.in_proj_bias"] = (3 * d,)
        out[p + ".self_attn.out_proj.weight"] = (d, d)
        out[p + ".self_attn.out_proj.bias"] = (d,)
        out[p + ".linear1.weight"] = (d, d)      # dim_feedforward == hidden_dim (prong_custom_bert_encoder.py:45-52)
        out[p + ".linear1.bias"] = (d,)
        out[p + ".linear2.weight"] = (d, d)
        out[p + ".linear2.bias"] = (d,)
        out[p + ".norm1.weight"] = (d,)
        out[p + ".norm1.bias"] = (d,)
        out[p + ".norm2.weight"] = (d,)
        out[p + ".norm2.bias"] = (d,)
    out["network.event_decoder.hidden_layer.weight"] = (cfg.num_event_classes, d)
    out["network.event_decoder.hidden_layer.bias"] = (cfg.num_event_classes,)
    dims, final = prong_decoder_dims(cfg)
    idx = 0
    for (i, o) in dims:                           # layers/encoder.py:10-24
        p = "network.prong_decoder.hidden_layers"
        out[f"{p}.{idx}.weight"] = (o, i)
        out[f"{p}.{idx}.bias"] = (o,)
        idx += 1
        if cfg.linear_batch_norm:
            _bn(f"{p}.{idx}", o, out)
            idx += 1
        if cfg.linear_prelu_activation:
            out[f"{p}.{idx}.weight"] = (o,)
        idx += 1
        if cfg.dropout > 0.0:
            idx += 1
    out["network.prong_decoder.output_layer.weight"] = (cfg.num_prong_classes, final)
    out["network.prong_decoder.output_layer.bias"] = (cfg.num_prong_classes,)
    return out


# ----------------------------------------------------------------------------------------------
# Deterministic weight fill keyed by state_dict key name so that fixtures
# need not store 23 MB of weights.
# ----------------------------------------------------------------------------------------------
def tensor_kind(name: str, layout: Dict[str, Tuple[int, ...]]) -> str:
    """Classify a state_dict entry from its name and its siblings (PReLU owns only `weight`)."""
    leaf = name.rsplit(".", 1)[-1]
    base = name[:-(len(leaf) + 1)] if "." in name else ""
    shape = layout[name]
    if leaf == "num_batches_tracked":
        return "count"
    if name in ("std", "extra_std"):
        return "std"
    if name in ("mean", "extra_mean"):
        return "mean"
    if leaf in ("running_mean", "running_var"):
        return leaf
    if "position_embedding" in name:
        return "position"
    if len(shape) >= 2:
        return "matrix"
    if leaf == "weight":
        has_bias = (base + ".bias") in layout
        return "gamma" if has_bias else "prelu"
    return "bias"


def fill_tensor(name: str, shape: Tuple[int, ...], kind: str, seed: int = 0) -> Tensor:
    rng = np.random.Generator(np.random.Philox(key=(zlib.crc32(name.encode()) << 16) ^ seed))
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "count":
        return torch.tensor(3, dtype=torch.int64)
    if kind in ("std", "running_var"):
        v = 0.5 + rng.random(n)
    elif kind == "mean":
        v = 0.2 * rng.standard_normal(n)
    elif kind == "running_mean":
        v = 0.1 * rng.standard_normal(n)
    elif kind == "position":
        v = rng.standard_normal(n)
    elif kind == "matrix":                                    # conv / linear / in_proj weights
        v = rng.standard_normal(n) * math.sqrt(2.0 / int(np.prod(shape[1:])))
    elif kind == "prelu":
        v = 0.1 + 0.3 * rng.random(n)
    elif kind == "gamma":                                     # BN / LN scale
        v = 1.0 + 0.2 * (2 * rng.random(n) - 1)
    else:                                                     # any bias / beta
        v = 0.1 * rng.standard_normal(n)
    return torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape)).clone()


def fill_state(cfg, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Closed-form deterministic weights keyed by state_dict key name + seed."""
    layout = state_layout(cfg)
    sd = {}
    for k, shp in layout.items():
        t = fill_tensor(k, shp, tensor_kind(k, layout), seed)
        sd[k] = t if t.dtype == torch.int64 else t.to(dtype)
    return sd


# ----------------------------------------------------------------------------------------------
# Synthetic batches: unique coordinates per image, every image >= 1 hit.
# ----------------------------------------------------------------------------------------------
def synthetic_batch(prongs_per_event: List[int], seed: int, cfg=None, max_prongs: Optional[int] = None,
                    event_hits=(500, 4000), prong_hits=(20, 800)) -> Tuple[Tensor, ...]:
    """10-tuple in the order of dataset/minkowski_dataset.py:75-86 (MinkowskiCollection.__call__)."""
    cfg = cfg or tutorial_config()
    H, W = cfg.pixel_shape
    rng = np.random.Generator(np.random.Philox(key=seed))
    B = len(prongs_per_event)
    P = max_prongs or max(prongs_per_event)

    def images(n_img, lo, hi):
        coords, vals = [], []
        for i in range(n_img):
            nnz = int(rng.integers(lo, hi + 1))
            # clustered track-like hits: random walk segments keep the maps "mostly empty" but structured
            flat = rng.choice(H * W, size=nnz, replace=False)
            flat.sort()
            y, x = flat // W, flat % W
            coords.append(np.stack([np.full(nnz, i), y, x], 1))
            vals.append(rng.integers(1, 256, size=(nnz, 3)).astype(np.float32))
        return (torch.from_numpy(np.concatenate(coords).astype(np.int32)),
                torch.from_numpy(np.concatenate(vals)))

    event_coords, event_values = images(B, *event_hits)
    prong_coords, prong_values = images(int(sum(prongs_per_event)), *prong_hits)
    prong_mask = torch.zeros(B, P, dtype=torch.bool)
    for b, n in enumerate(prongs_per_event):
        prong_mask[b, :n] = True
    features = torch.from_numpy(rng.standard_normal((B, P, cfg.features_dim)).astype(np.float32))
    extra = torch.from_numpy(rng.standard_normal((B, cfg.extra_dim)).astype(np.float32))
    event_mask = torch.ones(B, 1, dtype=torch.bool)
    event_targets = torch.from_numpy(rng.integers(0, cfg.num_event_classes, size=B).astype(np.int64))
    pt = rng.integers(0, cfg.num_prong_classes, size=(B, P)).astype(np.int8)
    prong_targets = torch.from_numpy(pt)
    prong_targets[~prong_mask] = -1
    return (features, extra, event_coords, event_values, event_mask,
            prong_coords, prong_values, prong_mask, event_targets, prong_targets)


# ----------------------------------------------------------------------------------------------
# Forward path
# ----------------------------------------------------------------------------------------------
def sparse_to_dense(values: Tensor, coords: Tensor, image_size: Tuple[int, int]) -> Tensor:
    """trainers/neutrino_full_dense_trainer.py:15-24: batch = last image index + 1; non-accumulating
    indexed write; NHWC -> NCHW."""
    c = coords.T.long()
    n = int(c[0, -1].item()) + 1
    out = torch.zeros(n, *image_size, values.shape[1], dtype=values.dtype)
    out[c[0], c[1], c[2]] += values
    return out.permute(0, 3, 1, 2).contiguous()


def preprocess_pixels(cfg, coords: Tensor, values: Tensor, training: bool, noise: Optional[Tensor] = None) -> Tensor:
    """trainers/neutrino_full_dense_trainer.py:46-67.  `noise` (N(0,1) draw) may be injected for tests;
    with pixel_noise_std == 0 the training branch is the identity."""
    if cfg.one_hot_pixels:
        n, f = values.shape
        values = F.one_hot(values.long(), 256).reshape(n, 256 * f).to(values.dtype)
    else:
        values = torch.log(values + 1) if cfg.log_pixels else values / 255.0
        if training and cfg.pixel_noise_std != 0.0:
            if noise is None:
                noise = torch.randn_like(values)
            values = values * (1 + noise * cfg.pixel_noise_std)
    return sparse_to_dense(values, coords, tuple(cfg.pixel_shape))


class _Ctx:
    """Carries mode flags, collects intermediates and BN running-stat updates."""

    def __init__(self, training: bool, dropout: float = 0.0, mask_provider=None):
        self.training = training
        self.dropout = dropout
        # mask_provider(site: str, shape) -> keep-scale tensor (0 or 1/(1-p)) replaces torch's generator at every dropout
        # site, so that a GPU step's counter-based masks can be replayed here (tests/test_dropout_noise_gpu.py).  Sites:
        # "<densenet prefix>:dense<b>.<l>" [n,g,H,W], "<densenet prefix>:out" [n,out], "<linear block prefix>" [rows,D],
        # "<encoder layer prefix>:attn" [B*H,S,S], ":sa" / ":ffn_act" / ":ffn" [S,B,D], "decoder.<i>" [T*B,width].
        self.mask_provider = mask_provider
        # checkpoint=True: the stem and every dense layer / transition of the DenseNets run under
        # torch.utils.checkpoint (recomputed in backward: same arithmetic, same gradients), which keeps the
        # 288-map train step of BASELINE config 2 under ~15 GB instead of ~60 GB of saved activations.
        self.checkpoint = False
        self.taps: Dict[str, Tensor] = {}
        self.new_running: Dict[str, Tensor] = {}


def _batch_norm(sd, prefix: str, x: Tensor, ctx: _Ctx) -> Tensor:
    """torch BatchNorm{1,2}d: batch statistics (biased var) in training, running stats in eval;
    running_var is updated with the unbiased variance, momentum 0.1, eps 1e-5."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if ctx.training:
        n = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        with torch.no_grad():
            ctx.new_running[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            ctx.new_running[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * n / max(n - 1, 1)
    else:
        mean, var = rm, rv
    return (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)


def _prelu(x: Tensor, slope: Tensor) -> Tensor:
    shape = [1, -1] + [1] * (x.dim() - 2)
    s = slope.view(shape)
    return torch.where(x > 0, x, s * x)


def _dropout(x: Tensor, ctx: _Ctx, site: str = "") -> Tensor:
    if not (ctx.training and ctx.dropout > 0):
        return x
    if ctx.mask_provider is not None:
        return x * ctx.mask_provider(site, tuple(x.shape)).to(x.dtype)
    return F.dropout(x, ctx.dropout, True)


def _segment(ctx: _Ctx, fn, x: Tensor) -> Tensor:
    """Run one DenseNet segment, under activation checkpointing when ctx.checkpoint is set (memory only: the
    recomputation repeats the same CPU arithmetic; dropout sites replay the generator state / the mask provider)."""
    if ctx.checkpoint and ctx.training and torch.is_grad_enabled():
        from torch.utils.checkpoint import checkpoint
        return checkpoint(fn, x, use_reentrant=False)
    return fn(x)


def densenet_forward(sd, prefix: str, cfg, x: Tensor, ctx: _Ctx) -> Tensor:
    """layers/dense_net.py:8-45 (Bottleneck), 48-75 (DenseBlock), 78-94 (Transition), 97-167 (DenseNet)."""
    f = prefix + ".features"
    keep_taps = not ctx.checkpoint                               # full-resolution taps would defeat the checkpointing

    def stem(x):
        x = F.conv2d(x, sd[f + ".conv0.weight"], sd[f + ".conv0.bias"], stride=2, padding=3)
        if keep_taps:
            ctx.taps[prefix + ":conv0"] = x
        x = _prelu(_batch_norm(sd, f + ".norm0", x, ctx), sd[f + ".relu0.weight"])
        return F.avg_pool2d(x, kernel_size=3, stride=2)

    def dense_layer(p, tap):
        def run(x):
            y = _prelu(_batch_norm(sd, p + ".bottleneck_block.norm1", x, ctx), sd[p + ".bottleneck_block.relu1.weight"])
            y = F.conv2d(y, sd[p + ".bottleneck_block.conv1.weight"], sd[p + ".bottleneck_block.conv1.bias"])
            if tap is not None and keep_taps:
                ctx.taps[tap] = y
            y = _prelu(_batch_norm(sd, p + ".output_block.norm2", y, ctx), sd[p + ".output_block.relu2.weight"])
            y = F.conv2d(y, sd[p + ".output_block.conv2.weight"], sd[p + ".output_block.conv2.bias"], padding=1)
            return _dropout(y, ctx, tap_site[p])
        return run

    def transition(p):
        def run(x):
            x = _prelu(_batch_norm(sd, p + ".norm", x, ctx), sd[p + ".relu.weight"])
            x = F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])
            return F.avg_pool2d(x, kernel_size=2, stride=2)
        return run

    tap_site: Dict[str, str] = {}
    x = _segment(ctx, stem, x)
    ctx.taps[prefix + ":pool0"] = x
    nblocks = len(cfg.densenet_structure)
    for b, nl in enumerate(cfg.densenet_structure):
        for i in range(nl):
            p = f"{f}.dense{b + 1}.layers.{i}"
            tap_site[p] = f"{prefix}:dense{b + 1}.{i}"
            y = _segment(ctx, dense_layer(p, f"{prefix}:dense{b + 1}.bottleneck0" if i == 0 else None), x)
            x = torch.cat((x, y), dim=1)
        ctx.taps[f"{prefix}:dense{b + 1}"] = x
        if b != nblocks - 1:
            x = _segment(ctx, transition(f"{f}.transition{b + 1}"), x)
            ctx.taps[f"{prefix}:transition{b + 1}"] = x
    x = _prelu(_batch_norm(sd, f + ".final_norm", x, ctx), sd[f + ".final_relu.weight"])
    x = x.mean(dim=(2, 3))                                      # AdaptiveAvgPool2d((1,1)) + Flatten
    ctx.taps[prefix + ":condense"] = x
    o = prefix + ".output_block"
    x = F.linear(x, sd[o + ".linear.weight"])
    x = _prelu(_batch_norm(sd, o + ".norm", x, ctx), sd[o + ".relu.weight"])
    x = _dropout(x, ctx, prefix + ":out")
    ctx.taps[prefix + ":out"] = x
    return x


def linear_block(sd, prefix: str, cfg, x: Tensor, ctx: _Ctx) -> Tensor:
    """layers/prong_feature_embedding.py:25-33."""
    x = F.linear(x, sd[prefix + ".linear.weight"], sd.get(prefix + ".linear.bias"))
    if cfg.linear_batch_norm:
        x = _batch_norm(sd, prefix + ".norm", x, ctx)
    x = _prelu(x, sd[prefix + ".activation.weight"]) if cfg.linear_prelu_activation else F.relu(x)
    return _dropout(x, ctx, prefix)


def pack_indices(mask: Tensor) -> Tuple[Tensor, Tensor]:
    """layers/packed_data.py:59-66: I1 = event index, I2 = slot index of every true mask entry (row-major)."""
    B, L = mask.shape
    I1 = torch.arange(B).repeat_interleave(mask.sum(1))
    I2 = torch.masked_select(torch.arange(L).view(1, -1).repeat(B, 1), mask)
    return I1, I2


def prong_embedding_forward(sd, cfg, features, extra, event_pixels, event_mask, prong_pixels, prong_mask, ctx: _Ctx):
    """networks/neutrino_full_base_network.py:87-125."""
    pe = "network.prong_embedding"
    B, P, _ = features.shape
    pix, feat, pos = embed_dims(cfg)
    if getattr(cfg, "embedder", "dense") == "sdxl":
        from oracle.sdxl_oracle import sdxl_forward
        embed = lambda prefix, px: sdxl_forward(sd, prefix, px, ctx.taps)
    else:
        embed = lambda prefix, px: densenet_forward(sd, prefix, cfg, px, ctx)
    ev = embed(pe + ".event_pixel_embedding", event_pixels)
    ev = torch.cat((ev, sd[pe + ".event_position_embedding"].expand(B, -1)), dim=1)
    I1, I2 = pack_indices(prong_mask)
    packed = features[I1, I2]
    if cfg.disable_smart_features:                              # prong_feature_embedding.py:73-78
        fe = torch.zeros(packed.shape[0], feat, dtype=packed.dtype)
    else:
        fe = torch.cat([packed, extra[I1]], dim=1)
        for j in range(len(feature_embedding_dims(cfg, feat))):
            fe = linear_block(sd, f"{pe}.feature_embedding.embedding.{j}", cfg, fe, ctx)
    pp = embed(pe + ".prong_pixel_embedding", prong_pixels)
    # quirk: prongs also receive the *event* position embedding (neutrino_full_base_network.py:107)
    pr = torch.cat((fe, pp, sd[pe + ".event_position_embedding"].expand(pp.shape[0], -1)), dim=1)
    comb = linear_block(sd, pe + ".combined_embedding", cfg, torch.cat((ev, pr), dim=0), ctx)
    ctx.taps["combined"] = comb
    ev, pr = comb[:B], comb[B:]
    padded = torch.zeros(B, P, comb.shape[1], dtype=comb.dtype)
    padded[I1, I2] = pr                                         # packed_data.py:70-76
    tokens = torch.cat((ev.view(B, 1, -1), padded), dim=1)
    mask = torch.cat((event_mask, prong_mask), dim=1)
    ctx.taps["tokens"] = tokens
    return tokens, mask


def encoder_layer_forward(sd, p: str, cfg, x: Tensor, key_padding: Tensor, ctx: _Ctx) -> Tensor:
    """torch.nn.TransformerEncoderLayer (post-norm unless transformer_norm_first), seq-first [S,B,D],
    math attention path; as instantiated at layers/prong_custom_bert_encoder.py:45-54."""
    S, B, D = x.shape
    H = cfg.num_attention_heads
    hd = D // H
    act = F.gelu if cfg.transformer_activation == "gelu" else F.relu

    def sa(v):
        qkv = F.linear(v, sd[p + ".self_attn.in_proj_weight"], sd[p + ".self_attn.in_proj_bias"])
        q, k, vv = qkv.chunk(3, dim=-1)
        q = q.reshape(S, B * H, hd).transpose(0, 1)
        k = k.reshape(S, B * H, hd).transpose(0, 1)
        vv = vv.reshape(S, B * H, hd).transpose(0, 1)
        scores = torch.bmm(q * (1.0 / math.sqrt(hd)), k.transpose(1, 2))      # [B*H,S,S]
        bias = torch.zeros(B, 1, 1, S, dtype=x.dtype).masked_fill(key_padding.view(B, 1, 1, S), float("-inf"))
        scores = (scores.view(B, H, S, S) + bias).view(B * H, S, S)
        attn = _dropout(torch.softmax(scores, dim=-1), ctx, p + ":attn")
        o = torch.bmm(attn, vv).transpose(0, 1).reshape(S, B, D)
        o = F.linear(o, sd[p + ".self_attn.out_proj.weight"], sd[p + ".self_attn.out_proj.bias"])
        return _dropout(o, ctx, p + ":sa")

    def ff(v):
        h = _dropout(act(F.linear(v, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])), ctx, p + ":ffn_act")
        return _dropout(F.linear(h, sd[p + ".linear2.weight"], sd[p + ".linear2.bias"]), ctx, p + ":ffn")

    def ln(v, n):
        return F.layer_norm(v, (D,), sd[f"{p}.{n}.weight"], sd[f"{p}.{n}.bias"], LN_EPS)

    if cfg.transformer_norm_first:
        x = x + sa(ln(x, "norm1"))
        x = x + ff(ln(x, "norm2"))
    else:
        x = ln(x + sa(x), "norm1")
        x = ln(x + ff(x), "norm2")
    return x


def encoder_forward(sd, cfg, tokens: Tensor, mask: Tensor, ctx: _Ctx) -> Tensor:
    """layers/prong_custom_bert_encoder.py:57-75."""
    B, S, _ = tokens.shape
    seq_mask = mask.view(B, S, 1).transpose(0, 1).to(tokens.dtype)
    x = tokens.transpose(0, 1) * seq_mask
    for l in range(cfg.num_encoder_layers):
        x = encoder_layer_forward(sd, f"network.encoder.encoder.layers.{l}", cfg, x, ~mask, ctx)
    x = x * seq_mask
    ctx.taps["hidden"] = x
    return x


def decoders_forward(sd, cfg, hidden: Tensor, ctx: _Ctx) -> Tuple[Tensor, Tensor]:
    """layers/prong_decoder.py:15-16; layers/prong_target_decoder.py:34-41; networks/neutrino_full_base_network.py:186-188."""
    ev = F.linear(hidden[0], sd["network.event_decoder.hidden_layer.weight"], sd["network.event_decoder.hidden_layer.bias"])
    h = hidden[1:]
    T, B, D = h.shape
    h = h.reshape(T * B, D)
    dims, _ = prong_decoder_dims(cfg)
    idx = 0
    p = "network.prong_decoder.hidden_layers"
    for di, _ in enumerate(dims):
        h = F.linear(h, sd[f"{p}.{idx}.weight"], sd[f"{p}.{idx}.bias"])
        idx += 1
        if cfg.linear_batch_norm:
            h = _batch_norm(sd, f"{p}.{idx}", h, ctx)
            idx += 1
        h = _prelu(h, sd[f"{p}.{idx}.weight"]) if cfg.linear_prelu_activation else F.relu(h)
        idx += 1
        if cfg.dropout > 0.0:
            h = _dropout(h, ctx, f"decoder.{di}")
            idx += 1
    h = F.linear(h, sd["network.prong_decoder.output_layer.weight"], sd["network.prong_decoder.output_layer.bias"])
    return ev, h.reshape(T, B, -1).transpose(0, 1)


def forward(sd, cfg, batch8: Tuple[Tensor, ...], training: bool = False, apply_dropout: bool = False,
            noise: Optional[Tuple[Tensor, Tensor]] = None, mask_provider=None, checkpoint: bool = False):
    """trainers/neutrino_full_base_trainer.py:90-116 followed by networks/neutrino_full_base_network.py:166-188.
    Returns (event_logits [B,Ce], prong_logits [B,P,Cp], ctx)."""
    features, extra, event_coords, event_values, event_mask, prong_coords, prong_values, prong_mask = batch8
    dt = sd["network.event_decoder.hidden_layer.weight"].dtype
    ctx = _Ctx(training, cfg.dropout if apply_dropout else 0.0, mask_provider)
    ctx.checkpoint = checkpoint
    features = features.clone().to(dt)
    extra = extra.clone().to(dt)
    if cfg.normalize_features:
        features[prong_mask] = (features[prong_mask] - sd["mean"]) / sd["std"]
        extra = (extra - sd["extra_mean"]) / sd["extra_std"]
    ev_px = preprocess_pixels(cfg, event_coords, event_values.to(dt), training, None if noise is None else noise[0])
    pr_px = preprocess_pixels(cfg, prong_coords, prong_values.to(dt), training, None if noise is None else noise[1])
    tokens, mask = prong_embedding_forward(sd, cfg, features, extra, ev_px, event_mask, pr_px, prong_mask, ctx)
    hidden = encoder_forward(sd, cfg, tokens, mask, ctx)
    ev, pr = decoders_forward(sd, cfg, hidden, ctx)
    return ev, pr, ctx


def shared_step(sd, cfg, batch10, training: bool, apply_dropout: bool = False, noise=None, mask_provider=None,
                checkpoint: bool = False):
    """trainers/neutrino_full_base_trainer.py:118-146 (truncate to the max real prong count)."""
    features, extra, ec, evv, em, pc, pv, pm, et, pt = batch10
    mp = int(pm.sum(1).max())
    ev, pr, ctx = forward(sd, cfg, (features[:, :mp].contiguous(), extra, ec, evv, em, pc, pv, pm[:, :mp].contiguous()),
                          training, apply_dropout, noise, mask_provider, checkpoint)
    return et, pt[:, :mp].contiguous(), ev, pr, ctx


def focal_loss(logits: Tensor, targets: Tensor, gamma: float) -> Tensor:
    """trainers/neutrino_full_base_trainer.py:148-160."""
    if gamma == 0:
        return F.cross_entropy(logits, targets)
    logp = torch.log_softmax(logits, dim=-1).gather(1, targets.view(-1, 1)).squeeze(1)
    p = torch.softmax(logits, dim=-1).gather(1, targets.view(-1, 1)).squeeze(1)
    return (-logp * (1 - p) ** gamma).mean()


def training_loss(cfg, event_logits, prong_logits, event_targets, prong_targets):
    """trainers/neutrino_full_base_trainer.py:162-177. Returns (total, event_loss, prong_loss)."""
    el = focal_loss(event_logits, event_targets.long(), cfg.loss_gamma)
    valid = prong_targets >= 0
    pl = focal_loss(prong_logits[valid], prong_targets[valid].long(), cfg.loss_gamma)
    s = cfg.event_prong_loss_proportion
    return s * el + (1.0 - s) * pl, el, pl


def train_step(sd, cfg, batch10, apply_dropout: bool = False, noise=None, mask_provider=None,
               checkpoint: Optional[bool] = None):
    """One forward+backward of training_step; returns (losses, logits, grads dict, ctx).
    Gradients are taken w.r.t. every floating-point entry of `sd` that is not a BN running statistic
    or a normalisation constant.  `checkpoint` (default: on above 64 maps) recomputes the DenseNet segments in
    backward instead of keeping their activations (BASELINE config 2 = 288 maps: ~60 GB without)."""
    if checkpoint is None:
        n_maps = int(batch10[7].sum()) + batch10[7].shape[0]
        checkpoint = n_maps > 64
    skip = ("running_mean", "running_var", "num_batches_tracked")
    leaves = {}
    sd2 = {}
    for k, v in sd.items():
        if v.is_floating_point() and not k.endswith(skip) and k not in ("mean", "std", "extra_mean", "extra_std"):
            leaves[k] = v.detach().clone().requires_grad_(True)
            sd2[k] = leaves[k]
        else:
            sd2[k] = v
    et, pt, ev, pr, ctx = shared_step(sd2, cfg, batch10, True, apply_dropout, noise, mask_provider, checkpoint)
    total, el, pl = training_loss(cfg, ev, pr, et, pt)
    names = list(leaves)
    gs = torch.autograd.grad(total, [leaves[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in zip(names, gs)}
    return (total.detach(), el.detach(), pl.detach()), (ev.detach(), pr.detach()), grads, ctx
