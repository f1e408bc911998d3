"""Train-mode dropout and pixel noise of the path bench.py times (dropout 0.1, pixel_noise_std 0.001).

The reference draws its masks from torch's global generator (nn.Dropout inside layers/dense_net.py:29-40,
nn.TransformerEncoderLayer, create_linear_block; noise: trainers/neutrino_full_dense_trainer.py:58-60), so only the
distribution is comparable.  The kernels use stateless counter-based masks recomputed in backward.  Checked here:
  * the zeros of real kernel outputs (3x3 convolution slices of the concat buffers, the embedder output) are exactly the mask
    tcvn_dropout_keep() reports, kept values are scaled by 1/(1-p), keep rate = 1-p within 4 sigma, seeds decorrelate;
  * the noise factor recovered from the scattered pixel map is N(0,1)*std;
  * replaying those masks / that noise in the CPU oracle reproduces the GPU logits within the 1e-3 gate and the gradients
    within the fp32 band -- which can only hold if forward and backward apply the same masks at every site."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, rel_err
from model_utils import build_trainer, to_device
from test_oracle_golden import is_noise_grad

pytestmark = pytest.mark.gpu

EV = "network.prong_embedding.event_pixel_embedding"
PR = "network.prong_embedding.prong_pixel_embedding"


def keep(kind, p, seed, sid, rows, cols):
    from transformercvn.hip._lib import lib, check
    out = torch.empty(rows, cols, device="cuda")
    check(lib.tcvn_dropout_keep(kind, float(p), C.c_uint64(seed), C.c_uint32(sid), rows, cols, C.c_void_p(out.data_ptr()),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream)), "dropout_keep")
    return out


def _provider(cfg, rt, B, P):
    """site -> keep-scale tensor in the oracle's layout, from the kernels' own generators."""
    p = cfg.dropout
    g = cfg.densenet_growth_rate
    seeds = rt.last_seeds
    lp = "network.encoder.encoder.layers."

    def fn(site, shape):
        if ":dense" in site:                                      # [n, g, H, W]  <- [pixels, g]
            pref, rest = site.split(":dense")
            b, l = (int(v) for v in rest.split("."))
            n, _, H, W = shape
            m = keep(1, p, seeds["event" if pref == EV else "prong"], (b - 1) * 64 + l + 1, n * H * W, g)
            return m.view(n, H, W, g).permute(0, 3, 1, 2).cpu()
        if site.endswith(":out"):
            return keep(0, p, seeds["event" if site.startswith(EV) else "prong"], 0x4000, shape[0], shape[1]).cpu()
        if site.endswith("combined_embedding"):
            return keep(0, p, seeds["head"], 0x5000, shape[0], shape[1]).cpu()
        if site.startswith(lp):
            l = int(site[len(lp):].split(":")[0])
            kind = site.split(":")[1]
            sid = 0x6000 + 8 * l + {"attn": 0, "sa": 1, "ffn_act": 2, "ffn": 3}[kind]
            n = int(np.prod(shape))
            return keep(0, p, seeds["head"], sid, n // shape[-1], shape[-1]).view(shape).cpu()
        if site.startswith("decoder."):
            return keep(0, p, seeds["head"], 0x7000 + int(site.split(".")[1]), shape[0], shape[1]).cpu()
        raise KeyError(site)
    return fn


def _noise_from_map(eng, coords, values, std):
    """N(0,1) draws recovered from the scattered fp32 map: img = v/255 * (1 + z*std)."""
    img = eng.tap("img").float()
    c = coords.long()
    got = img[c[:, 0], c[:, 1], c[:, 2]].cpu()
    return (got / (values / 255.0) - 1.0) / std


def _replay_case(which):
    """small_b3: the committed golden's batch (hidden 64: layer-by-layer encoder kernels).  hidden128_p8: the same small DenseNets under
    the tutorial token path -- hidden 128, 6 layers, 8 prongs/event (S = 9) -- i.e. the FUSED encoder kernels (csrc/encoder_fused.hip,
    row bucket <16,5>: BASELINE config 2's instantiation) with their LDS dropout masks.  hidden128_p12: S = 13, bucket <16,8>.
    norm_first: the pre-norm branch (layer-by-layer kernels) with dropout > 0."""
    cfg, over, batch, g = load_case("small_b3")                  # dropout 0.1, pixel_noise_std 0.001 (tutorial values)
    if which == "small_b3":
        return cfg, batch, int(g["weight_seed"])
    over = dict(over)
    if which == "norm_first":
        over.update(transformer_norm_first=True)
        prongs = [2, 5, 3]
    else:
        over.update(hidden_dim=128, num_encoder_layers=6, num_prong_decoder_layers=4)
        prongs = [8, 8, 8] if which == "hidden128_p8" else [12, 9, 12, 11]
    cfg = O.tutorial_config(**over)
    return cfg, O.synthetic_batch(prongs, 31, cfg), 7


@pytest.mark.parametrize("which", ["small_b3", "hidden128_p8", "hidden128_p12", "norm_first"])
def test_dropout_masks_and_noise_replayed_in_oracle_fp32(which):
    cfg, batch, wseed = _replay_case(which)
    assert cfg.dropout == 0.1 and cfg.pixel_noise_std == 0.001
    sd = O.fill_state(cfg, wseed)
    model = build_trainer(cfg, sd)
    model.train()
    rt = model.network.hip_runtime()
    rt.zero_grad()
    dbatch = to_device(batch)
    loss = model.training_step(dbatch, 0)
    loss.backward()
    torch.cuda.synchronize()
    B, P = batch[7].shape[0], int(batch[7].sum(1).max())
    p = cfg.dropout
    # --- masks seen in real outputs == tcvn_dropout_keep, for every 3x3 site of both embedders ------------------------------
    total = kept = 0
    for eng, who in ((rt.ev_engine, "event"), (rt.pr_engine, "prong")):
        ch = cfg.initial_pixel_dim
        for b, nl in enumerate(cfg.densenet_structure):
            D = eng.tap(f"dense{b + 1}")
            n, H, W, _ = D.shape
            for l in range(nl):
                sl = D[..., ch + l * cfg.densenet_growth_rate: ch + (l + 1) * cfg.densenet_growth_rate].reshape(n * H * W, -1)
                m = keep(1, p, rt.last_seeds[who], b * 64 + l + 1, n * H * W, cfg.densenet_growth_rate)
                assert set(m.unique().tolist()) <= {0.0, float(np.float32(1.0) / np.float32(1.0 - p))}
                assert torch.equal(sl != 0, m != 0), (who, b, l)          # an un-dropped conv output is never exactly 0
                total += m.numel(); kept += int((m != 0).sum())
            ch = (ch + nl * cfg.densenet_growth_rate) // 2
    rate = kept / total
    sigma = (p * (1 - p) / total) ** 0.5
    print(f"3x3 keep rate {rate:.5f} over {total} elements (expect {1 - p}, sigma {sigma:.1e})")
    assert abs(rate - (1 - p)) < 4 * sigma
    # --- pixel noise ------------------------------------------------------------------------------------------------------
    z_ev = _noise_from_map(rt.ev_engine, batch[2], batch[3], cfg.pixel_noise_std)
    z_pr = _noise_from_map(rt.pr_engine, batch[5], batch[6], cfg.pixel_noise_std)
    z = torch.cat([z_ev.flatten(), z_pr.flatten()])
    print(f"pixel noise draws: n {z.numel()} mean {z.mean():.4f} std {z.std():.4f} |max| {z.abs().max():.2f}")
    assert abs(z.mean()) < 4 / z.numel() ** 0.5 and abs(z.std() - 1) < 0.05 and z.abs().max() < 6
    # --- replay in the oracle: logits (1e-3 gate), losses, gradients ---------------------------------------------------------
    (total_l, el, pl), (ev, pr), grads, ctx = O.train_step(sd, cfg, batch, apply_dropout=True, noise=(z_ev, z_pr),
                                                            mask_provider=_provider(cfg, rt, B, P))
    with torch.no_grad():
        model.network.hip_runtime().step -= 1                 # same step number -> same seeds -> same masks
        _, _, ev_g, pr_g = model.shared_step(dbatch)
    e1, e2 = rel_err(ev_g.cpu(), ev), rel_err(pr_g.cpu(), pr)
    print("train-mode logits with dropout+noise vs oracle replay:", e1, e2, "loss", loss.item(), total_l.item())
    assert e1 < 1e-3 and e2 < 1e-3
    assert abs(loss.item() - total_l.item()) < 1e-4 * abs(total_l.item())
    named = dict(model.named_parameters())
    worst = 0.0
    for k, ref in grads.items():
        mine = named[k].grad.detach().cpu()
        if is_noise_grad(k) or ref.abs().max() < 1e-6 or k.endswith("event_position_embedding"):
            continue            # (the position embedding feeds a train-mode BatchNorm1d only: exactly-zero true gradient, noise on both sides)
        l2 = ((mine - ref).norm() / ref.norm()).item()
        worst = max(worst, l2)
        assert l2 < 5e-3, (k, l2)           # a backward mask that differs from the forward mask gives O(1) errors
    print("worst relative L2 gradient error vs oracle replay", worst)


def test_dropout_scale_and_seed_decorrelation_bf16():
    """bf16 throughput kernels (the ones bench.py times) on the tutorial widths: first 3x3 layer of dense block 1 with and
    without dropout -- kept values are the undropped values times 1/(1-p), dropped ones are 0; another seed gives another mask."""
    from test_densenet_gpu import _engine
    cfg, over, batch, g = load_case("tutorial_b2p4")
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    n_img = int(batch[7].sum())
    coords, values = batch[5].cuda(), batch[6].cuda()
    p, gr, c0 = cfg.dropout, cfg.densenet_growth_rate, cfg.initial_pixel_dim

    def first_slice(drop, seed):
        eng, _, _ = _engine(O.tutorial_config(**dict(over, dropout=drop)), sd, mode=1)
        out = torch.empty(n_img, eng.out_dim, device="cuda")
        eng.forward(coords, values, n_img, out, train=True, seed=seed)
        torch.cuda.synchronize()
        return eng.tap("dense1")[..., c0:c0 + gr].float().reshape(-1, gr).clone(), out.clone()

    base, out0 = first_slice(0.0, 7)
    d1, out1 = first_slice(p, 7)
    d1b, _ = first_slice(p, 7)
    d2, _ = first_slice(p, 8)
    assert torch.equal(d1, d1b)                                   # stateless: same seed, same mask, same bits
    m1 = keep(1, p, 7, 1, base.shape[0], gr)
    assert torch.equal(d1 != 0, m1 != 0)
    k = d1 != 0
    ratio = d1[k] / base[k]
    print("kept/undropped ratio: mean", ratio.mean().item(), "min", ratio.min().item(), "max", ratio.max().item())
    assert (ratio - 1 / (1 - p)).abs().max() < 1.2e-2             # two bf16 roundings around the exact 1/(1-p)
    rate = k.float().mean().item()
    assert abs(rate - (1 - p)) < 4 * (p * (1 - p) / k.numel()) ** 0.5
    agree = ((d1 != 0) == (d2 != 0)).float().mean().item()        # independent masks agree on p^2 + (1-p)^2 = 0.82
    print("mask agreement between seeds 7 and 8:", agree)
    assert abs(agree - (p * p + (1 - p) ** 2)) < 0.01
    # the embedder output's own dropout (output block, stream 0x4000): zeros <=> mask
    mo = keep(0, p, 7, 0x4000, n_img, out1.shape[1])
    assert torch.equal(out1 != 0, mo != 0)


def test_token_path_dropout_sites_bf16_step_matches_statistics():
    """Token-path dropout sites (combined embedding, FFN activation, prong decoder) leave exact zeros in the head workspace's
    outputs only indirectly; their masks are replayed in the fp32 oracle test above.  Here: every site's mask has the right
    keep rate and the sites are pairwise decorrelated (distinct stream ids)."""
    p = 0.1
    sites = [0x5000, 0x6000, 0x6001, 0x6002, 0x6003, 0x6008, 0x7000, 0x7001]
    masks = [keep(0, p, 1234, s, 288, 128) != 0 for s in sites]
    for m in masks:
        assert abs(m.float().mean().item() - 0.9) < 4 * (0.09 / m.numel()) ** 0.5
    for i in range(len(masks)):
        for j in range(i + 1, len(masks)):
            agree = (masks[i] == masks[j]).float().mean().item()
            assert abs(agree - 0.82) < 0.02, (sites[i], sites[j], agree)
