"""SDXL-style embedder (BASELINE config 4; reference: layers/sdxl_net.py:7-42) on the MI355X against the CPU restatement in
oracle/sdxl_oracle.py.  PARITY UNPINNED: the arithmetic lives in an un-vendored, un-pinned `diffusers`, so these
tests establish oracle <-> HIP self-consistency plus the shape facts the reference fixes ([N,3,400,280] -> [N,out] through a
1x1 final map); they are not reference parity."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from oracle import sdxl_oracle as S
from golden_utils import rel_err
from model_utils import build_trainer, to_device

pytestmark = pytest.mark.gpu

PFX = "network.prong_embedding.prong_pixel_embedding"


def _cfg(**over):
    base = dict(embedder="sdxl", initial_pixel_dim=8, pixel_embedding_dim=64, hidden_dim=64, num_encoder_layers=2,
                num_prong_decoder_layers=3, dropout=0.0, pixel_noise_std=0.0)
    base.update(over)
    return O.tutorial_config(**base)


def _engine(cfg, sd, mode, with_grad):
    from transformercvn.hip.engine import SdxlEngine
    pix, feat, pos = O.embed_dims(cfg)
    eng = SdxlEngine(cfg.pixel_dim, pix, cfg.initial_pixel_dim, 2, 4, cfg.pixel_shape[0], cfg.pixel_shape[1], mode)
    data = {k[len(PFX) + 1:]: v.cuda().contiguous() for k, v in sd.items() if k.startswith(PFX + ".")}
    grads = {k: torch.zeros_like(v) for k, v in data.items()} if with_grad else None
    eng.bind(data, grads)
    return eng, data, grads


def _oracle(cfg, sd, batch, d_out=None, dtype=torch.float64):
    sdd = {k: v.to(dtype).requires_grad_(d_out is not None) for k, v in sd.items() if k.startswith(PFX + ".")}
    taps = {}
    px = O.preprocess_pixels(cfg, batch[5], batch[6].to(dtype), False)
    out = S.sdxl_forward(sdd, PFX, px, taps)
    grads = None
    if d_out is not None:
        gs = torch.autograd.grad(out, list(sdd.values()), d_out.to(dtype), allow_unused=True)
        grads = {k[len(PFX) + 1:]: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(sdd.items(), gs)}
    return out.detach(), {k: v.detach() for k, v in taps.items()}, grads


@pytest.mark.parametrize("mode,tol,gtol", [(0, 2e-5, 1e-4), (1, 3e-2, 8e-2)])      # measured: 1.1e-6 / 7.6e-6 and 1.3e-2 / 3.9e-2
def test_sdxl_embedder_forward_backward_vs_oracle(mode, tol, gtol):
    cfg = _cfg()
    sd = O.fill_state(cfg, 11)
    batch = O.synthetic_batch([2, 1], 5, cfg)
    n_img = int(batch[7].sum())
    d_out = torch.randn(n_img, 64, generator=torch.Generator().manual_seed(3))
    ref, taps, g_ref = _oracle(cfg, sd, batch, d_out)
    assert ref.shape == (n_img, 64)                                   # shape fact: [N,3,400,280] -> [N,out]
    assert tuple(taps[PFX + ":mid"].shape[2:]) == (1, 1)                # ... through a 1x1 final map
    eng, data, grads = _engine(cfg, sd, mode, True)
    out = torch.empty(n_img, 64, device="cuda")
    eng.forward(batch[5].cuda(), batch[6].cuda(), n_img, out, train=True, seed=1)
    errs = {}
    for tap in ("conv_in", "block0", "block1", "block4", "block8", "mid"):
        mine = eng.tap(tap).permute(0, 3, 1, 2).float().cpu()
        errs[tap] = ((mine.double() - taps[f"{PFX}:{tap}"]).norm() / taps[f"{PFX}:{tap}"].norm()).item()
    errs["out"] = ((out.cpu().double() - ref).norm() / ref.norm()).item()
    print("sdxl forward rel L2 errors, mode", mode, errs)
    assert max(errs.values()) < tol, errs
    eng.backward(d_out.cuda())
    torch.cuda.synchronize()
    worst, bad = 0.0, []
    for k, r in g_ref.items():
        mine = grads[k].cpu().double().reshape(r.shape)
        if "to_q" in k or "to_k" in k:                                 # one token: softmax == 1, no gradient
            assert mine.abs().max().item() == 0.0 and r.abs().max().item() < 1e-12
            continue
        e = ((mine - r).norm() / r.norm().clamp_min(1e-30)).item()
        worst = max(worst, e)
        if e > gtol:
            bad.append((k, e))
    print("sdxl backward worst rel L2 gradient error, mode", mode, worst, bad[:6])
    assert not bad, bad[:8]


def _run_wide(shape=(264, 280), n_maps=1):
    """Width-64 embedder on `n_maps` maps of `shape` (bf16): taps of the 64-channel stages and every gradient."""
    cfg = _cfg(initial_pixel_dim=64, pixel_embedding_dim=512, pixel_shape=tuple(shape))
    sd = O.fill_state(cfg, 13)
    batch = O.synthetic_batch([n_maps], 6, cfg)
    n_img = int(batch[7].sum())
    d_out = torch.randn(n_img, 512, generator=torch.Generator().manual_seed(4))
    eng, data, grads = _engine(cfg, sd, 1, True)
    out = torch.empty(n_img, 512, device="cuda")
    eng.forward(batch[5].cuda(), batch[6].cuda(), n_img, out, train=True, seed=1)
    taps = {t: eng.tap(t).float().cpu() for t in ("conv_in", "block0", "block1", "block2")}
    eng.backward(d_out.cuda())
    torch.cuda.synchronize()
    return cfg, sd, batch, d_out, out.cpu(), taps, {k: v.cpu() for k, v in grads.items()}


@pytest.mark.parametrize("shape", [(264, 280), (256, 304)])      # deep maps 33x35 .. 1x1 and 32x38 .. 1x1: different tile overhangs / packings
def test_sdxl_c64_tile_kernels_vs_generic_and_oracle(shape):
    """The 64 -> 64 halo-patch kernels (sdxl_conv3x3.hip: forward, data gradient, weight gradient) at the production width on maps
    that are not multiples of the 8 x 32 tile: against the generic implicit-GEMM kernels (TCVN_DISABLE_TILE=1 on the debug build,
    separate process; same bf16 products, other summation order) and against the fp64 oracle within the bf16 band of the
    small-width test above."""
    import os, subprocess, sys
    cfg, sd, batch, d_out, out, taps, grads = _run_wide(shape)
    from variant_utils import run_on_debug_build
    ref = run_on_debug_build(f"""
import test_sdxl_gpu as T
cfg, sd, batch, d_out, out, taps, grads = T._run_wide({tuple(shape)!r})
result = dict(out=out, taps=taps, grads=grads)
""", dict(TCVN_DISABLE_TILE="1"))
    e_taps = {k: ((taps[k] - ref["taps"][k]).norm() / ref["taps"][k].norm()).item() for k in taps}
    e_out = ((out - ref["out"]).norm() / ref["out"].norm()).item()
    e_grads = sorted(((((grads[k] - ref["grads"][k]).norm() / ref["grads"][k].norm().clamp_min(1e-30)).item(), k)
                      for k in grads if ref["grads"][k].abs().max() > 0), reverse=True)
    print("sdxl c64 tile vs generic: taps", e_taps, "out", e_out, "worst grads", e_grads[:4])
    assert max(e_taps.values()) < 1e-2 and e_out < 2e-2 and e_grads[0][0] < 5e-2
    o_ref, o_taps, g_ref = _oracle(cfg, sd, batch, d_out, dtype=torch.float32)
    e_o = ((out.double() - o_ref.double()).norm() / o_ref.double().norm()).item()
    e_b0 = ((taps["block0"].permute(0, 3, 1, 2).double() - o_taps[PFX + ":block0"].double()).norm() / o_taps[PFX + ":block0"].double().norm()).item()
    worst = max(((grads[k].double().reshape(r.shape) - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item()
                for k, r in g_ref.items() if "to_q" not in k and "to_k" not in k)
    worst_generic = max(((ref["grads"][k].double().reshape(r.shape) - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item()
                        for k, r in g_ref.items() if "to_q" not in k and "to_k" not in k)
    print("sdxl c64 vs fp32 oracle: out", e_o, "block0", e_b0, "worst grad", worst, "(generic kernels vs oracle:", worst_generic, ")")
    assert e_o < 3e-2 and e_b0 < 3e-2 and worst < 8e-2


def test_sdxl_full_model_train_step_vs_oracle():
    cfg = _cfg()
    sd = O.fill_state(cfg, 7)
    batch = O.synthetic_batch([2, 3, 1], 9, cfg)
    (total, el, pl), (ev, pr), grads, _ = O.train_step(sd, cfg, batch)
    model = build_trainer(cfg, sd)
    assert type(model).__name__ == "NeutrinoFullSDXLTrainer"
    model.train()
    model.network.hip_runtime().zero_grad()
    loss = model.training_step(to_device(batch), 0)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - total.item()) < 2e-4 * abs(total.item()), (loss.item(), total.item())
    named = dict(model.named_parameters())
    worst = 0.0
    for k, r in grads.items():
        if r.abs().max() < 1e-7:
            continue
        if k.endswith("event_position_embedding"):       # true gradient exactly 0 (a train-mode BatchNorm1d follows): rounding noise
            assert named[k].grad.abs().max().item() < 1e-4
            continue
        e = ((named[k].grad.cpu() - r).norm() / r.norm()).item()
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    print("sdxl full model: loss", loss.item(), total.item(), "worst rel L2 gradient error", worst)
    model = build_trainer(cfg, sd)                 # fresh module: the train step above moved the BatchNorm running statistics
    model.eval()
    with torch.no_grad():
        _, _, ev_g, pr_g = model.shared_step(to_device(batch))
    _, _, ev_o, pr_o, _ = O.shared_step(sd, cfg, batch, training=False)
    assert rel_err(ev_g.cpu(), ev_o) < 1e-3 and rel_err(pr_g.cpu(), pr_o) < 1e-3


@pytest.mark.parametrize("which", ["prong", "event"])
def test_sdxl_production_width_at_400x280_vs_oracle(which):
    """BASELINE config 4's own embedders on ONE 3x400x280 map each -- block widths [64,64,128,128,256,256,512,512,out], out = 256 (prong)
    / 288 (event: not a multiple of 64, so its last stage runs the general kernels) -- bf16 forward + backward against the fp32 CPU
    oracle (about 170 GFLOP of CPU work per map), inside the bf16 band of the small-width test.  Exercises every production tile
    kernel at the production map sizes 400x280 ... 1x1 (k_sconv3_c64, k_sconv3_g, the stride-2 kernels, packing of the small maps,
    the fused GroupNorm statistics).  Parity of this embedder stays unpinned (diffusers is not available)."""
    from transformercvn.hip.engine import SdxlEngine
    cfg = O.tutorial_config(embedder="sdxl", dropout=0.0, pixel_noise_std=0.0)          # tutorial widths: init 64, pixel embedding 256
    pfx = f"network.prong_embedding.{which}_pixel_embedding"
    pix, feat, pos = O.embed_dims(cfg)
    out_dim = pix if which == "prong" else pix + feat
    assert cfg.initial_pixel_dim == 64 and tuple(cfg.pixel_shape) == (400, 280) and out_dim == (256 if which == "prong" else 288)
    sd = {k: v for k, v in O.fill_state(cfg, 17).items() if k.startswith(pfx + ".")}
    batch = O.synthetic_batch([1], 8, cfg)
    coords, values = (batch[5], batch[6]) if which == "prong" else (batch[2], batch[3])
    d_out = torch.randn(1, out_dim, generator=torch.Generator().manual_seed(6))
    # fp32 oracle
    sdd = {k: v.float().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    ref = S.sdxl_forward(sdd, pfx, O.preprocess_pixels(cfg, coords, values.float(), False), taps)
    assert ref.shape == (1, out_dim) and tuple(taps[pfx + ":mid"].shape[2:]) == (1, 1)
    gs = torch.autograd.grad(ref, list(sdd.values()), d_out, allow_unused=True)
    g_ref = {k[len(pfx) + 1:]: (g if g is not None else torch.zeros_like(v)) for (k, v), g in zip(sdd.items(), gs)}
    # HIP, bf16
    eng = SdxlEngine(cfg.pixel_dim, out_dim, cfg.initial_pixel_dim, 2, 4, 400, 280, 1)
    data = {k[len(pfx) + 1:]: v.cuda().contiguous() for k, v in sd.items()}
    grads = {k: torch.zeros_like(v) for k, v in data.items()}
    eng.bind(data, grads)
    out = torch.empty(1, out_dim, device="cuda")
    eng.forward(coords.cuda(), values.cuda(), 1, out, train=True, seed=1)
    errs = {}
    for tap in ("conv_in", "block0", "block1", "block2", "block4", "block8", "mid"):
        mine = eng.tap(tap).permute(0, 3, 1, 2).float().cpu().double()
        r = taps[f"{pfx}:{tap}"].detach().double()
        errs[tap] = ((mine - r).norm() / r.norm()).item()
    errs["out"] = ((out.cpu().double() - ref.detach().double()).norm() / ref.detach().double().norm()).item()
    print(f"sdxl {which} embedder, production width at 400x280, bf16 vs fp32 oracle: forward rel L2", errs)
    assert max(errs.values()) < 3e-2, errs
    eng.backward(d_out.cuda())
    torch.cuda.synchronize()
    worst = []
    for k, r in g_ref.items():
        mine = grads[k].cpu().double().reshape(r.shape)
        if "to_q" in k or "to_k" in k:
            assert mine.abs().max().item() == 0.0
            continue
        worst.append((((mine - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item(), k))
    worst.sort(reverse=True)
    print("worst gradient rel L2:", worst[:4])
    assert worst[0][0] < 8e-2, worst[:6]


def _per_image_rel(mine, ref):
    """rel L2 error of every image of an [n, ...] tensor (a mixed-up per-image GroupNorm sum or a packed-tile pixel taken from the
    neighbouring image shows up in ONE image's figure and would be diluted in the batch norm)."""
    n = ref.shape[0]
    d = (mine.double() - ref.double()).reshape(n, -1).norm(dim=1)
    return (d / ref.double().reshape(n, -1).norm(dim=1).clamp_min(1e-30)).tolist()


def test_sdxl_production_width_five_maps_forward_400x280_vs_oracle():
    """the production-width kernels on SEVERAL maps.  `TileMap` (csrc/sdxl_conv3x3.hip) packs several images side by
    side / stacked into one 8x32 tile once the maps are <= 15 wide (block 5 onwards at 400x280: 12x8, 6x4, 3x2, 1x1), and the convolution
    epilogues add a tile's GroupNorm sums per image -- with one map neither the multi-image branch of the tile walk nor per-image sums
    inside a packed tile run.  Five distinct prong maps (config 4 runs 144), every tap and the output per image against the fp32 oracle."""
    from transformercvn.hip.engine import SdxlEngine
    cfg = O.tutorial_config(embedder="sdxl", dropout=0.0, pixel_noise_std=0.0)
    pfx = PFX
    pix, _, _ = O.embed_dims(cfg)
    n = 5
    sd = {k: v for k, v in O.fill_state(cfg, 19).items() if k.startswith(pfx + ".")}
    batch = O.synthetic_batch([n], 21, cfg)
    coords, values = batch[5], batch[6]
    taps = {}
    with torch.no_grad():
        ref = S.sdxl_forward({k: v.float() for k, v in sd.items()}, pfx, O.preprocess_pixels(cfg, coords, values.float(), False), taps)
    assert ref.shape == (n, pix) and (ref[0] - ref[1]).abs().max() > 1e-3          # distinct maps, distinct embeddings
    eng = SdxlEngine(cfg.pixel_dim, pix, cfg.initial_pixel_dim, 2, 4, 400, 280, 1)
    eng.bind({k[len(pfx) + 1:]: v.cuda().contiguous() for k, v in sd.items()}, None)
    out = torch.empty(n, pix, device="cuda")
    eng.forward(coords.cuda(), values.cuda(), n, out, train=False, seed=1)
    torch.cuda.synchronize()
    errs = {}
    for tap in ("conv_in", "block0", "block1", "block2", "block3", "block4", "block5", "block6", "block7", "block8", "mid"):
        mine = eng.tap(tap).permute(0, 3, 1, 2).float().cpu()
        errs[tap] = max(_per_image_rel(mine, taps[f"{pfx}:{tap}"]))
    errs["out"] = max(_per_image_rel(out.cpu(), ref))
    print("sdxl production width, 5 maps at 400x280, bf16 vs fp32 oracle: worst per-image rel L2", errs)
    assert max(errs.values()) < 3e-2, errs


def test_sdxl_production_width_five_maps_gradients_264x280_vs_oracle():
    """The same multi-map coverage for backward (data gradients through packed tiles, weight gradients summed over the images of a packed
    tile, GroupNorm backward sums per image): five distinct 264x280 maps (deep maps 33x35 ... 1x1; packed from 8x8 down), every gradient
    against the fp32 oracle inside the bf16 band of the one-map test, plus forward taps per image."""
    n = 5
    cfg, sd, batch, d_out, out, taps, grads = _run_wide((264, 280), n)
    o_ref, o_taps, g_ref = _oracle(cfg, sd, batch, d_out, dtype=torch.float32)
    e_out = max(_per_image_rel(out, o_ref))
    e_taps = {k: max(_per_image_rel(taps[k].permute(0, 3, 1, 2), o_taps[f"{PFX}:{k}"])) for k in taps}
    worst = sorted(((((grads[k].double().reshape(r.shape) - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item(), k)
                    for k, r in g_ref.items() if "to_q" not in k and "to_k" not in k), reverse=True)
    print("sdxl width 64..512, 5 maps at 264x280: out", e_out, "taps", e_taps, "worst gradients", worst[:4])
    assert e_out < 3e-2 and max(e_taps.values()) < 3e-2, (e_out, e_taps)
    assert worst[0][0] < 8e-2, worst[:6]
