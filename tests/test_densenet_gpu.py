"""HIP DenseNet embedder (through the C ABI) against the CPU oracle. Needs an MI355X."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg, rel_err

pytestmark = pytest.mark.gpu

PFX = "network.prong_embedding.prong_pixel_embedding"


def _engine(cfg, sd, mode=0, with_grad=False):
    from transformercvn.hip.engine import DenseNetEngine
    pix, feat, pos = O.embed_dims(cfg)
    eng = DenseNetEngine(cfg.pixel_dim, pix, cfg.initial_pixel_dim, cfg.densenet_growth_rate, cfg.densenet_batch_norm_size,
                         list(cfg.densenet_structure), cfg.pixel_shape[0], cfg.pixel_shape[1], cfg.dropout, mode)
    data = {k[len(PFX) + 1:]: v.cuda().contiguous() for k, v in sd.items() if k.startswith(PFX + ".") and v.is_floating_point()}
    grads = {k: torch.zeros_like(v) for k, v in data.items()} if with_grad else None
    eng.bind(data, grads)
    return eng, data, grads


def _oracle_densenet(cfg, sd, batch, training, dtype=torch.float32):
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    ctx = O._Ctx(training, 0.0)
    px = O.preprocess_pixels(cfg, batch[5], batch[6].to(dtype), False)
    out = O.densenet_forward(sd, PFX, cfg, px, ctx)
    return out, ctx


def _nchw(t):
    return t.permute(0, 3, 1, 2).float().cpu()


@pytest.mark.parametrize("name,training", [("small_b3", False), ("small_b3", True), ("tutorial_b2p4", False),
                                           ("tutorial_b2p4", True)])
def test_densenet_forward_fp32(name, training):
    cfg, over, batch, g = load_case(name)
    if training:
        cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    with torch.no_grad():
        ref, ctx = _oracle_densenet(cfg, sd, batch, training)
    eng, data, _ = _engine(cfg, sd)
    n_img = int(batch[7].sum())
    out = torch.empty(n_img, eng.out_dim, device="cuda")
    eng.forward(batch[5].cuda(), batch[6].cuda(), n_img, out, train=training, seed=1)
    torch.cuda.synchronize()
    errs = {}
    nb = len(cfg.densenet_structure)
    for tap, key in [("conv0", ":conv0"), ("dense1", ":dense1"), (f"dense{nb}", f":dense{nb}")]:
        errs[tap] = rel_err(_nchw(eng.tap(tap)), ctx.taps[PFX + key])
    errs["bottleneck1.0"] = rel_err(_nchw(eng.tap("bottleneck1.0")), ctx.taps[PFX + ":dense1.bottleneck0"])
    errs["condense"] = rel_err(eng.tap("condense").reshape(n_img, -1).cpu(), ctx.taps[PFX + ":condense"])
    errs["out"] = rel_err(out.cpu(), ref)
    print(name, training, errs)
    for k, v in errs.items():
        assert v < 2e-4, (k, v, errs)
    if training:
        for k in ("features.norm0.running_mean", "features.norm0.running_var", "output_block.norm.running_var",
                  "features.final_norm.running_mean"):
            assert rel_err(data[k].cpu(), ctx.new_running[PFX + "." + k]) < 1e-4, k
