"""SURVEY.md 8(f) row 3 -- TorchScript export (reference: CreateCompiled.ipynb cells 6-14): torch.jit.script of the drop-in
network compiles the ATen branch of every stage (the HIP branch is @torch.jit.unused), the scripted graph runs on the CPU and
reproduces the reference's golden eval logits, survives save / load, and the notebook's three wrapper modules script."""
import io

import numpy as np
import torch
from torch import nn

from oracle import tcvn_oracle as O
from golden_utils import load_case, rel_err
from model_utils import build_trainer


def _dense_inputs(cfg, batch):
    f, x, ec, ev, em, pc, pv, pm, et, pt = batch
    width = int(pm.sum(1).max())
    epx = O.preprocess_pixels(cfg, ec, ev, False)
    ppx = O.preprocess_pixels(cfg, pc, pv, False)
    return f[:, :width].contiguous(), x, epx, em, ppx, pm[:, :width].contiguous()


def test_scripted_network_matches_reference_goldens_on_cpu():
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), device=None)
    model.eval()
    scripted = torch.jit.script(model.network)
    with torch.no_grad():
        ev, pr = scripted(*_dense_inputs(cfg, batch))
    assert rel_err(ev, g["eval_event_logits"]) < 1e-4 and rel_err(pr, g["eval_prong_logits"]) < 1e-4
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf)
    with torch.no_grad():
        ev2, pr2 = loaded(*_dense_inputs(cfg, batch))
    assert torch.equal(ev, ev2) and torch.equal(pr, pr2)
    # stage by stage, as CreateCompiled.ipynb cell 7 calls them
    with torch.no_grad():
        tokens, mask = scripted.prong_embedding(*_dense_inputs(cfg, batch))
        hidden, _, _ = scripted.encoder(tokens, mask)
    assert tuple(hidden.shape) == (tokens.shape[1], tokens.shape[0], cfg.hidden_dim)


class _Simplified(nn.Module):
    """The structure of CreateCompiled.ipynb's DynamicSimplifedNetwork (cell 6): one event, every image a real prong."""
    __constants__ = ["pixel_features", "pixel_width", "pixel_height", "num_features", "num_extra"]

    def __init__(self, trainer):
        super().__init__()
        self.network = trainer.network
        self.num_features = trainer.training_dataset.num_features
        self.num_extra = trainer.training_dataset.num_extra
        self.pixel_features = trainer.training_dataset.pixel_features
        self.pixel_width, self.pixel_height = trainer.training_dataset.pixel_shape
        self.log_pixels = bool(trainer.options.log_pixels)

    def forward(self, pixels):
        pixels = torch.log(pixels.float() + 1) if self.log_pixels else pixels.float() / 255
        pixels = pixels.reshape(-1, self.pixel_features, self.pixel_width, self.pixel_height)
        num_images = pixels.shape[0]
        mask = torch.ones(num_images, device=pixels.device, dtype=torch.bool)
        features = torch.zeros(1, num_images - 1, self.num_features, device=pixels.device, dtype=pixels.dtype)
        extra = torch.zeros(1, self.num_extra, device=pixels.device, dtype=pixels.dtype)
        event, prongs = self.network(features, extra, pixels[:1], mask[:1].unsqueeze(0), pixels[1:], mask[1:].unsqueeze(0))
        return torch.softmax(event[0], 0), torch.softmax(prongs[0], 1)


def test_notebook_wrapper_scripts_and_keeps_the_export_shape_contract():
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), device=None)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)
    wrapper = torch.jit.script(_Simplified(model))
    pixels = torch.zeros(4, 3, 400, 280)
    pixels[:, :, 10:14, 20:25] = 37.0
    with torch.no_grad():
        ev, pr = wrapper(pixels)
    assert tuple(ev.shape) == (4,) and tuple(pr.shape) == (3, 8)          # [7,3,400,280] -> [4], [6,8]
    assert abs(ev.sum().item() - 1) < 1e-5 and torch.allclose(pr.sum(1), torch.ones(3), atol=1e-5)


def test_eager_cpu_call_still_fails_loudly():
    """The ATen branch exists for TorchScript export only: the eager module has no CPU fallback."""
    import pytest
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, None, device=None)
    with pytest.raises(RuntimeError):
        model.network(*_dense_inputs(cfg, batch))


def test_scripted_network_with_registered_operator_on_cpu():
    """the embedders as ONE dispatcher operator (tcvn::densenet_embed, transformercvn/hip/torch_ops.py).  On CPU
    tensors the scripted module dispatches to the operator's ATen kernel: reference goldens, save / load round trip, and the graph
    really holds the operator."""
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), device=None)
    model.eval()
    scripted = torch.jit.script(model.network.prepare_export(use_ops=True))
    assert "tcvn::densenet_embed" in str(scripted.prong_embedding.prong_pixel_embedding.graph)
    with torch.no_grad():
        ev, pr = scripted(*_dense_inputs(cfg, batch))
    assert rel_err(ev, g["eval_event_logits"]) < 1e-4 and rel_err(pr, g["eval_prong_logits"]) < 1e-4
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf)
    with torch.no_grad():
        ev2, pr2 = loaded(*_dense_inputs(cfg, batch))
    assert torch.equal(ev, ev2) and torch.equal(pr, pr2)
    # the operator is callable directly, and the default export is still the pure ATen graph
    emb = model.network.prong_embedding.prong_pixel_embedding
    px = O.preprocess_pixels(cfg, batch[5], batch[6], False)
    with torch.no_grad():
        direct = torch.ops.tcvn.densenet_embed(px, emb.op_tensors, emb.op_cfg)
    assert direct.shape == (int(batch[7].sum()), O.embed_dims(cfg)[0])
    plain = torch.jit.script(model.network.prepare_export(use_ops=False))
    assert "tcvn::" not in str(plain.prong_embedding.prong_pixel_embedding.graph)
