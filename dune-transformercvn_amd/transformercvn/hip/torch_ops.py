"""Dispatcher registration of the pixel-map embedder (reference: CreateCompiled.ipynb cells 6-14).

``torch.jit.script(network)`` compiles, per stage, an ATen branch -- that export loads anywhere, also in a C++ process without this
package (the reference's LArSoft use).  To let a scripted module reach the gfx950 kernels as well, the embedder is registered in the
PyTorch dispatcher -- the Python face of ``TORCH_LIBRARY`` / ``TORCH_LIBRARY_IMPL`` -- as

    tcvn::densenet_embed(Tensor image, Tensor[] tensors, int[] cfg) -> Tensor        [N, C, H, W] -> [N, out]   (eval mode)

with two kernels: ``CPU`` = the reference arithmetic through ATen, ``CUDA`` (= HIP on ROCm) = ``tcvn_densenet_forward`` of
libtcvn_hip.so behind the plan cache below.  ``tensors`` are the embedder's parameters and floating-point buffers in the plan's slot
order (``DenseNetEngine.slots()``), ``cfg`` = [in_ch, out_dim, init_ch, growth, bn_size, precision, layers...].  A network prepared
with ``prepare_export(use_ops=True)`` scripts its embedders to this operator; the scripted module then dispatches per device: CPU
tensors run ATen, GPU tensors run the HIP kernels.  A process that loads such a file imports ``transformercvn`` first (that registers
the operator); the pure-ATen export needs nothing.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

_LIB = torch.library.Library("tcvn", "DEF")
_LIB.define("densenet_embed(Tensor image, Tensor[] tensors, int[] cfg) -> Tensor")

_EPS = 1e-5
_slot_names: Dict[Tuple[int, ...], List[str]] = {}
_engines: Dict[Tuple, object] = {}


def _names(cfg: List[int]) -> List[str]:
    """slot names (parameters + floating-point buffers, plan order) of the embedder described by cfg"""
    key = tuple(cfg[:5]) + tuple(cfg[6:])
    if key not in _slot_names:
        from .engine import DenseNetEngine
        from . import _lib
        eng = DenseNetEngine(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], list(cfg[6:]), 400, 280, 0.0, 0)      # the slot table does not depend on H, W
        _slot_names[key] = [n for n, _, kind in eng.slots() if kind != _lib.SLOT_COUNTER]
    return _slot_names[key]


def embedder_tensors(module) -> Tuple[List[Tensor], List[int]]:
    """(tensors, cfg) of a transformercvn DenseNet holder module for ``tcvn::densenet_embed``"""
    h = module.hyper
    cfg = [h["in_ch"], h["out_dim"], h["init_ch"], h["growth"], h["bn_size"], int(getattr(module, "hip_mode", 0))] + list(h["layers"])
    named = module.hip_tensors()
    return [named[n] for n in _names(cfg)], cfg


def _bn_prelu(x: Tensor, t: Dict[str, Tensor], norm: str, relu: str) -> Tensor:
    y = F.batch_norm(x, t[norm + ".running_mean"], t[norm + ".running_var"], t[norm + ".weight"], t[norm + ".bias"], False, 0.1, _EPS)
    return F.prelu(y, t[relu + ".weight"])


def _densenet_cpu(image: Tensor, tensors: List[Tensor], cfg: List[int]) -> Tensor:
    """layers/dense_net.py:97-167 of the reference in eval mode, through ATen"""
    t = dict(zip(_names(cfg), tensors))
    layers = list(cfg[6:])
    x = F.conv2d(image.float(), t["features.conv0.weight"], t["features.conv0.bias"], stride=2, padding=3)
    x = F.avg_pool2d(_bn_prelu(x, t, "features.norm0", "features.relu0"), 3, 2)
    for b, nl in enumerate(layers):
        for l in range(nl):
            p = f"features.dense{b + 1}.layers.{l}."
            y = _bn_prelu(x, t, p + "bottleneck_block.norm1", p + "bottleneck_block.relu1")
            y = F.conv2d(y, t[p + "bottleneck_block.conv1.weight"], t[p + "bottleneck_block.conv1.bias"])
            y = _bn_prelu(y, t, p + "output_block.norm2", p + "output_block.relu2")
            y = F.conv2d(y, t[p + "output_block.conv2.weight"], t[p + "output_block.conv2.bias"], padding=1)
            x = torch.cat((x, y), dim=1)
        if b + 1 != len(layers):
            p = f"features.transition{b + 1}."
            x = F.conv2d(_bn_prelu(x, t, p + "norm", p + "relu"), t[p + "conv.weight"], t[p + "conv.bias"])
            x = F.avg_pool2d(x, 2, 2)
    x = _bn_prelu(x, t, "features.final_norm", "features.final_relu").mean(dim=(2, 3))
    x = F.linear(x, t["output_block.linear.weight"])
    x = F.batch_norm(x, t["output_block.norm.running_mean"], t["output_block.norm.running_var"], t["output_block.norm.weight"],
                     t["output_block.norm.bias"], False, 0.1, _EPS)
    return F.prelu(x, t["output_block.relu.weight"])


def _densenet_hip(image: Tensor, tensors: List[Tensor], cfg: List[int]) -> Tensor:
    """tcvn_densenet_forward (csrc/densenet.hip) in eval mode; the plan is cached per (shape, configuration, parameter storage)"""
    from .engine import DenseNetEngine
    from .pixels import SparsePixels
    n, c, H, W = image.shape
    key = (tuple(cfg), H, W, image.device.index, tuple(x.data_ptr() for x in tensors))
    eng = _engines.get(key)
    if eng is None:
        if len(_engines) > 16:
            _engines.clear()
        eng = DenseNetEngine(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], list(cfg[6:]), H, W, 0.0, cfg[5])
        eng.bind({nm: x.detach().float().contiguous() for nm, x in zip(_names(cfg), tensors)}, None)
        _engines[key] = eng
    px = SparsePixels.from_dense(image.float())
    out = torch.empty(n, cfg[1], device=image.device)
    eng.forward(px.coords, px.values, n, out, train=False, seed=0, log_pixels=px.value_mode, noise_std=0.0)
    return out


_LIB.impl("densenet_embed", _densenet_cpu, "CPU")
_LIB.impl("densenet_embed", _densenet_hip, "CUDA")
