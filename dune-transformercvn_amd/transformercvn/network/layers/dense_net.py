"""DenseNet pixel-map embedder: parameter container + HIP execution.

The module tree reproduces the reference's parameter/buffer names (reference: transformercvn/network/layers/
dense_net.py:8-167 -- ``features.conv0 ... features.dense<i>.layers.<j>.{bottleneck_block,output_block} ...
features.transition<i> ... features.final_norm/final_relu``, ``output_block.{linear,norm,relu}``) so checkpoints load
strictly.  The sub-modules are only *holders*: no torch convolution is ever called.  ``forward`` hands the tensors to
the gfx950 engine (csrc/densenet.hip) which runs the whole stack as fused implicit-GEMM kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence

import torch
from torch import Tensor, nn

from transformercvn.hip.pixels import SparsePixels
from transformercvn.hip import torch_ops  # noqa: F401  (registers tcvn::densenet_embed with the dispatcher; no GPU needed)


def _holder(**mods: nn.Module) -> nn.Sequential:
    return nn.Sequential(OrderedDict(mods))


class Bottleneck(nn.Module):
    """BN-PReLU-conv1x1 -> BN-PReLU-conv3x3-Dropout, output concatenated to the input (dense_net.py:8-45)."""

    def __init__(self, input_features: int, growth_rate: int, batch_norm_size: int, dropout: float) -> None:
        super().__init__()
        mid = batch_norm_size * growth_rate
        self.bottleneck_block = _holder(norm1=nn.BatchNorm2d(input_features), relu1=nn.PReLU(input_features),
                                        conv1=nn.Conv2d(input_features, mid, kernel_size=1))
        self.output_block = _holder(norm2=nn.BatchNorm2d(mid), relu2=nn.PReLU(mid),
                                    conv2=nn.Conv2d(mid, growth_rate, kernel_size=3, padding=1), dropout=nn.Dropout(dropout))

    def forward(self, x: Tensor) -> Tensor:
        """ATen execution of the holder modules -- reached only through torch.jit.script export (see DenseNet.forward)."""
        return torch.cat((x, self.output_block(self.bottleneck_block(x))), dim=1)


class DenseBlock(nn.Module):
    def __init__(self, num_layers: int, input_features: int, batch_norm_size: int, growth_rate: int, dropout: float) -> None:
        super().__init__()
        self.layers = nn.ModuleList(Bottleneck(input_features + i * growth_rate, growth_rate, batch_norm_size, dropout)
                                    for i in range(num_layers))

    def forward(self, x: Tensor) -> Tensor:
        for layer in self.layers:
            x = layer(x)
        return x


class Transition(nn.Sequential):
    def __init__(self, input_features: int, output_features: int) -> None:
        super().__init__(OrderedDict(norm=nn.BatchNorm2d(input_features), relu=nn.PReLU(input_features),
                                     conv=nn.Conv2d(input_features, output_features, kernel_size=1),
                                     pooling=nn.AvgPool2d(kernel_size=2, stride=2)))


class DenseNet(nn.Module):
    __constants__ = ["export_ops"]       # a script-time constant: the branch not taken is not compiled (the ATen export holds no tcvn:: node)

    def __init__(self, input_features: int, output_features: int, initial_latent_features: int = 64, growth_rate: int = 32,
                 batch_norm_size: int = 4, block_config: Sequence[int] = (6, 12, 24, 16), dropout: float = 0.0) -> None:
        super().__init__()
        self.export_ops = False
        self.op_tensors = []
        self.op_cfg = []
        self.hyper = dict(in_ch=input_features, out_dim=output_features, init_ch=initial_latent_features, growth=growth_rate,
                          bn_size=batch_norm_size, layers=list(block_config), dropout=float(dropout))
        c = initial_latent_features
        self.features = _holder(conv0=nn.Conv2d(input_features, c, kernel_size=7, padding=3, stride=2),
                                norm0=nn.BatchNorm2d(c), relu0=nn.PReLU(c), pooling0=nn.AvgPool2d(kernel_size=3, stride=2))
        for i, n in enumerate(block_config):
            self.features.add_module(f"dense{i + 1}", DenseBlock(n, c, batch_norm_size, growth_rate, dropout))
            c += n * growth_rate
            if i + 1 != len(block_config):
                self.features.add_module(f"transition{i + 1}", Transition(c, c // 2))
                c //= 2
        self.features.add_module("final_norm", nn.BatchNorm2d(c))
        self.features.add_module("final_relu", nn.PReLU(c))
        self.condense = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())
        self.output_block = _holder(linear=nn.Linear(c, output_features, bias=False), norm=nn.BatchNorm1d(output_features),
                                    relu=nn.PReLU(output_features), dropout=nn.Dropout(dropout))
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------------------------------------------------
    def hip_tensors(self) -> Dict[str, Tensor]:
        t = dict(self.named_parameters())
        t.update({k: v for k, v in self.named_buffers() if v.is_floating_point()})
        return t

    def hip_engine(self, mode: int, H: int, W: int):
        from transformercvn.hip.engine import DenseNetEngine
        key = (mode, H, W)
        if self._engine is None or self._engine_key != key:
            h = self.hyper
            self._engine = DenseNetEngine(h["in_ch"], h["out_dim"], h["init_ch"], h["growth"], h["bn_size"], h["layers"], H, W,
                                          h["dropout"], mode)
            self._engine_key = key
            self._bound_sig = None
        return self._engine

    def batch_norms(self):
        return [m for m in self.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]

    def forward(self, x: Tensor) -> Tensor:
        """Eager: the gfx950 engine (``x`` is a SparsePixels bundle or a dense NCHW map on the GPU; no autograd, no CPU fallback).
        Under ``torch.jit.script`` (CreateCompiled.ipynb cells 6-14: TorchScript export for CPU inference in LArSoft) the holder
        modules are real torch modules with the reference's parameters, so the exported graph runs them through ATen."""
        if torch.jit.is_scripting():
            if self.export_ops:       # one dispatcher operator: CPU tensors -> ATen, GPU tensors -> libtcvn_hip.so (hip/torch_ops.py)
                return torch.ops.tcvn.densenet_embed(x, self.op_tensors, self.op_cfg)
            return self.output_block(self.condense(self.features(x)))
        return self._hip_forward(x)

    def use_export_ops(self, on: bool = True) -> None:
        """Call before torch.jit.script: the scripted embedder becomes one ``tcvn::densenet_embed`` node over this module's parameters
        (eval mode) instead of the ATen graph of the holder modules."""
        self.export_ops = bool(on)
        if on:
            self.op_tensors, self.op_cfg = torch_ops.embedder_tensors(self)
        else:
            self.op_tensors, self.op_cfg = [], []

    @torch.jit.unused
    def _hip_forward(self, x: Tensor) -> Tensor:
        if not isinstance(x, SparsePixels):
            x = SparsePixels.from_dense(x)
        if not x.coords.is_cuda:
            raise RuntimeError("transformercvn (MI355X build): the DenseNet embedder runs on the GPU only -- move the model "
                               "and the batch to cuda; there is no CPU fallback")
        mode = getattr(self, "hip_mode", 0)
        eng = self.hip_engine(mode, x.shape[0], x.shape[1])
        tensors = self.hip_tensors()
        sig = tuple(t.data_ptr() for t in tensors.values())
        if getattr(self, "_bound_sig", None) != sig:
            eng.bind({k: v.detach() for k, v in tensors.items()}, None)
            self._bound_sig = sig
        n_img = x.resolve_count()
        out = torch.empty(n_img, self.hyper["out_dim"], device=x.coords.device)
        with torch.no_grad():
            eng.forward(x.coords, x.values, n_img, out, train=self.training, seed=0, log_pixels=x.value_mode, noise_std=0.0)
            if self.training:
                torch._foreach_add_([m.num_batches_tracked for m in self.batch_norms()], 1)
        return out


# TorchScript attribute types of the operator export (real types: this module uses postponed annotations, which TorchScript cannot resolve)
DenseNet.__annotations__ = {"op_tensors": List[Tensor], "op_cfg": List[int]}
