"""SDXL-style pixel-map embedder: parameter container + HIP execution (reference: transformercvn/network/layers/
sdxl_net.py:7-42, which instantiates ``diffusers.models.vae.Encoder`` -- a third-party module that is not vendored, pinned
or installed here; see oracle/sdxl_oracle.py for the restated block definitions and the "parity unpinned" note).

The module tree reproduces that encoder's parameter names (``encoder.conv_in``, ``encoder.down_blocks.<i>.resnets.<j>.{norm1,
conv1,norm2,conv2,conv_shortcut}``, ``...downsamplers.0.conv``, ``encoder.mid_block.{attentions.0.{group_norm,to_q,to_k,to_v,
to_out.0},resnets.<j>}``, ``encoder.conv_norm_out``, ``encoder.conv_out``, ``output_layer.1``) so checkpoints of the reference
load strictly.  The sub-modules are holders only; ``forward`` hands the tensors to the gfx950 engine (csrc/sdxl.hip)."""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor, nn

from transformercvn.hip.pixels import SparsePixels

GN_EPS = 1e-6


class _Resnet(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=GN_EPS)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=GN_EPS)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Downsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)          # applied after F.pad(0,1,0,1) in diffusers


class _DownBlock(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int, downsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin, cout, groups), _Resnet(cout, cout, groups)])
        if downsample:
            self.downsamplers = nn.ModuleList([_Downsample(cout)])


class _Attention(nn.Module):
    def __init__(self, c: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=GN_EPS)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _MidBlock(nn.Module):
    def __init__(self, c: int, groups: int):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(c, groups)])
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])


class _Encoder(nn.Module):
    def __init__(self, in_ch: int, chans, groups: int):
        super().__init__()
        self.conv_in = nn.Conv2d(in_ch, chans[0], 3, padding=1)
        blocks, cin = [], chans[0]
        for i, c in enumerate(chans):
            blocks.append(_DownBlock(cin, c, groups, i + 1 != len(chans)))
            cin = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _MidBlock(chans[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, chans[-1], eps=GN_EPS)
        self.conv_out = nn.Conv2d(chans[-1], chans[-1], 3, padding=1)


class SDXLNet(nn.Module):
    def __init__(self, input_features: int, output_features: int, init_block_dim: int, repeat_block_dim: int, num_blocks: int,
                 norm_num_groups: int = 8):
        super().__init__()
        if norm_num_groups != 1:
            raise NotImplementedError("the MI355X SDXL embedder implements GroupNorm with one group (what the reference passes: "
                                      "networks/neutrino_full_sdxl_network.py:14)")
        chans, d = [], init_block_dim
        for _ in range(num_blocks):
            chans += [d] * repeat_block_dim
            d *= 2
        chans.append(output_features)
        self.hyper = dict(in_ch=input_features, out_dim=output_features, init_ch=init_block_dim, repeat=repeat_block_dim,
                          num_blocks=num_blocks)
        self.encoder = _Encoder(input_features, chans, norm_num_groups)
        self.output_layer = nn.Sequential(nn.Flatten(), nn.Linear(output_features, output_features))
        self._engine = None
        self._engine_key = None

    def hip_tensors(self) -> Dict[str, Tensor]:
        return dict(self.named_parameters())

    def hip_engine(self, mode: int, H: int, W: int):
        from transformercvn.hip.engine import SdxlEngine
        key = (mode, H, W)
        if self._engine is None or self._engine_key != key:
            h = self.hyper
            self._engine = SdxlEngine(h["in_ch"], h["out_dim"], h["init_ch"], h["repeat"], h["num_blocks"], H, W, mode)
            self._engine_key = key
            self._bound_sig = None
        return self._engine

    def forward(self, x: Tensor) -> Tensor:
        """Stand-alone forward of the embedder (no autograd): ``x`` is a SparsePixels bundle or a dense NCHW map on the GPU.
        (TorchScript export of this embedder is not provided: the reference exports the DenseNet model only, CreateCompiled.ipynb.)"""
        return self._hip_forward(x)

    @torch.jit.unused
    def _hip_forward(self, x: Tensor) -> Tensor:
        if not isinstance(x, SparsePixels):
            x = SparsePixels.from_dense(x)
        if not x.coords.is_cuda:
            raise RuntimeError("transformercvn (MI355X build): the SDXL embedder runs on the GPU only; there is no CPU fallback")
        eng = self.hip_engine(getattr(self, "hip_mode", 0), x.shape[0], x.shape[1])
        tensors = self.hip_tensors()
        sig = tuple(t.data_ptr() for t in tensors.values())
        if getattr(self, "_bound_sig", None) != sig:
            eng.bind({k: v.detach() for k, v in tensors.items()}, None)
            self._bound_sig = sig
        n_img = x.resolve_count()
        out = torch.empty(n_img, self.hyper["out_dim"], device=x.coords.device)
        with torch.no_grad():
            eng.forward(x.coords, x.values, n_img, out, train=self.training, seed=0, log_pixels=x.value_mode, noise_std=0.0)
        return out
