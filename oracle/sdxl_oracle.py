"""CPU restatement of the SDXL-style embedder of the reference (transformercvn/network/layers/sdxl_net.py:7-42,
networks/neutrino_full_sdxl_network.py:6-20) -- TEST INFRASTRUCTURE, not product code: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

*** PARITY UNPINNED ***  The arithmetic of this embedder lives in the third-party package `diffusers`
(`from diffusers.models.vae import Encoder`, sdxl_net.py:4), which is not vendored in the reference, not pinned by any
requirements/lock file, and not installed here (no network).  The reference holds no test, golden vector or notebook output
for this path (CreateCompiled.ipynb was run for the DenseNet model only).  What follows restates the published definitions
of `diffusers.models.vae.Encoder` and its blocks as of the 0.2x releases that still expose that module path:
    Encoder            conv_in 3x3 -> DownEncoderBlock2D x len(block_out_channels) -> UNetMidBlock2D -> GroupNorm -> SiLU -> conv_out 3x3
    DownEncoderBlock2D layers_per_block (=2) ResnetBlock2D, then (all blocks but the last) Downsample2D(use_conv, padding=0):
                       F.pad (0,1,0,1) + Conv2d 3x3 stride 2
    ResnetBlock2D      GroupNorm(groups, eps 1e-6) - SiLU - conv 3x3 - GroupNorm - SiLU - dropout(0.0) - conv 3x3, plus x
                       (through a 1x1 conv_shortcut when the channel count changes), output_scale_factor 1
    UNetMidBlock2D     ResnetBlock2D - Attention(1 head, head_dim = channels, GroupNorm, residual) - ResnetBlock2D
    Attention          GroupNorm over [B,C,HW] -> to_q / to_k / to_v (Linear C->C, bias) -> softmax(q k^T / sqrt(C)) v -> to_out.0
                       (Linear C->C) -> + residual
with the constructor arguments the reference passes (sdxl_net.py:19-34; neutrino_full_sdxl_network.py:8-15):
block_out_channels [d,d,2d,2d,4d,4d,8d,8d,out] (d = options.initial_pixel_dim), norm_num_groups 1, double_z False, then
Flatten + Linear(out, out) (sdxl_net.py:36-39).  Eight stride-2 stages take 400x280 to 1x1, so the mid-block attention sees
ONE token (softmax == 1: to_q / to_k receive no gradient) and Flatten yields [N, out] -- the only in-repo fixtures for this
path are these shape facts.  State-dict key names follow that diffusers generation (`encoder.down_blocks.<i>.resnets.<j>.*`,
`...downsamplers.0.conv.*`, `encoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}.*`); older releases name the
attention parameters query/key/value/proj_attn.  Numbers produced with this file are self-consistent (oracle <-> HIP), not
reference parity, until a pinned diffusers is available."""
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

GN_EPS = 1e-6
LAYERS_PER_BLOCK = 2


def block_channels(init_dim: int, out_dim: int, repeat: int = 2, num_blocks: int = 4) -> List[int]:
    """sdxl_net.py:19-25."""
    ch, d = [], init_dim
    for _ in range(num_blocks):
        ch += [d] * repeat
        d *= 2
    return ch + [out_dim]


def _resnet_layout(p: str, cin: int, cout: int, out: Dict[str, Tuple[int, ...]]):
    out[p + ".norm1.weight"] = (cin,); out[p + ".norm1.bias"] = (cin,)
    out[p + ".conv1.weight"] = (cout, cin, 3, 3); out[p + ".conv1.bias"] = (cout,)
    out[p + ".norm2.weight"] = (cout,); out[p + ".norm2.bias"] = (cout,)
    out[p + ".conv2.weight"] = (cout, cout, 3, 3); out[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        out[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); out[p + ".conv_shortcut.bias"] = (cout,)


def sdxl_layout(prefix: str, in_ch: int, out_dim: int, init_dim: int, out: Dict[str, Tuple[int, ...]]):
    """Ordered name -> shape map of SDXLNet's state_dict (module registration order of diffusers' Encoder)."""
    chans = block_channels(init_dim, out_dim)
    e = prefix + ".encoder"
    out[e + ".conv_in.weight"] = (chans[0], in_ch, 3, 3); out[e + ".conv_in.bias"] = (chans[0],)
    cin = chans[0]
    for i, cout in enumerate(chans):
        for j in range(LAYERS_PER_BLOCK):
            _resnet_layout(f"{e}.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, out)
        if i != len(chans) - 1:
            out[f"{e}.down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            out[f"{e}.down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    c = chans[-1]
    a = e + ".mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (c,); out[a + ".group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        out[f"{a}.{n}.weight"] = (c, c); out[f"{a}.{n}.bias"] = (c,)
    for j in range(2):
        _resnet_layout(f"{e}.mid_block.resnets.{j}", c, c, out)
    out[e + ".conv_norm_out.weight"] = (c,); out[e + ".conv_norm_out.bias"] = (c,)
    out[e + ".conv_out.weight"] = (out_dim, c, 3, 3); out[e + ".conv_out.bias"] = (out_dim,)
    out[prefix + ".output_layer.1.weight"] = (out_dim, out_dim); out[prefix + ".output_layer.1.bias"] = (out_dim,)


def _gn(sd, p: str, x: Tensor) -> Tensor:
    return F.group_norm(x, 1, sd[p + ".weight"], sd[p + ".bias"], GN_EPS)


def _resnet(sd, p: str, x: Tensor) -> Tensor:
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attention(sd, p: str, x: Tensor) -> Tensor:
    B, C, H, W = x.shape
    t = _gn(sd, p + ".group_norm", x.view(B, C, H * W)).transpose(1, 2)            # [B, HW, C]
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    attn = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (C ** -0.5), dim=-1)   # one head of width C
    o = F.linear(torch.bmm(attn, v), sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def sdxl_forward(sd, prefix: str, x: Tensor, taps: Dict[str, Tensor] = None) -> Tensor:
    """SDXLNet.forward (sdxl_net.py:41-42): [N, in_ch, 400, 280] -> [N, out]."""
    e = prefix + ".encoder"
    h = F.conv2d(x, sd[e + ".conv_in.weight"], sd[e + ".conv_in.bias"], padding=1)
    if taps is not None:
        taps[prefix + ":conv_in"] = h
    i = 0
    while f"{e}.down_blocks.{i}.resnets.0.conv1.weight" in sd:
        for j in range(LAYERS_PER_BLOCK):
            h = _resnet(sd, f"{e}.down_blocks.{i}.resnets.{j}", h)
        if taps is not None:
            taps[f"{prefix}:block{i}"] = h
        d = f"{e}.down_blocks.{i}.downsamplers.0.conv"
        if (d + ".weight") in sd:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[d + ".weight"], sd[d + ".bias"], stride=2)
        i += 1
    h = _resnet(sd, e + ".mid_block.resnets.0", h)
    h = _attention(sd, e + ".mid_block.attentions.0", h)
    h = _resnet(sd, e + ".mid_block.resnets.1", h)
    if taps is not None:
        taps[prefix + ":mid"] = h
    h = F.conv2d(F.silu(_gn(sd, e + ".conv_norm_out", h)), sd[e + ".conv_out.weight"], sd[e + ".conv_out.bias"], padding=1)
    if h.shape[2] != 1 or h.shape[3] != 1:
        raise ValueError(f"SDXLNet needs a 1x1 final map for Flatten + Linear(out, out); got {tuple(h.shape[2:])}")
    return F.linear(h.flatten(1), sd[prefix + ".output_layer.1.weight"], sd[prefix + ".output_layer.1.bias"])
