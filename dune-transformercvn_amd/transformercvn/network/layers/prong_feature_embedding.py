"""LinearBlock / ProngFeatureEmbedding parameter holders (reference: transformercvn/network/layers/
prong_feature_embedding.py:7-33, :36-78).  Linear -> BatchNorm1d | Identity -> PReLU | ReLU -> Dropout (options.linear_batch_norm,
options.linear_prelu_activation); executed by the row kernels of csrc/rows.hip through the head engine."""
from __future__ import annotations

import torch
from torch import Tensor, nn

from transformercvn.options import Options


class LinearBlock(nn.Module):
    def __init__(self, options: Options, input_dim: int, output_dim: int):
        super().__init__()
        use_bn = bool(options.linear_batch_norm)
        self.linear = nn.Linear(input_dim, output_dim, bias=not use_bn)
        self.norm = nn.BatchNorm1d(output_dim) if use_bn else nn.Identity()
        self.activation = nn.PReLU(output_dim) if options.linear_prelu_activation else nn.ReLU()
        self.dropout = nn.Dropout(options.dropout)

    def forward(self, x: Tensor) -> Tensor:
        """Linear -> BatchNorm1d -> PReLU -> Dropout (reference :25-33).  Eager: the HIP row kernels (forward only, no autograd);
        under torch.jit.script (TorchScript export) the holder modules run through ATen."""
        if torch.jit.is_scripting():
            return self.dropout(self.activation(self.norm(self.linear(x))))
        return self._hip_forward(x)

    @torch.jit.unused
    def _hip_forward(self, x: Tensor) -> Tensor:
        from transformercvn.hip import rowops
        z = rowops.linear(x, self.linear.weight, self.linear.bias)
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if (self.training and self.dropout.p > 0) else 0
        # norm = BatchNorm1d | Identity (options.linear_batch_norm), activation = PReLU | ReLU (options.linear_prelu_activation)
        return rowops.bn_prelu(z, self.norm, getattr(self.activation, "weight", None), self.training, self.dropout.p, seed, 0x5000)


class ProngFeatureEmbedding(nn.Module):
    """MLP over the reconstructed per-prong features.  With ``disable_smart_features`` (both shipped option files) its
    output is identically zero (prong_feature_embedding.py:73-78) and its parameters receive no gradient."""

    def __init__(self, options: Options, sequence_dim: int, extra_dim: int, output_dim: int):
        super().__init__()
        self.extra_dim, self.output_dim, self.sequence_dim = extra_dim, output_dim, sequence_dim
        self.disable_smart_features = bool(options.disable_smart_features)
        self.embedding = self.create_embedding_layers(options, sequence_dim + extra_dim, output_dim)
        self.embedding_dim = options.hidden_dim

    @staticmethod
    def create_embedding_layers(options: Options, input_dim: int, output_dim: int) -> nn.Sequential:
        widths = [options.initial_feature_dim]
        for _ in range(options.num_embedding_layers):          # doubling widths, capped below output_dim
            if 2 * widths[-1] >= output_dim:
                break
            widths.append(2 * widths[-1])
        dims = [input_dim] + widths + [output_dim]
        return nn.Sequential(*(LinearBlock(options, i, o) for i, o in zip(dims[:-1], dims[1:])))

    def forward(self, data: Tensor, extra: Tensor) -> Tensor:
        if self.disable_smart_features:
            return torch.zeros(data.shape[0], self.output_dim, dtype=data.dtype, device=data.device)
        return self.embedding(torch.cat([data, extra], dim=1))          # LinearBlocks: HIP row kernels eagerly, ATen when scripted
