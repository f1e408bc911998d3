"""Per-prong classification head (reference: layers/prong_target_decoder.py:8-41): halving Linear-BN-PReLU-Dropout
blocks down to >= 8 features, then Linear(final, classes)."""
import torch
from torch import Tensor, nn

from transformercvn.options import Options
from transformercvn.network.layers.encoder import create_linear_block


class ProngTargetDecoder(nn.Module):
    def __init__(self, options: Options, num_hidden: int, output_dim: int):
        super().__init__()
        self.output_dim = output_dim
        self.hidden_layers, final_dimension, self.widths = self.create_decoder_layers(options, num_hidden)
        self.output_layer = nn.Linear(final_dimension, output_dim)

    @staticmethod
    def create_decoder_layers(options: Options, num_layers: int):
        """Returns (layers, in_features of the output layer, hidden widths).  Quirk kept from the reference
        (prong_target_decoder.py:19-32): the reported final width is the *last computed* half-width, which is the real
        width only if the loop did not stop early."""
        width = options.hidden_dim
        reported = width
        modules, widths = [], []
        for _ in range(num_layers):
            reported = width // 2
            if reported < 8:
                break
            modules.extend(create_linear_block(width, reported, options))
            widths.append(reported)
            width = reported
        return nn.Sequential(*modules), reported, widths

    def forward(self, hidden: Tensor) -> Tensor:
        """[T, B, hidden_dim] -> [T, B, classes] (reference :34-41): the Linear-BN-PReLU-Dropout blocks and the output layer;
        BatchNorm1d sees the zeroed padding tokens exactly like the reference.  HIP row kernels eagerly (forward only), ATen when
        scripted (TorchScript export)."""
        if torch.jit.is_scripting():
            T, B, D = hidden.shape
            h = self.output_layer(self.hidden_layers(hidden.reshape(T * B, D)))
            return h.reshape(T, B, self.output_dim)
        return self._hip_forward(hidden)

    @torch.jit.unused
    def _hip_forward(self, hidden: Tensor) -> Tensor:
        import torch
        from torch import nn
        from transformercvn.hip import rowops
        T, B, D = hidden.shape
        h = hidden.reshape(T * B, D)
        mods = list(self.hidden_layers)
        i = blk = 0
        while i < len(mods):
            lin = mods[i]                     # create_linear_block: [Linear, BatchNorm1d?, PReLU | ReLU, Dropout?]
            j = i + 1
            norm = mods[j] if j < len(mods) and isinstance(mods[j], nn.BatchNorm1d) else None
            j += norm is not None
            act = mods[j]
            j += 1
            drop = mods[j] if j < len(mods) and isinstance(mods[j], nn.Dropout) else None
            j += drop is not None
            h = rowops.linear(h, lin.weight, lin.bias)
            p = drop.p if (drop is not None and self.training) else 0.0
            seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0
            h = rowops.bn_prelu(h, norm, getattr(act, "weight", None), self.training, p, seed, 0x7000 + blk)
            i = j
            blk += 1
        h = rowops.linear(h, self.output_layer.weight, self.output_layer.bias)
        return h.reshape(T, B, self.output_dim)
