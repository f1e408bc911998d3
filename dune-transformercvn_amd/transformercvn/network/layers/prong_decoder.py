"""Event classification head: one Linear(hidden_dim -> classes) (reference: layers/prong_decoder.py:7-16)."""
import torch
from torch import Tensor, nn

from transformercvn.options import Options


class ProngDecoder(nn.Module):
    def __init__(self, options: Options, output_dim: int, hidden_dim_factor: int = 1):
        super().__init__()
        self.options = options
        self.hidden_dim_factor = hidden_dim_factor
        self.hidden_layer = nn.Linear(hidden_dim_factor * options.hidden_dim, output_dim)

    def forward(self, hidden: Tensor) -> Tensor:
        """[B, hidden_dim] -> [B, classes] (reference :15-16): HIP row GEMM eagerly (forward only), ATen when scripted."""
        if torch.jit.is_scripting():
            return self.hidden_layer(hidden)
        return self._hip_forward(hidden)

    @torch.jit.unused
    def _hip_forward(self, hidden: Tensor) -> Tensor:
        from transformercvn.hip import rowops
        return rowops.linear(hidden.reshape(-1, hidden.shape[-1]), self.hidden_layer.weight,
                             self.hidden_layer.bias).reshape(hidden.shape[:-1] + (self.hidden_layer.out_features,))
