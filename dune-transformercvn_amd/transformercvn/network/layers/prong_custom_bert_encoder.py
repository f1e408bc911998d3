"""Transformer encoder over the (event + prongs) token set (reference: layers/prong_custom_bert_encoder.py:29-75).

``self.encoder`` is a stock ``nn.TransformerEncoder`` used purely as a parameter holder so that the state_dict keys
(``encoder.layers.<l>.self_attn.in_proj_weight`` ...) and the default initialisation match the reference; note the
reference passes ``hidden_dim`` as ``dim_feedforward``.  Execution: csrc/encoder.hip via the head engine."""
import warnings

from typing import Tuple

import torch
from torch import Tensor, nn

from transformercvn.options import Options


class ProngCustomBertEncoder(nn.Module):
    def __init__(self, options: Options, hidden_dim: int, num_heads: int, dropout: float, activation: str, norm_first: bool):
        super().__init__()
        self.options = options
        layer = nn.TransformerEncoderLayer(hidden_dim, num_heads, hidden_dim, dropout, activation, norm_first=bool(norm_first))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.encoder = nn.TransformerEncoder(layer, options.num_encoder_layers)

    def forward(self, embeddings: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """(tokens [B, S, D], mask [B, S] bool) -> (hidden [S, B, D], padding_mask [B, S], sequence_mask [S, B, 1]) like the
        reference (:57-75).  Eager: the encoder stage of the head engine (tcvn_head_encode; forward only); scripted: ATen."""
        batch_size, max_particles, _ = embeddings.shape
        padding_mask = ~mask
        sequence_mask = mask.view(batch_size, max_particles, 1).transpose(0, 1).contiguous()
        if torch.jit.is_scripting():
            hidden = embeddings.transpose(0, 1).contiguous() * sequence_mask
            hidden = self.encoder(hidden, src_key_padding_mask=padding_mask) * sequence_mask
        else:
            hidden = self._hip_forward(embeddings, mask)
        return hidden, padding_mask, sequence_mask

    @torch.jit.unused
    def _hip_forward(self, embeddings: Tensor, mask: Tensor) -> Tensor:
        from transformercvn.hip.owners import owner_of
        net = owner_of(self)
        if net is None:
            raise RuntimeError("ProngCustomBertEncoder.forward needs the owning NeutrinoBaseNetwork (its HIP runtime holds the plan)")
        return net.hip_runtime().encode(embeddings, mask, self.training)
