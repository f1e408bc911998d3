"""Full TransformerCVN network: pixel/feature embeddings -> token set -> encoder -> decoders.

Module tree and constructor signatures follow the reference (transformercvn/network/networks/
neutrino_full_base_network.py:17-188) so state_dicts load strictly; ``forward`` runs the whole step on the MI355X
through ``transformercvn.hip.runtime.HipRuntime`` (no torch operator is used for the arithmetic).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from transformercvn.hip.pixels import SparsePixels
from transformercvn.network.layers.packed_data import masked_pack_1d_precomputed, masked_pad_1d_precomputed
from transformercvn.network.layers.prong_custom_bert_encoder import ProngCustomBertEncoder
from transformercvn.network.layers.prong_decoder import ProngDecoder
from transformercvn.network.layers.prong_feature_embedding import ProngFeatureEmbedding, LinearBlock
from transformercvn.network.layers.prong_masked_mobilenet_embedding import make_divisible_channel_count
from transformercvn.network.layers.prong_target_decoder import ProngTargetDecoder
from transformercvn.options import Options


class BaseProngEmbedding(nn.Module, ABC):
    @abstractmethod
    def create_pixel_embedding(self, options: Options, pixel_dim: int, output_dim: int):
        raise NotImplementedError()

    def create_feature_embedding(self, options: Options, features_dim: int, extra_dim: int):
        return ProngFeatureEmbedding(options=options, sequence_dim=features_dim, extra_dim=extra_dim,
                                     output_dim=self.feature_embedding_dim)

    def __init__(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int):
        super().__init__()
        self.hidden_dim = options.hidden_dim
        self.one_hot_pixels = options.one_hot_pixels
        # widths rounded to multiples of 8 (neutrino_full_base_network.py:51-53)
        self.pixel_embedding_dim = make_divisible_channel_count(options.pixel_embedding_dim, 8)
        self.feature_embedding_dim = make_divisible_channel_count(options.feature_embedding_dim, 8)
        self.position_embedding_dim = make_divisible_channel_count(options.position_embedding_dim, 8)
        self.feature_embedding = self.create_feature_embedding(options, features_dim, extra_dim)
        self.prong_pixel_embedding = self.create_pixel_embedding(options, pixel_dim, output_dim=self.pixel_embedding_dim)
        # the event map embedding is wider: it also fills the slot prongs use for their feature embedding (:67-71)
        self.event_pixel_embedding = self.create_pixel_embedding(
            options, pixel_dim, output_dim=self.pixel_embedding_dim + self.feature_embedding_dim)
        self.event_position_embedding = nn.Parameter(torch.randn(1, self.position_embedding_dim))
        # never used in forward -- prongs receive event_position_embedding (reference quirk, :107); kept for the state_dict
        self.prong_position_embedding = nn.Parameter(torch.randn(1, self.position_embedding_dim))
        self.combined_embedding = LinearBlock(
            options, self.feature_embedding_dim + self.pixel_embedding_dim + self.position_embedding_dim, options.hidden_dim)


    def forward(self, features: Tensor, extra: Tensor, event_pixels: Tensor, event_mask: Tensor, prong_pixels: Tensor,
                prong_mask: Tensor) -> Tuple[Tensor, Tensor]:
        """-> (tokens [B, 1+P, hidden], mask [B, 1+P]) like the reference (:87-125): both pixel embedders, position embeddings, the
        shared LinearBlock over event + packed prong rows, masked pad, event token first.  Eager: the two embedder engines and the
        embedding stage of the head engine (tcvn_head_embed; pixels may be SparsePixels bundles; forward only, no autograd).
        Under torch.jit.script (TorchScript export, CreateCompiled.ipynb cells 6-14): the same graph through ATen."""
        if torch.jit.is_scripting():
            batch_size, max_prongs, _ = features.shape
            event_embeddings = torch.cat((self.event_pixel_embedding(event_pixels),
                                          self.event_position_embedding.expand(batch_size, -1)), dim=1)
            packed, I1, I2 = masked_pack_1d_precomputed(features, prong_mask)
            prong_pixel_embeddings = self.prong_pixel_embedding(prong_pixels)
            # prongs also receive the *event* position embedding (reference quirk, :107)
            prong_embeddings = torch.cat((self.feature_embedding(packed, extra[I1]), prong_pixel_embeddings,
                                          self.event_position_embedding.expand(prong_pixel_embeddings.shape[0], -1)), dim=1)
            combined = self.combined_embedding(torch.cat((event_embeddings, prong_embeddings), dim=0))
            padded = masked_pad_1d_precomputed(combined[batch_size:], I1, I2, batch_size, max_prongs)
            tokens = torch.cat((combined[:batch_size].view(batch_size, 1, -1), padded), dim=1)
        else:
            tokens = self._hip_forward(features, extra, event_pixels, event_mask, prong_pixels, prong_mask)
        return tokens, torch.cat((event_mask.to(tokens.device), prong_mask.to(tokens.device)), dim=1)

    @torch.jit.unused
    def _hip_forward(self, features: Tensor, extra: Tensor, event_pixels: Tensor, event_mask: Tensor, prong_pixels: Tensor,
                     prong_mask: Tensor) -> Tensor:
        from transformercvn.hip.owners import owner_of
        net = owner_of(self)
        if net is None:
            raise RuntimeError("BaseProngEmbedding.forward needs the owning NeutrinoBaseNetwork (its HIP runtime holds the plans)")
        if not isinstance(event_pixels, SparsePixels):
            event_pixels = SparsePixels.from_dense(event_pixels)
        if not isinstance(prong_pixels, SparsePixels):
            prong_pixels = SparsePixels.from_dense(prong_pixels)
        return net.hip_runtime().embed(features, extra, event_pixels, event_mask, prong_pixels, prong_mask, self.training)


class NeutrinoBaseNetwork(nn.Module):
    @abstractmethod
    def create_prong_embedding(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int):
        raise NotImplementedError

    def __init__(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int, num_prong_classes: int,
                 num_event_classes: int):
        super().__init__()
        self.prong_embedding = self.create_prong_embedding(options, features_dim, extra_dim, pixel_dim)
        self.encoder = ProngCustomBertEncoder(options, options.hidden_dim, options.num_attention_heads, options.dropout,
                                              options.transformer_activation, options.transformer_norm_first)
        self.event_decoder = ProngDecoder(options, num_event_classes)
        self.prong_decoder = ProngTargetDecoder(options, options.num_prong_decoder_layers, num_prong_classes)
        self._options = options
        self._runtime = None
        from transformercvn.hip.owners import register
        register(self.prong_embedding, self)
        register(self.encoder, self)
        self.pixel_shape: Tuple[int, int] = (400, 280)

    def prepare_export(self, use_ops: bool = False):
        """TorchScript export (CreateCompiled.ipynb cells 6-14).  use_ops=False (default): every stage scripts its ATen branch -- the
        file loads anywhere.  use_ops=True: the two pixel-map embedders script to the registered operator tcvn::densenet_embed, which
        runs ATen on CPU tensors and the gfx950 kernels on GPU tensors (needs `import transformercvn` in the loading process)."""
        for emb in (self.prong_embedding.prong_pixel_embedding, self.prong_embedding.event_pixel_embedding):
            if hasattr(emb, "use_export_ops"):
                emb.hip_mode = self.hip_runtime().mode
                emb.use_export_ops(use_ops)
        return self

    def hip_runtime(self):
        """Lazily created fused runtime.  Precision: ``options.hip_precision`` ('fp32' parity mode or 'bf16') or, when the option
        file does not name it, what the Lightning trainer was given (``train.py -fp16`` -> precision 16 -> bf16 engines; see
        NeutrinoFullBaseTrainer.adopt_trainer_precision)."""
        if self._runtime is None:
            from transformercvn.hip.runtime import HipRuntime
            precision = getattr(self, "_precision_override", None) or getattr(self._options, "hip_precision", "fp32")
            self._runtime = HipRuntime(self, self._options, self.pixel_shape, precision, seed=int(getattr(self._options, "seed", 0)))
        return self._runtime

    def forward(self, features: Tensor, extra: Tensor, event_pixels: Tensor, event_mask: Tensor, prong_pixels: Tensor,
                prong_mask: Tensor, counts: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, Tensor]:
        """-> (event_logits [B, Ce], prong_logits [B, P, Cp]) (reference :166-188).  Eager: the fused MI355X step (pixels are
        SparsePixels bundles or dense NCHW maps).  Under torch.jit.script: the stage modules through ATen (export)."""
        if torch.jit.is_scripting():
            tokens, mask = self.prong_embedding(features, extra, event_pixels, event_mask, prong_pixels, prong_mask)
            hidden, padding_mask, sequence_mask = self.encoder(tokens, mask)
            return self.event_decoder(hidden[0]), self.prong_decoder(hidden[1:]).transpose(0, 1)
        return self._hip_forward(features, extra, event_pixels, event_mask, prong_pixels, prong_mask, counts)

    @torch.jit.unused
    def _hip_forward(self, features: Tensor, extra: Tensor, event_pixels: Tensor, event_mask: Tensor, prong_pixels: Tensor,
                     prong_mask: Tensor, counts: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, Tensor]:
        if not isinstance(event_pixels, SparsePixels):
            event_pixels = SparsePixels.from_dense(event_pixels)
        if not isinstance(prong_pixels, SparsePixels):
            prong_pixels = SparsePixels.from_dense(prong_pixels)
        return self.hip_runtime().forward(features, extra, event_pixels, event_mask, prong_pixels, prong_mask, counts)
