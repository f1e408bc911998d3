"""DenseNet trainer (reference: trainers/neutrino_full_dense_trainer.py:15-67)."""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from transformercvn.options import Options
from transformercvn.hip.pixels import SparsePixels, VALUE_RAW_255, VALUE_LOG, VALUE_ONE_HOT
from transformercvn.network.networks.neutrino_full_dense_network import NeutrinoDenseNetwork
from transformercvn.network.trainers.neutrino_full_base_trainer import NeutrinoFullBaseTrainer


def sparse_to_dense(features: Tensor, coordinates: Tensor, image_size: Tuple[int, ...]) -> Tensor:
    """COO list -> dense [N, C, H, W] map; N = last image index + 1 (reference :15-24).  Kept for callers that want the
    dense tensor; the network itself consumes the COO list directly."""
    return SparsePixels(coordinates, features, tuple(image_size), value_mode=2).to_dense()


class NeutrinoFullDenseTrainer(NeutrinoFullBaseTrainer):
    def create_network(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int, num_prong_classes: int,
                       num_event_classes: int) -> NeutrinoDenseNetwork:
        return NeutrinoDenseNetwork(options, features_dim, extra_dim, pixel_dim, num_prong_classes, num_event_classes)

    def preprocess_pixels(self, pixel_coords: Tensor, pixel_values: Tensor, image_size: Tuple[int, ...]) -> SparsePixels:
        """v/255 (or log(v+1)), and in training v*(1 + N(0,1)*pixel_noise_std) (reference :46-67) -- recorded on the
        SparsePixels bundle and applied by the scatter kernel."""
        if self.options.one_hot_pixels:          # 256-way one-hot per value channel, no scaling and no noise (reference :47-52)
            return SparsePixels(pixel_coords, pixel_values, tuple(image_size), VALUE_ONE_HOT, 0.0)
        mode = VALUE_LOG if self.options.log_pixels else VALUE_RAW_255
        noise = float(self.options.pixel_noise_std) if self.training else 0.0
        return SparsePixels(pixel_coords, pixel_values, tuple(image_size), mode, noise)
