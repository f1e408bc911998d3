"""DenseNet flavour of the full network (reference: networks/neutrino_full_dense_network.py:1-21)."""
from transformercvn.options import Options
from transformercvn.network.layers.dense_net import DenseNet
from transformercvn.network.networks.neutrino_full_base_network import BaseProngEmbedding, NeutrinoBaseNetwork


class DenseProngEmbedding(BaseProngEmbedding):
    def create_pixel_embedding(self, options: Options, pixel_dim: int, output_dim: int):
        in_ch = pixel_dim * 256 if self.one_hot_pixels else pixel_dim
        return DenseNet(in_ch, output_dim, options.initial_pixel_dim, options.densenet_growth_rate,
                        options.densenet_batch_norm_size, tuple(options.densenet_structure), options.dropout)


class NeutrinoDenseNetwork(NeutrinoBaseNetwork):
    def create_prong_embedding(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int):
        return DenseProngEmbedding(options, features_dim, extra_dim, pixel_dim)
