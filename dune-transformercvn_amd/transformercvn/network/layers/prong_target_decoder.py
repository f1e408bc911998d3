"""Per-prong classification head (reference: layers/prong_target_decoder.py:8-41): halving Linear-BN-PReLU-Dropout
blocks down to >= 8 features, then Linear(final, classes)."""
from torch import nn

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
