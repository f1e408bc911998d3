"""``create_linear_block`` of the reference (transformercvn/network/layers/encoder.py:10-24): the Linear / BatchNorm1d /
PReLU / Dropout group the prong decoder is made of.  The Dropout module only exists when options.dropout > 0, which is
why the decoder's state_dict indices depend on it."""
from torch import nn

from transformercvn.options import Options


def create_linear_block(input_dim: int, output_dim: int, options: Options):
    block = [nn.Linear(input_dim, output_dim)]
    if options.linear_batch_norm:
        block.append(nn.BatchNorm1d(output_dim))
    block.append(nn.PReLU(output_dim) if options.linear_prelu_activation else nn.ReLU())
    if options.dropout > 0.0:
        block.append(nn.Dropout(options.dropout))
    return block
