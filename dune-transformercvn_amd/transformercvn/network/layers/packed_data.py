"""Ragged prong <-> padded [B, P, C] index helpers (reference: transformercvn/network/layers/packed_data.py:59-76).

On the HIP path the gather/scatter itself is folded into the token gather of the encoder kernel; these helpers only
produce the index tensors (and keep the reference's function names for callers that use them directly)."""
from typing import Tuple

import torch
from torch import Tensor


def pack_indices(mask: Tensor) -> Tuple[Tensor, Tensor]:
    """I1 = event index, I2 = slot index of every true mask entry, row-major."""
    nz = torch.nonzero(mask)
    return nz[:, 0], nz[:, 1]


def masked_pack_1d_precomputed(data: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    i1, i2 = pack_indices(mask)
    return data[i1, i2], i1, i2


def masked_pad_1d_precomputed(packed_data: Tensor, I1: Tensor, I2: Tensor, batch_size: int, max_length: int) -> Tensor:
    out = torch.zeros(batch_size, max_length, packed_data.shape[1], dtype=packed_data.dtype, device=packed_data.device)
    out[I1, I2] = packed_data
    return out


def token_rows(prong_mask: Tensor, batch_size: int) -> Tensor:
    """tok_row [B, 1+P] int32 for the HIP encoder: row index into the (event rows, packed prong rows) matrix, -1 = padding."""
    flat = prong_mask.reshape(-1)
    packed = torch.cumsum(flat.to(torch.int32), 0, dtype=torch.int32) - 1 + batch_size
    prong = torch.where(flat, packed, torch.full_like(packed, -1)).view_as(prong_mask)
    event = torch.arange(batch_size, dtype=torch.int32, device=prong_mask.device).view(-1, 1)
    return torch.cat((event, prong), dim=1).contiguous()
