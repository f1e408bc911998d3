"""Sparse pixel-map bundle handed from ``preprocess_pixels`` to the DenseNet engine."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

VALUE_RAW_255, VALUE_LOG, VALUE_READY, VALUE_ONE_HOT = 0, 1, 2, 3     # how the scatter kernel turns `values` into pixel intensities


class SparsePixels:
    """COO pixel list (coords [nnz,3] int32 = (image, y, x); values [nnz,C] fp32) + the map shape.

    The reference materialises a dense [N,C,H,W] tensor at this point (trainers/neutrino_full_dense_trainer.py:15-24);
    here the scatter happens inside the engine, straight into the NHWC layout conv0 consumes.  ``to_dense()`` rebuilds the
    reference's tensor for callers that want it (CreateCompiled.ipynb calls ``.to_dense()`` on the result)."""

    def __init__(self, coords: Tensor, values: Tensor, shape: Tuple[int, int], value_mode: int = VALUE_RAW_255,
                 noise_std: float = 0.0, count: Optional[int] = None):
        self.coords = coords if coords.dtype == torch.int32 else coords.to(torch.int32)
        self.coords = self.coords.contiguous()
        self.values = values.to(torch.float32).contiguous()
        self.shape = tuple(shape)
        self.value_mode = value_mode
        self.noise_std = noise_std
        self.count = count

    def resolve_count(self) -> int:
        """Number of images: given by the caller, else the reference's rule `last image index + 1` (host sync)."""
        if self.count is None:
            self.count = int(self.coords[-1, 0].item()) + 1
        return self.count

    def preprocessed_values(self) -> Tensor:
        if self.value_mode == VALUE_RAW_255:
            return self.values / 255.0
        if self.value_mode == VALUE_LOG:
            return torch.log(self.values + 1)
        return self.values

    def to_dense(self) -> Tensor:
        n = self.resolve_count()
        c = self.coords.long()
        if self.value_mode == VALUE_ONE_HOT:                 # reference :47-52: 256-way one-hot per value channel
            f = self.values.shape[1]
            hot = torch.nn.functional.one_hot(self.values.long().clamp(0, 255), 256).reshape(-1, 256 * f).to(self.values.dtype)
            out = torch.zeros(n, *self.shape, 256 * f, dtype=self.values.dtype, device=self.values.device)
            out[c[:, 0], c[:, 1], c[:, 2]] = hot
            return out.permute(0, 3, 1, 2).contiguous()
        out = torch.zeros(n, *self.shape, self.values.shape[1], dtype=self.values.dtype, device=self.values.device)
        out[c[:, 0], c[:, 1], c[:, 2]] = self.preprocessed_values()
        return out.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def from_dense(x: Tensor) -> "SparsePixels":
        """Dense NCHW map -> COO list of its non-zero pixels (any channel non-zero)."""
        n, c, h, w = x.shape
        nhwc = x.permute(0, 2, 3, 1)
        idx = (nhwc != 0).any(dim=-1).nonzero(as_tuple=False)
        vals = nhwc[idx[:, 0], idx[:, 1], idx[:, 2]]
        return SparsePixels(idx.to(torch.int32), vals, (h, w), VALUE_READY, 0.0, n)
