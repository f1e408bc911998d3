"""Forward-only row operators of libtcvn_hip.so behind the holder modules' own ``forward`` (LinearBlock, ProngDecoder,
ProngTargetDecoder): ``tcvn_linear_forward`` and ``tcvn_rows_bn_prelu_forward``.  GPU only, fp32, no autograd -- training runs
through the fused network step (``HipRuntime``); these exist so that the reference's sub-module call surface
(CreateCompiled.ipynb cells 7-8, Evaluate.ipynb) keeps working stage by stage."""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from ._lib import lib, check


def _need_cuda(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"transformercvn (MI355X build): {what} runs on the GPU only; there is no CPU fallback")


def _st() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def linear(x: Tensor, weight: Tensor, bias: Tensor = None) -> Tensor:
    """y = x W^T + b over the rows of a 2-d fp32 matrix."""
    _need_cuda(x, "Linear")
    x = x.detach().float().contiguous()
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    y = torch.empty(x.shape[0], w.shape[0], device=x.device)
    check(lib.tcvn_linear_forward(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(w.data_ptr()),
                                  C.c_void_p(0 if b is None else b.data_ptr()), C.c_void_p(y.data_ptr()), y.stride(0), x.shape[0],
                                  w.shape[0], w.shape[1], _st()), "linear_forward")
    return y


def bn_prelu(x: Tensor, norm: nn.BatchNorm1d, slope: Tensor, training: bool, drop_p: float = 0.0, seed: int = 0,
             stream_id: int = 0) -> Tensor:
    """dropout(prelu(batchnorm1d(x))) over rows; updates the module's running statistics in training mode."""
    _need_cuda(x, "BatchNorm1d-PReLU")
    x = x.detach().float().contiguous()
    rows, ch = x.shape
    y = torch.empty_like(x)
    scratch = torch.empty(2 * ch, device=x.device)
    check(lib.tcvn_rows_bn_prelu_forward(C.c_void_p(x.data_ptr()), x.stride(0), rows, ch, C.c_void_p(norm.weight.data_ptr()),
                                         C.c_void_p(norm.bias.data_ptr()), C.c_void_p(slope.data_ptr()),
                                         C.c_void_p(norm.running_mean.data_ptr()), C.c_void_p(norm.running_var.data_ptr()),
                                         C.c_void_p(y.data_ptr()), y.stride(0), C.c_void_p(scratch.data_ptr()), int(training),
                                         float(drop_p), C.c_uint64(seed), C.c_uint32(stream_id), _st()), "rows_bn_prelu_forward")
    if training:
        norm.num_batches_tracked += 1
    return y
