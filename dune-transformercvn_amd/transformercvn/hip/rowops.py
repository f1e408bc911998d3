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


def _p(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _norm_of(norm):
    """(BatchNorm1d or None): the reference's LinearBlock has nn.Identity() in its place when options.linear_batch_norm is False."""
    return norm if isinstance(norm, nn.BatchNorm1d) else None


def bn_prelu(x: Tensor, norm, slope, training: bool, drop_p: float = 0.0, seed: int = 0, stream_id: int = 0) -> Tensor:
    """dropout(act(norm(x))) over rows: norm = BatchNorm1d (running statistics updated in training mode) or None / nn.Identity (option
    linear_batch_norm False); slope = the PReLU weight or None for ReLU (option linear_prelu_activation False)."""
    _need_cuda(x, "BatchNorm1d-PReLU")
    x = x.detach().float().contiguous()
    rows, ch = x.shape
    y = torch.empty_like(x)
    norm = _norm_of(norm)
    scratch = torch.empty(2 * ch, device=x.device)
    check(lib.tcvn_rows_bn_prelu_forward(C.c_void_p(x.data_ptr()), x.stride(0), rows, ch, _p(norm.weight if norm else None),
                                         _p(norm.bias if norm else None), _p(slope), _p(norm.running_mean if norm else None),
                                         _p(norm.running_var if norm else None), C.c_void_p(y.data_ptr()), y.stride(0),
                                         C.c_void_p(scratch.data_ptr()), int(training), float(drop_p), C.c_uint64(seed),
                                         C.c_uint32(stream_id), _st()), "rows_bn_prelu_forward")
    if training and norm is not None:
        norm.num_batches_tracked += 1
    return y


class FeatureMLP:
    """Forward / backward of ``ProngFeatureEmbedding.embedding`` (a chain of LinearBlocks: Linear(no bias) - BatchNorm1d - PReLU -
    Dropout, or their option variants Linear(bias) - Identity / ReLU; reference layers/prong_feature_embedding.py:7-33, :36-78) on the HIP row kernels, inside the fused training step: the
    forward keeps what the backward needs, the backward accumulates parameter gradients into the runtime's gradient arena views."""

    def __init__(self, blocks):
        self.blocks = list(blocks)
        self.saved = []

    def forward(self, x: Tensor, training: bool, seed: int) -> Tensor:
        self.saved = []
        for i, blk in enumerate(self.blocks):
            x = x.detach().float().contiguous()
            z = linear(x, blk.linear.weight, blk.linear.bias)
            rows, ch = z.shape
            y = torch.empty_like(z)
            stat = torch.empty(2 * ch, device=z.device)
            p = float(blk.dropout.p) if training else 0.0
            norm, slope = _norm_of(blk.norm), getattr(blk.activation, "weight", None)
            check(lib.tcvn_rows_bn_prelu_forward(C.c_void_p(z.data_ptr()), z.stride(0), rows, ch, _p(norm.weight if norm else None),
                                                 _p(norm.bias if norm else None), _p(slope), _p(norm.running_mean if norm else None),
                                                 _p(norm.running_var if norm else None),
                                                 C.c_void_p(y.data_ptr()), y.stride(0), C.c_void_p(stat.data_ptr()), int(training), p,
                                                 C.c_uint64(seed), C.c_uint32(0x4800 + i), _st()), "rows_bn_prelu_forward")
            if training and norm is not None:
                norm.num_batches_tracked += 1
            self.saved.append((x, z, stat, p, seed, 0x4800 + i))
            x = y
        return x

    def backward(self, dy: Tensor, grads) -> None:
        """grads: module parameter -> gradient tensor (views of the arena) to accumulate into."""
        dy = dy.contiguous()
        for blk, (x, z, stat, p, seed, sid) in zip(reversed(self.blocks), reversed(self.saved)):
            rows, ch = z.shape
            dz = torch.empty_like(z)
            norm, slope = _norm_of(blk.norm), getattr(blk.activation, "weight", None)
            check(lib.tcvn_rows_bn_prelu_backward(C.c_void_p(z.data_ptr()), z.stride(0), C.c_void_p(dy.data_ptr()), dy.stride(0), rows, ch,
                                                  _p(norm.weight if norm else None), _p(norm.bias if norm else None),
                                                  _p(slope), C.c_void_p(stat.data_ptr()),
                                                  C.c_void_p(dz.data_ptr()), dz.stride(0), _p(grads[norm.weight] if norm else None),
                                                  _p(grads[norm.bias] if norm else None), _p(grads[slope] if slope is not None else None),
                                                  float(p), C.c_uint64(seed), C.c_uint32(sid), _st()), "rows_bn_prelu_backward")
            dx = torch.empty_like(x)
            gb = grads.get(blk.linear.bias) if blk.linear.bias is not None else None
            check(lib.tcvn_linear_backward(C.c_void_p(dz.data_ptr()), dz.stride(0), C.c_void_p(x.data_ptr()), x.stride(0),
                                           C.c_void_p(blk.linear.weight.data_ptr()), C.c_void_p(dx.data_ptr()), dx.stride(0),
                                           C.c_void_p(grads[blk.linear.weight].data_ptr()), C.c_void_p(0 if gb is None else gb.data_ptr()),
                                           rows, blk.linear.weight.shape[0], blk.linear.weight.shape[1], _st()), "linear_backward")
            dy = dx
        self.saved = []
