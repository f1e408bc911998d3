"""Thin Python owners of the native plans in libtcvn_hip.so.  PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import lib, check


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


class _Plan:
    """Common slot/bind/workspace handling for the DenseNet and head plans."""
    _prefix = ""

    def __init__(self):
        self.handle = C.c_void_p()
        self._ws: Optional[torch.Tensor] = None
        self._keep: List[torch.Tensor] = []

    def _fn(self, name):
        return getattr(lib, f"tcvn_{self._prefix}_{name}")

    def slots(self) -> List[Tuple[str, int, int]]:
        n = self._fn("num_slots")(self.handle)
        out = []
        buf = C.create_string_buffer(256)
        numel, kind = C.c_int64(), C.c_int()
        for i in range(n):
            check(self._fn("slot")(self.handle, i, buf, 256, C.byref(numel), C.byref(kind)), "slot")
            out.append((buf.value.decode(), numel.value, kind.value))
        return out

    def bind(self, data: Dict[str, torch.Tensor], grad: Optional[Dict[str, torch.Tensor]] = None):
        sl = self.slots()
        d = (C.c_void_p * len(sl))()
        g = (C.c_void_p * len(sl))()
        keep = []
        for i, (name, numel, kind) in enumerate(sl):
            if kind == _lib.SLOT_COUNTER:
                d[i] = None
                g[i] = None
                continue
            t = data[name]
            if t.numel() != numel or t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
                raise ValueError(f"slot {name}: expected contiguous cuda float32[{numel}], got {t.dtype} {tuple(t.shape)} {t.device}")
            d[i] = t.data_ptr()
            keep.append(t)
            gt = None if (grad is None or kind != _lib.SLOT_PARAM) else grad.get(name)
            if gt is not None:
                if gt.numel() != numel or gt.dtype != torch.float32 or not gt.is_cuda or not gt.is_contiguous():
                    raise ValueError(f"grad slot {name}: bad tensor")
                keep.append(gt)
            g[i] = None if gt is None else gt.data_ptr()
        check(self._fn("bind")(self.handle, d, g), "bind")
        self._keep = keep

    def workspace(self, nbytes: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        return self._ws

    def __del__(self):
        try:
            if self.handle:
                self._fn("destroy")(self.handle)
        except Exception:
            pass


class DenseNetEngine(_Plan):
    _prefix = "densenet"

    def __init__(self, in_ch: int, out_dim: int, init_ch: int, growth: int, bn_size: int, layers, H: int, W: int,
                 dropout: float, mode: int):
        super().__init__()
        cfg = _lib.DenseNetCfg()
        cfg.in_ch, cfg.out_dim, cfg.init_ch, cfg.growth, cfg.bn_size = in_ch, out_dim, init_ch, growth, bn_size
        cfg.n_blocks = len(layers)
        for i, l in enumerate(layers):
            cfg.layers[i] = l
        cfg.H, cfg.W, cfg.dropout, cfg.mode = H, W, dropout, mode
        self.cfg = cfg
        self.mode = mode
        self.out_dim = out_dim
        check(lib.tcvn_densenet_create(C.byref(cfg), C.byref(self.handle)), "densenet_create")
        self._n = 0

    def workspace_bytes(self, n_img: int, with_backward: bool) -> int:
        return lib.tcvn_densenet_workspace_bytes(self.handle, n_img, int(with_backward))

    def forward(self, coords: torch.Tensor, values: torch.Tensor, n_img: int, out: torch.Tensor, train: bool, seed: int = 0,
                log_pixels: bool = False, noise_std: float = 0.0):
        """coords int32 [nnz,3], values fp32 [nnz,C]; out: fp32 2-d view with row stride out.stride(0)."""
        assert coords.dtype == torch.int32 and coords.is_contiguous() and values.dtype == torch.float32 and values.is_contiguous()
        assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape == (n_img, self.out_dim)
        ws = self.workspace(self.workspace_bytes(n_img, train), coords.device)
        self._n = n_img
        check(lib.tcvn_densenet_forward(self.handle, n_img, _ptr(coords), _ptr(values), coords.shape[0], int(log_pixels),
                                        float(noise_std), _ptr(out), out.stride(0), _ptr(ws), ws.numel(), int(train),
                                        C.c_uint64(seed), _stream_ptr()), "densenet_forward")

    def backward(self, d_out: torch.Tensor):
        assert d_out.dtype == torch.float32 and d_out.stride(1) == 1 and d_out.shape == (self._n, self.out_dim)
        ws = self._ws
        check(lib.tcvn_densenet_backward(self.handle, self._n, _ptr(d_out), d_out.stride(0), _ptr(ws), ws.numel(),
                                         _stream_ptr()), "densenet_backward")

    def tap(self, name: str) -> torch.Tensor:
        """NHWC view [n,h,w,c] of an intermediate of the last forward (validation only)."""
        off, n, h, w, c, ld, es = C.c_int64(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(lib.tcvn_densenet_tap(self.handle, self._n, name.encode(), C.byref(off), C.byref(n), C.byref(h), C.byref(w),
                                    C.byref(c), C.byref(ld), C.byref(es)), f"tap {name}")
        dt = torch.float32 if es.value == 4 else torch.bfloat16
        raw = self._ws[off.value: off.value + n.value * h.value * w.value * ld.value * es.value].view(dt)
        return raw.view(n.value, h.value, w.value, ld.value)[..., :c.value]
