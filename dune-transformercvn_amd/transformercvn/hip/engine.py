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


class _EmbedderEngine(_Plan):
    """Shared driver of the two pixel-map embedder plans (tcvn_densenet_* / tcvn_sdxl_*: same calling convention)."""
    out_dim = 0

    def __init__(self):
        super().__init__()
        self._n = 0

    def workspace_bytes(self, n_img: int, with_backward: bool) -> int:
        return self._fn("workspace_bytes")(self.handle, n_img, int(with_backward))

    def forward(self, coords: torch.Tensor, values: torch.Tensor, n_img: int, out: torch.Tensor, train: bool, seed: int = 0,
                log_pixels: bool = False, noise_std: float = 0.0):
        """coords int32 [nnz,3], values fp32 [nnz,C]; out: fp32 2-d view with row stride out.stride(0)."""
        assert coords.dtype == torch.int32 and coords.is_contiguous() and values.dtype == torch.float32 and values.is_contiguous()
        assert out.dtype == torch.float32 and out.stride(1) == 1 and out.shape == (n_img, self.out_dim)
        ws = self.workspace(self.workspace_bytes(n_img, train), coords.device)
        self._n = n_img
        self._inputs = (coords, values)          # backward reads the COO list again (sparse stem weight gradient)
        check(self._fn("forward")(self.handle, n_img, _ptr(coords), _ptr(values), coords.shape[0], int(log_pixels),
                                  float(noise_std), _ptr(out), out.stride(0), _ptr(ws), ws.numel(), int(train),
                                  C.c_uint64(seed), _stream_ptr()), f"{self._prefix}_forward")

    def backward(self, d_out: torch.Tensor):
        assert d_out.dtype == torch.float32 and d_out.stride(1) == 1 and d_out.shape == (self._n, self.out_dim)
        ws = self._ws
        check(self._fn("backward")(self.handle, self._n, _ptr(d_out), d_out.stride(0), _ptr(ws), ws.numel(), _stream_ptr()),
              f"{self._prefix}_backward")

    def tap(self, name: str) -> torch.Tensor:
        """NHWC view [n,h,w,c] of an intermediate of the last forward (validation only)."""
        off, n, h, w, c, ld, es = C.c_int64(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        check(self._fn("tap")(self.handle, self._n, name.encode(), C.byref(off), C.byref(n), C.byref(h), C.byref(w),
                              C.byref(c), C.byref(ld), C.byref(es)), f"tap {name}")
        dt = {8: torch.float64, 4: torch.float32}.get(es.value, torch.bfloat16)
        raw = self._ws[off.value: off.value + n.value * h.value * w.value * ld.value * es.value].view(dt)
        return raw.view(n.value, h.value, w.value, ld.value)[..., :c.value]


class DenseNetEngine(_EmbedderEngine):
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
        self.n_parts = lib.tcvn_densenet_num_blocks(self.handle)       # backward can be issued block by block (last block first)

    def backward_part(self, d_out: torch.Tensor, part: int):
        """Backward of dense block `part` alone (call with part = n_parts-1 ... 0): tcvn_densenet_backward_blocks."""
        assert d_out.dtype == torch.float32 and d_out.stride(1) == 1 and d_out.shape == (self._n, self.out_dim)
        ws = self._ws
        check(lib.tcvn_densenet_backward_blocks(self.handle, self._n, _ptr(d_out), d_out.stride(0), _ptr(ws), ws.numel(), part, part,
                                                _stream_ptr()), "densenet_backward_blocks")

    def part_prefixes(self, part: int):
        """Parameter-name prefixes (relative to the DenseNet module) whose gradients are final after backward_part(part)."""
        pre = [f"features.dense{part + 1}.", f"features.transition{part + 1}."]
        if part == self.n_parts - 1:
            pre += ["features.final_norm.", "features.final_relu.", "output_block."]
        if part == 0:
            pre += ["features.conv0.", "features.norm0.", "features.relu0."]
        return pre


class SdxlEngine(_EmbedderEngine):
    """SDXL-style embedder plan (tcvn_sdxl_* in include/tcvn_hip.h; parity unpinned, see oracle/sdxl_oracle.py)."""
    _prefix = "sdxl"

    def __init__(self, in_ch: int, out_dim: int, init_ch: int, repeat: int, num_blocks: int, H: int, W: int, mode: int):
        super().__init__()
        cfg = _lib.SdxlCfg()
        cfg.in_ch, cfg.out_dim, cfg.init_ch, cfg.repeat, cfg.num_blocks = in_ch, out_dim, init_ch, repeat, num_blocks
        cfg.H, cfg.W, cfg.mode = H, W, mode
        self.cfg = cfg
        self.mode = mode
        self.out_dim = out_dim
        check(lib.tcvn_sdxl_create(C.byref(cfg), C.byref(self.handle)), "sdxl_create")


class HeadEngine(_Plan):
    """Combined embedding + transformer encoder + decoders + focal loss (tcvn_head_* in include/tcvn_hip.h)."""
    _prefix = "head"

    def __init__(self, hidden_dim: int, heads: int, n_layers: int, in_dim: int, event_classes: int, prong_classes: int,
                 dec_dims, dec_out_in: int, gelu: bool, norm_first: bool, dropout: float, gamma: float, event_weight: float,
                 linear_batch_norm: bool = True, linear_prelu_activation: bool = True):
        super().__init__()
        cfg = _lib.HeadCfg()
        cfg.hidden_dim, cfg.heads, cfg.n_layers, cfg.in_dim = hidden_dim, heads, n_layers, in_dim
        cfg.event_classes, cfg.prong_classes = event_classes, prong_classes
        cfg.n_dec = len(dec_dims)
        for i, d in enumerate(dec_dims):
            cfg.dec_dims[i] = d
        cfg.dec_out_in, cfg.gelu, cfg.norm_first = dec_out_in, int(gelu), int(norm_first)
        cfg.dropout_modules = int(dropout > 0.0)
        cfg.dropout, cfg.gamma, cfg.event_weight = dropout, gamma, event_weight
        # LinearBlock option variants (reference layers/prong_feature_embedding.py:11-21, layers/encoder.py:13-19)
        cfg.no_linear_bn, cfg.linear_relu = int(not linear_batch_norm), int(not linear_prelu_activation)
        self.cfg = cfg
        check(lib.tcvn_head_create(C.byref(cfg), C.byref(self.handle)), "head_create")
        self._shape = (0, 0, 0)

    def forward(self, rows: torch.Tensor, tok_row: torch.Tensor, batch: int, max_prongs: int, n_prongs: int, train: bool,
                seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        assert rows.dtype == torch.float32 and rows.is_contiguous() and rows.shape == (batch + n_prongs, self.cfg.in_dim)
        assert tok_row.dtype == torch.int32 and tok_row.is_contiguous() and tok_row.shape == (batch, 1 + max_prongs)
        ws = self.workspace(lib.tcvn_head_workspace_bytes(self.handle, batch, max_prongs, n_prongs), rows.device)
        ev = torch.empty(batch, self.cfg.event_classes, device=rows.device)
        pr = torch.empty(batch, max_prongs, self.cfg.prong_classes, device=rows.device)
        self._shape = (batch, max_prongs, n_prongs)
        check(lib.tcvn_head_forward(self.handle, batch, max_prongs, n_prongs, _ptr(rows), _ptr(tok_row), _ptr(ev), _ptr(pr),
                                    _ptr(ws), ws.numel(), int(train), C.c_uint64(seed), _stream_ptr()), "head_forward")
        return ev, pr

    def _stage_ws(self, batch: int, max_prongs: int, n_prongs: int, device):
        return self.workspace(lib.tcvn_head_workspace_bytes(self.handle, batch, max_prongs, n_prongs), device)

    def embed(self, rows: torch.Tensor, tok_row: torch.Tensor, batch: int, max_prongs: int, n_prongs: int, train: bool,
              seed: int = 0) -> torch.Tensor:
        """tcvn_head_embed: -> tokens [batch, 1+max_prongs, hidden] (forward only)."""
        assert rows.dtype == torch.float32 and rows.is_contiguous() and rows.shape == (batch + n_prongs, self.cfg.in_dim)
        assert tok_row.dtype == torch.int32 and tok_row.is_contiguous() and tok_row.shape == (batch, 1 + max_prongs)
        ws = self._stage_ws(batch, max_prongs, n_prongs, rows.device)
        tokens = torch.empty(batch, 1 + max_prongs, self.cfg.hidden_dim, device=rows.device)
        check(lib.tcvn_head_embed(self.handle, batch, max_prongs, n_prongs, _ptr(rows), _ptr(tok_row), _ptr(tokens), _ptr(ws),
                                  ws.numel(), int(train), C.c_uint64(seed), _stream_ptr()), "head_embed")
        return tokens

    def encode(self, tokens: torch.Tensor, tok_row: torch.Tensor, train: bool, seed: int = 0) -> torch.Tensor:
        """tcvn_head_encode: tokens [B, S, hidden] -> hidden [S, B, hidden] (forward only)."""
        batch, S, D = tokens.shape
        assert D == self.cfg.hidden_dim and tokens.dtype == torch.float32 and tokens.is_contiguous()
        assert tok_row.dtype == torch.int32 and tok_row.is_contiguous() and tok_row.shape == (batch, S)
        ws = self._stage_ws(batch, S - 1, 0, tokens.device)
        hidden = torch.empty(S, batch, D, device=tokens.device)
        check(lib.tcvn_head_encode(self.handle, batch, S - 1, _ptr(tokens), _ptr(tok_row), _ptr(hidden), _ptr(ws), ws.numel(),
                                   int(train), C.c_uint64(seed), _stream_ptr()), "head_encode")
        return hidden

    def decode(self, hidden: torch.Tensor, train: bool = False, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
        """tcvn_head_decode: hidden [S, B, hidden] -> (event_logits [B, Ce], prong_logits [B, S-1, Cp]) (forward only)."""
        S, batch, D = hidden.shape
        assert D == self.cfg.hidden_dim and hidden.dtype == torch.float32 and hidden.is_contiguous()
        ws = self._stage_ws(batch, S - 1, 0, hidden.device)
        ev = torch.empty(batch, self.cfg.event_classes, device=hidden.device)
        pr = torch.empty(batch, S - 1, self.cfg.prong_classes, device=hidden.device)
        check(lib.tcvn_head_decode(self.handle, batch, S - 1, _ptr(hidden), _ptr(ev), _ptr(pr), _ptr(ws), ws.numel(), int(train),
                                   C.c_uint64(seed), _stream_ptr()), "head_decode")
        return ev, pr

    def loss(self, ev: torch.Tensor, pr: torch.Tensor, event_targets: torch.Tensor, prong_targets: torch.Tensor):
        """-> (losses[3] = total/event/prong, accs[2], d_event_logits, d_prong_logits), all on the device."""
        batch, max_prongs = pr.shape[0], pr.shape[1]
        assert ev.is_contiguous() and pr.is_contiguous() and ev.dtype == torch.float32 and pr.dtype == torch.float32
        assert event_targets.dtype == torch.int64 and prong_targets.dtype == torch.int8
        assert event_targets.is_contiguous() and prong_targets.is_contiguous() and prong_targets.shape == (batch, max_prongs)
        losses = torch.empty(3, device=ev.device)
        accs = torch.empty(6, device=ev.device)
        d_ev, d_pr = torch.empty_like(ev), torch.empty_like(pr)
        check(lib.tcvn_head_loss(self.handle, batch, max_prongs, _ptr(ev), _ptr(pr), _ptr(event_targets), _ptr(prong_targets),
                                 _ptr(losses), _ptr(accs), _ptr(d_ev), _ptr(d_pr), _stream_ptr()), "head_loss")
        return losses, accs[:2], d_ev, d_pr

    def backward(self, rows: torch.Tensor, tok_row: torch.Tensor, d_ev: torch.Tensor, d_pr: torch.Tensor) -> torch.Tensor:
        batch, max_prongs, n_prongs = self._shape
        assert d_ev.is_contiguous() and d_pr.is_contiguous() and d_ev.dtype == torch.float32 and d_pr.dtype == torch.float32
        d_rows = torch.empty_like(rows)
        ws = self._ws
        check(lib.tcvn_head_backward(self.handle, batch, max_prongs, n_prongs, _ptr(rows), _ptr(tok_row), _ptr(d_ev), _ptr(d_pr),
                                     _ptr(d_rows), _ptr(ws), ws.numel(), _stream_ptr()), "head_backward")
        return d_rows


class _FocalRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, targets: torch.Tensor, gamma: float):
        lg = logits.contiguous().float()
        tg = targets.to(lg.device, torch.int64).contiguous()
        d = torch.empty_like(lg)
        out = torch.empty(2, device=lg.device)
        check(lib.tcvn_focal_loss(_ptr(lg), _ptr(tg), lg.shape[0], lg.shape[1], float(gamma), 1.0, _ptr(d), _ptr(out),
                                  _stream_ptr()), "focal_loss")
        ctx.save_for_backward(d)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None


def focal_rows(logits: torch.Tensor, targets: torch.Tensor, gamma: float) -> torch.Tensor:
    """mean_i(-log p_t (1-p_t)^gamma) over the rows of one logit matrix, on the HIP focal kernel."""
    if not logits.is_cuda:
        raise RuntimeError("transformercvn (MI355X build): the focal loss runs on the GPU only")
    return _FocalRows.apply(logits, targets, gamma)
