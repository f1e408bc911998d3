"""Host-side runtime of the fused network step.

Owns (a) the flat parameter / gradient / buffer arenas the ``nn.Parameter`` objects are views of -- one contiguous fp32
gradient buffer is what the data-parallel all-reduce works on --, (b) the native plans (two DenseNet engines + the token
path engine) bound to those views, and (c) the autograd glue: one ``torch.autograd.Function`` whose backward launches the
HIP backward and accumulates straight into the arena (parameter gradients never travel through autograd).

PyTorch supplies device memory, the current stream and (optionally) torch.distributed; all arithmetic is in libtcvn_hip.so.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _lib
from .engine import DenseNetEngine, HeadEngine
from .pixels import SparsePixels
from ..network.layers.packed_data import token_rows

PRECISIONS = {"fp32": _lib.MODE_F32, "f32": _lib.MODE_F32, "32": _lib.MODE_F32, "bf16": _lib.MODE_BF16, "16": _lib.MODE_BF16}


class _FusedStep(torch.autograd.Function):
    """(anchor) -> (event_logits, prong_logits).  The anchor is a dummy leaf that keeps the node in the graph."""

    @staticmethod
    def forward(ctx, anchor: Tensor, runtime: "HipRuntime", state: dict):
        ctx.runtime, ctx.state, ctx.anchor = runtime, state, anchor
        ev, pr = state["event_logits"], state["prong_logits"]
        return ev, pr

    @staticmethod
    def backward(ctx, d_ev: Tensor, d_pr: Tensor):
        ctx.runtime._backward(ctx.state, d_ev, d_pr)
        return torch.zeros_like(ctx.anchor), None, None      # a real (zero) gradient: a DDP wrapper's hook on the anchor must fire


class _FocalLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ev: Tensor, pr: Tensor, runtime: "HipRuntime", event_targets: Tensor, prong_targets: Tensor):
        losses, accs, d_ev, d_pr = runtime.head.loss(ev.contiguous(), pr.contiguous(), event_targets, prong_targets)
        ctx.save_for_backward(d_ev, d_pr)
        out = (losses[0], losses[1], losses[2], accs[0], accs[1])
        ctx.mark_non_differentiable(*out[1:])
        return out

    @staticmethod
    def backward(ctx, g_total, *_):
        d_ev, d_pr = ctx.saved_tensors
        return d_ev * g_total, d_pr * g_total, None, None, None


class HipRuntime:
    def __init__(self, network: nn.Module, options, pixel_shape: Tuple[int, int], precision: str = "fp32", seed: int = 0):
        self.network = network
        self.options = options
        self.mode = PRECISIONS[str(precision).lower()]
        self.pixel_shape = tuple(pixel_shape)
        self.seed = seed
        self.step = 0
        pe = network.prong_embedding
        self.smart_features = not bool(options.disable_smart_features)     # a8: ProngFeatureEmbedding MLP on the row kernels
        self.feature_mlp = None
        H, W = self.pixel_shape
        self.ev_engine = pe.event_pixel_embedding.hip_engine(self.mode, H, W)
        self.pr_engine = pe.prong_pixel_embedding.hip_engine(self.mode, H, W)
        dec = network.prong_decoder
        self.head = HeadEngine(options.hidden_dim, options.num_attention_heads, options.num_encoder_layers,
                               pe.feature_embedding_dim + pe.pixel_embedding_dim + pe.position_embedding_dim,
                               network.event_decoder.hidden_layer.out_features, dec.output_dim, dec.widths,
                               dec.output_layer.in_features, options.transformer_activation == "gelu",
                               bool(options.transformer_norm_first), float(options.dropout), float(options.loss_gamma),
                               float(options.event_prong_loss_proportion), bool(options.linear_batch_norm),
                               bool(options.linear_prelu_activation))
        self.anchor: Optional[Tensor] = None
        self.flat_param = self.flat_grad = self.flat_buf = None
        self._sig = None
        self._grad_views: List[Tensor] = []
        self.overlap_embedders = True        # event DenseNet on a side stream underneath the prong DenseNet
        self.side_priority = 0               # stream priority of that side stream (0 = default, -1 = high)
        self.anchor_param = None             # optional: module-owned anchor parameter (see NeutrinoFullBaseTrainer)
        self.grad_ready_hook = None          # called as hook(tag) when a gradient segment is final ("head", "event", "prong")
        self.segments: Dict[str, Tuple[int, int]] = {}
        self.offsets: Dict[str, Tuple[int, int]] = {}

    # ---------------------------------------------------------------------------------------------------------------
    # flat arenas
    # ---------------------------------------------------------------------------------------------------------------
    def _needs_rebind(self) -> bool:
        ps = [p for p in self.network.parameters()]
        if self.flat_param is None or not ps:
            return True
        first, last = ps[0], ps[-1]
        return (first.device != self.flat_param.device or first.data_ptr() != self._sig[0] or last.data_ptr() != self._sig[1])

    def ensure_bound(self):
        if not self._needs_rebind():
            return
        net = self.network
        params = [(n, p) for n, p in net.named_parameters()]
        dev = params[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("transformercvn (MI355X build): parameters must live on the GPU; there is no CPU fallback")
        total = sum(p.numel() for _, p in params)
        flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self._grad_views = []
        offsets = {}
        for n, p in params:
            k = p.numel()
            flat_p[off:off + k].copy_(p.detach().reshape(-1).float())
            p.data = flat_p[off:off + k].view(p.shape)
            gv = flat_g[off:off + k].view(p.shape)
            if p.requires_grad:
                p.grad = gv
            self._grad_views.append(gv)
            offsets[n] = (off, k)
            off += k
        bufs = [(n, b) for n, b in net.named_buffers() if b.is_floating_point()]
        flat_b = torch.empty(sum(b.numel() for _, b in bufs), dtype=torch.float32, device=dev)
        off = 0
        for n, b in bufs:
            k = b.numel()
            flat_b[off:off + k].copy_(b.reshape(-1).float())
            b.data = flat_b[off:off + k].view(b.shape)
            off += k
        self.flat_param, self.flat_grad, self.flat_buf = flat_p, flat_g, flat_b
        self.offsets = offsets                        # name (relative to the network) -> (offset, numel) in the arenas
        ps = [p for _, p in params]
        self._sig = (ps[0].data_ptr(), ps[-1].data_ptr())
        self._params = ps
        self._param_grads = {p: g for p, g in zip(ps, self._grad_views)}       # parameter object -> its gradient view in the arena
        # gradient segments for overlapped all-reduce: parameters are laid out in registration order
        def span(prefix):
            keys = [k for k in offsets if k.startswith(prefix)]
            lo = min(offsets[k][0] for k in keys)
            hi = max(offsets[k][0] + offsets[k][1] for k in keys)
            return lo, hi
        self.segments = {"event": span("prong_embedding.event_pixel_embedding.")}
        pfx = "prong_embedding.prong_pixel_embedding."
        parts = getattr(self.pr_engine, "n_parts", 0)
        if parts > 1:      # the prong embedder's backward is issued block by block: one exchange segment per dense block
            for part in range(parts):
                keys = [k for k in offsets if any(k.startswith(pfx + q) for q in self.pr_engine.part_prefixes(part))]
                self.segments[f"prong{part}"] = (min(offsets[k][0] for k in keys), max(offsets[k][0] + offsets[k][1] for k in keys))
            covered = sum(hi - lo for t, (lo, hi) in self.segments.items() if t.startswith("prong"))
            assert covered == span(pfx)[1] - span(pfx)[0], "prong embedder segments must tile its parameter span"
        else:
            self.segments["prong"] = span(pfx)
        # bind the native plans to the views
        named_p = dict(net.named_parameters())
        named_b = {n: b for n, b in net.named_buffers() if b.is_floating_point()}
        grads = {n: g for (n, _), g in zip(params, self._grad_views)}

        def sub(prefix):
            d = {k[len(prefix):]: v.detach() for k, v in named_p.items() if k.startswith(prefix)}
            d.update({k[len(prefix):]: v for k, v in named_b.items() if k.startswith(prefix)})
            g = {k[len(prefix):]: v for k, v in grads.items() if k.startswith(prefix)}
            return d, g
        d, g = sub("prong_embedding.event_pixel_embedding.")
        self.ev_engine.bind(d, g)
        d, g = sub("prong_embedding.prong_pixel_embedding.")
        self.pr_engine.bind(d, g)
        d, g = sub("")
        self.head.bind(d, g)
        self.anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._pos = named_p["prong_embedding.event_position_embedding"]
        self._pos_grad = grads["prong_embedding.event_position_embedding"]
        pe = net.prong_embedding
        ran = [m for mod in (pe.event_pixel_embedding, pe.prong_pixel_embedding, pe.combined_embedding, net.prong_decoder)
               for m in mod.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
        # num_batches_tracked of every BatchNorm become views of one int64 arena: one add per step instead of 139
        all_bn = [m for m in net.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
        self._side = torch.cuda.Stream(dev, priority=self.side_priority)     # event-embedder stream (forward/_backward)
        self.flat_nbt = torch.stack([m.num_batches_tracked.to(dev) for m in all_bn]).contiguous()
        ran_ids = {id(m) for m in ran}
        self._nbt_inc = torch.tensor([1 if id(m) in ran_ids else 0 for m in all_bn], dtype=torch.int64, device=dev)
        dec_ids = {id(m) for m in net.prong_decoder.modules()}
        self._nbt_inc_embed = torch.tensor([1 if (id(m) in ran_ids and id(m) not in dec_ids) else 0 for m in all_bn],
                                           dtype=torch.int64, device=dev)
        for i, m in enumerate(all_bn):
            m.num_batches_tracked.data = self.flat_nbt[i]

    def _mlp(self):
        if self.feature_mlp is None:
            from .rowops import FeatureMLP
            self.feature_mlp = FeatureMLP(self.network.prong_embedding.feature_embedding.embedding)
        return self.feature_mlp

    def zero_grad(self):
        """Zero the gradient arena in one memset and (re)attach the per-parameter views."""
        self.ensure_bound()
        self.flat_grad.zero_()
        for p, gv in zip(self._params, self._grad_views):
            if p.requires_grad and p.grad is not gv:
                p.grad = gv

    def _reattach_grads(self):
        stale = False
        for p, gv in zip(self._params, self._grad_views):
            if p.requires_grad and p.grad is not gv:
                if p.grad is not None:
                    gv.add_(p.grad)            # someone accumulated into a foreign tensor: fold it in
                else:
                    stale = True
                p.grad = gv
        return stale

    # ---------------------------------------------------------------------------------------------------------------
    # forward / backward
    # ---------------------------------------------------------------------------------------------------------------
    def forward(self, features: Tensor, extra: Tensor, event_px: SparsePixels, event_mask: Tensor, prong_px: SparsePixels,
                prong_mask: Tensor, counts: Optional[Tuple[int, int]] = None) -> Tuple[Tensor, Tensor]:
        self.ensure_bound()
        net, opt = self.network, self.options
        pe = net.prong_embedding
        dev = self.flat_param.device
        training = net.training
        B, P = prong_mask.shape
        n_prongs = int(counts[1]) if counts is not None else int(prong_mask.sum().item())
        prong_mask = prong_mask.to(dev)
        event_px.count, prong_px.count = B, n_prongs
        tok_row = token_rows(prong_mask, B)
        feat, pix, pos = pe.feature_embedding_dim, pe.pixel_embedding_dim, pe.position_embedding_dim
        in_dim = feat + pix + pos
        seed = (self.seed * 1000003 + self.step) & 0x7FFFFFFFFFFFFFFF
        self.step += 1
        # seeds of the three native plans for this step (validation: tcvn_dropout_keep replays the masks from them)
        self.last_seeds = {"event": seed ^ 0x1111, "prong": seed ^ 0x2222, "head": seed ^ 0x3333}
        with torch.no_grad():
            rows = torch.zeros(B + n_prongs, in_dim, device=dev)
            rows[:, feat + pix:] = self._pos            # prongs also get the *event* position embedding (reference quirk)
            if self.smart_features:                     # layers/prong_feature_embedding.py:73-78: MLP over [features | extra[event]]
                i1, i2 = prong_mask.nonzero(as_tuple=True)
                fin = torch.cat((features.to(dev)[i1, i2], extra.to(dev)[i1]), dim=1)
                rows[B:, :feat] = self._mlp().forward(fin, training, seed ^ 0x4444)
            # the two embedders are independent until the token path: the small event DenseNet (B images) runs on a side
            # stream underneath the prong DenseNet (n_prongs images), whose launches alone do not fill the chip in the deep blocks
            main = torch.cuda.current_stream(dev)
            side = self._side if self.overlap_embedders else main
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self.ev_engine.forward(event_px.coords, event_px.values, B, rows[:B, :feat + pix], training, seed ^ 0x1111,
                                       event_px.value_mode, event_px.noise_std if training else 0.0)
            self.pr_engine.forward(prong_px.coords, prong_px.values, n_prongs, rows[B:, feat:feat + pix], training,
                                   seed ^ 0x2222, prong_px.value_mode, prong_px.noise_std if training else 0.0)
            main.wait_stream(side)
            ev, pr = self.head.forward(rows, tok_row, B, P, n_prongs, training, seed ^ 0x3333)
            if training:
                self.flat_nbt += self._nbt_inc
        if not (training and torch.is_grad_enabled()):
            return ev, pr
        state = dict(rows=rows, tok_row=tok_row, B=B, P=P, n_prongs=n_prongs, event_logits=ev, prong_logits=pr,
                     feat=feat, pix=pix, keep=(event_px, prong_px))      # the COO lists are read again by backward
        anchor = self.anchor_param if (self.anchor_param is not None and self.anchor_param.device == dev) else self.anchor
        return _FusedStep.apply(anchor, self, state)

    # ---------------------------------------------------------------------------------------------------------------
    # stage-by-stage forward (the reference's sub-module call surface; forward only)
    # ---------------------------------------------------------------------------------------------------------------
    def _rows(self, event_px: SparsePixels, prong_px: SparsePixels, B: int, n_prongs: int, training: bool, seed: int) -> Tensor:
        """[B + n_prongs, feat+pix+pos] input rows of the combined embedding from the two DenseNet engines."""
        pe = self.network.prong_embedding
        dev = self.flat_param.device
        feat, pix, pos = pe.feature_embedding_dim, pe.pixel_embedding_dim, pe.position_embedding_dim
        rows = torch.zeros(B + n_prongs, feat + pix + pos, device=dev)
        rows[:, feat + pix:] = self._pos
        event_px.count, prong_px.count = B, n_prongs
        self.ev_engine.forward(event_px.coords, event_px.values, B, rows[:B, :feat + pix], training, seed ^ 0x1111,
                               event_px.value_mode, event_px.noise_std if training else 0.0)
        self.pr_engine.forward(prong_px.coords, prong_px.values, n_prongs, rows[B:, feat:feat + pix], training, seed ^ 0x2222,
                               prong_px.value_mode, prong_px.noise_std if training else 0.0)
        return rows

    def embed(self, features: Tensor, extra: Tensor, event_px: SparsePixels, event_mask: Tensor, prong_px: SparsePixels,
              prong_mask: Tensor, training: bool = False) -> Tensor:
        """BaseProngEmbedding.forward: -> tokens [B, 1+P, hidden] (padding rows zero)."""
        self.ensure_bound()
        with torch.no_grad():
            dev = self.flat_param.device
            prong_mask = prong_mask.to(dev)
            B, P = prong_mask.shape
            n_prongs = int(prong_mask.sum().item())
            seed = (self.seed * 1000003 + self.step) & 0x7FFFFFFFFFFFFFFF
            self.step += 1
            rows = self._rows(event_px, prong_px, B, n_prongs, training, seed)
            if self.smart_features:
                pe = self.network.prong_embedding
                i1, i2 = prong_mask.nonzero(as_tuple=True)
                fin = torch.cat((features.to(dev)[i1, i2], extra.to(dev)[i1]), dim=1)
                rows[B:, :pe.feature_embedding_dim] = self._mlp().forward(fin, training, seed ^ 0x4444)
            if training:
                self.flat_nbt += self._nbt_inc_embed
            return self.head.embed(rows, token_rows(prong_mask, B), B, P, n_prongs, training, seed ^ 0x3333)

    def encode(self, tokens: Tensor, mask: Tensor, training: bool = False) -> Tensor:
        """ProngCustomBertEncoder.forward: tokens [B, S, hidden], mask [B, S] -> hidden [S, B, hidden] (masked)."""
        self.ensure_bound()
        with torch.no_grad():
            dev = self.flat_param.device
            if not tokens.is_cuda:
                raise RuntimeError("transformercvn (MI355X build): the encoder runs on the GPU only; there is no CPU fallback")
            tok = torch.where(mask.to(dev), 0, -1).to(torch.int32).contiguous()
            seed = (self.seed * 1000003 + self.step) & 0x7FFFFFFFFFFFFFFF
            self.step += 1
            return self.head.encode(tokens.detach().float().contiguous(), tok, training, seed ^ 0x3333)

    def _backward(self, st: dict, d_ev: Tensor, d_pr: Tensor):
        """Backward of the fused step in the order the gradient segments become final -- token path, event embedder (side
        stream), prong embedder -- reporting each to ``grad_ready_hook`` so that its all-reduce overlaps with what is left."""
        self._reattach_grads()                       # grads set to None by zero_grad(set_to_none=True): views re-attached
        B, feat, pix = st["B"], st["feat"], st["pix"]
        hook = self.grad_ready_hook or (lambda tag: None)
        d_rows = self.head.backward(st["rows"], st["tok_row"], d_ev.contiguous(), d_pr.contiguous())
        self._pos_grad.add_(d_rows[:, feat + pix:].sum(0, keepdim=True))
        if self.smart_features:
            self._mlp().backward(d_rows[B:, :feat], self._param_grads)
        hook("head")
        d_pr_rows = d_rows[B:, feat:feat + pix]
        if not d_rows.is_cuda:                       # CPU stand-ins (tests of the exchange schedule): one queue
            self.ev_engine.backward(d_rows[:B, :feat + pix])
            hook("event")
            self._prong_backward(d_pr_rows, hook)
            return
        main = torch.cuda.current_stream(d_rows.device)
        side = self._side if self.overlap_embedders else main
        side.wait_stream(main)
        with torch.cuda.stream(side):                # event embedder backward underneath the prong embedder's (see forward)
            self.ev_engine.backward(d_rows[:B, :feat + pix])
            hook("event")                            # issued from the side stream: the collective is ordered behind the event backward
        self._prong_backward(d_pr_rows, hook)
        main.wait_stream(side)

    def _prong_backward(self, d_pr_rows: Tensor, hook):
        """Prong embedder backward.  Data parallel (a hook is installed) and an engine that can be driven block by block: dense
        blocks 5 -> 1, each block's gradient slice goes to the exchange while the earlier blocks still run, so only block 1 + stem
        (the last slice) is exposed.  Otherwise one call."""
        parts = getattr(self.pr_engine, "n_parts", 0)
        if parts > 1 and self.grad_ready_hook is not None:
            for part in range(parts - 1, -1, -1):
                self.pr_engine.backward_part(d_pr_rows, part)
                hook(f"prong{part}")
        else:
            self.pr_engine.backward(d_pr_rows)
            if parts > 1:
                for part in range(parts - 1, -1, -1):
                    hook(f"prong{part}")
            else:
                hook("prong")

    def loss(self, ev: Tensor, pr: Tensor, event_targets: Tensor, prong_targets: Tensor):
        """-> (total, event_loss, prong_loss, event_accuracy, prong_accuracy) as 0-d device tensors; total is differentiable."""
        dev = ev.device
        et = event_targets.to(dev, torch.int64).contiguous()
        pt = prong_targets.to(dev, torch.int8).contiguous()
        return _FocalLoss.apply(ev, pr, self, et, pt)
