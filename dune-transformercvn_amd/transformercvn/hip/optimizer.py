"""Fused AdamW + global-norm gradient clip over the runtime's flat arenas.

Reference: ``NeutrinoBase.configure_optimizers`` (transformercvn/network/trainers/neutrino_base.py:88-152) builds
``torch.optim.AdamW`` with two parameter groups and Lightning clips with ``gradient_clip_val`` (train.py:140).  Stock torch
walks 782 small tensors per step; here every parameter is a view of one arena (``HipRuntime.ensure_bound``), so a step is the
two launches of ``tcvn_grad_sumsq`` + ``tcvn_adamw_step`` (include/tcvn_hip.h).  The class is a ``torch.optim.Optimizer`` with
the reference's two ``param_groups``, so ``LambdaLR`` schedules and Lightning drive it unchanged."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable

import torch

from . import _lib
from ._lib import check, lib


def _ptr(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, groups, runtime, names: Dict[int, str], lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 clip: float = 0.0, frozen: Iterable[str] = ()):
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=0.0))
        runtime.ensure_bound()
        self.rt = runtime
        self.clip = float(clip or 0.0)
        dev = runtime.flat_param.device
        n = runtime.flat_param.numel()
        wd = torch.full((n,), -1.0, dtype=torch.float32, device=dev)         # < 0: the reference's optimizer never touches it
        frozen = tuple(frozen)
        for g in self.param_groups:
            for p in g["params"]:
                name = names.get(id(p), "")
                if not p.requires_grad or name not in runtime.offsets or any(f in name for f in frozen):
                    continue
                off, k = runtime.offsets[name]
                wd[off:off + k] = float(g["weight_decay"])
        self.weight_decay = wd
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._partials = torch.zeros(1024, dtype=torch.float64, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._step = 0
        self._arena = runtime.flat_param.data_ptr()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rt = self.rt
        if rt.flat_param.data_ptr() != self._arena:
            raise RuntimeError("FlatAdamW: the parameter arena was rebuilt (module moved or re-created); build a new optimizer")
        rt._reattach_grads()                           # grads set to None by zero_grad(set_to_none=True) keep living in the arena
        lrs = {float(g["lr"]) for g in self.param_groups}
        if len(lrs) != 1:
            raise RuntimeError("FlatAdamW expects one learning rate for both parameter groups (reference: one LambdaLR for both)")
        g0 = self.param_groups[0]
        self._step += 1
        st = C.c_void_p(torch.cuda.current_stream(rt.flat_param.device).cuda_stream)
        n = rt.flat_param.numel()
        gss = None
        if self.clip > 0:
            check(lib.tcvn_grad_sumsq(_ptr(rt.flat_grad), n, _ptr(self._partials), 1024, _ptr(self._sumsq), st), "grad_sumsq")
            gss = _ptr(self._sumsq)
        check(lib.tcvn_adamw_step(_ptr(rt.flat_param), _ptr(rt.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                  _ptr(self.weight_decay), n, lrs.pop(), float(g0["betas"][0]), float(g0["betas"][1]),
                                  float(g0["eps"]), self._step, gss, self.clip, st), "adamw_step")
        return loss

    def grad_norm(self) -> torch.Tensor:
        """2-norm of the whole gradient arena as of the last clipped step (0-d device tensor)."""
        return self._sumsq.sqrt().squeeze(0)

    # checkpoints: the moments live in two flat tensors instead of per-parameter state
    def state_dict(self):
        d = super().state_dict()
        d["flat"] = {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self._step}
        return d

    def load_state_dict(self, state_dict):
        flat = state_dict.get("flat")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "flat"})
        if flat is not None:
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self._step = int(flat["step"])
