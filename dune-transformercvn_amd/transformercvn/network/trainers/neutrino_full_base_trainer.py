"""Lightning-module level of the hot path (reference: trainers/neutrino_full_base_trainer.py:20-230): feature
normalisation, pixel preprocessing, network call, softmax focal loss, training/validation steps."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Tuple

import torch
from torch import Tensor

from transformercvn.options import Options
from transformercvn.network.trainers.neutrino_base import NeutrinoBase
from transformercvn.dataset.minkowski_dataset import MinkowskiDataset, MinkowskiCollection
from transformercvn.network.trainers.metrics import make_metrics


def _hide_anchor(module, state_dict, prefix, local_metadata):
    state_dict.pop(prefix + "_ddp_anchor", None)
    return state_dict


def _supply_anchor(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    state_dict.setdefault(prefix + "_ddp_anchor", torch.zeros(1))


class NeutrinoFullBaseTrainer(NeutrinoBase, ABC):
    @abstractmethod
    def create_network(self, options: Options, features_dim: int, extra_dim: int, pixel_dim: int, num_prong_classes: int,
                       num_event_classes: int):
        raise NotImplementedError()

    @abstractmethod
    def preprocess_pixels(self, pixel_coords: Tensor, pixel_values: Tensor, image_size: Tuple[int, ...]):
        raise NotImplementedError()

    def __init__(self, options: Options):
        super().__init__(options)
        ds = self.training_dataset
        self.hidden_dim = options.hidden_dim
        self.network = self.create_network(options, ds.num_features, ds.num_extra, ds.pixel_features, ds.num_prong_classes,
                                           ds.num_event_classes)
        self.network.pixel_shape = tuple(ds.pixel_shape)
        self.beta = 1 - 1 / len(ds)
        self.gamma = options.loss_gamma
        self.event_loss_scale = options.event_prong_loss_proportion
        self.prong_loss_scale = 1.0 - options.event_prong_loss_proportion
        (self.event_accuracy, self.prong_accuracy, self.event_auc, self.prong_auc) = make_metrics(ds.num_event_classes,
                                                                                                 ds.num_prong_classes)
        self._init_data_parallel()

    # ---- data parallel (reference: train.py:123-127 wraps the module in Lightning's DDPStrategy) ---------------------
    def _init_data_parallel(self):
        """Parameter gradients never travel through autograd here (the HIP backward writes them into one flat arena), so torch
        DDP's per-parameter hooks would never fire.  Everything is therefore listed in `_ddp_params_and_buffers_to_ignore`
        (Lightning's wrapper forwards that list to DistributedDataParallel) except one hidden 1-element anchor parameter that
        does get an autograd gradient every step -- DDP needs at least one parameter to manage and a hook that fires.  The
        real exchange is the arena all-reduce of transformercvn.hip.distributed.GradReducer, installed by on_fit_start().
        The anchor is kept out of state_dict()/load_state_dict() so checkpoints stay the reference's 1 208 keys."""
        self._ddp_anchor = torch.nn.Parameter(torch.zeros(1))
        self._ddp_params_and_buffers_to_ignore = ([n for n, _ in self.named_parameters() if n != "_ddp_anchor"] +
                                                  [n for n, _ in self.named_buffers()])
        self._register_state_dict_hook(_hide_anchor)
        self._register_load_state_dict_pre_hook(_supply_anchor)
        self._reducer = None

    def enable_data_parallel(self, group=None):
        """Install the overlapped arena all-reduce (RCCL via torch.distributed) on the runtime's segment hooks.  Called by
        on_fit_start() under Lightning; custom loops (bench.py) call it after init_process_group.  No-op at world size 1."""
        from transformercvn.hip import distributed as hd
        from transformercvn.hip.distributed import GradReducer
        if not hd._active(group):
            return None
        rt = self.network.hip_runtime()
        rt.ensure_bound()
        # the ranks may have started from different random weights (the reference sets no seed and relies on DDP's initial
        # broadcast, which here covers the anchor only): rank 0's parameters, BatchNorm buffers and optimizer moments everywhere
        from transformercvn.hip.distributed import sync_state
        sync_state(rt, getattr(self, "_flat_optimizer", None), group)
        self._reducer = GradReducer(rt.flat_grad, rt.segments, group)
        rt.grad_ready_hook = self._reducer.on_ready
        self._dp_group = group
        return self._reducer

    # ---- precision (reference: train.py:141,172  `-fp16` -> pl.Trainer(precision=16 if fp16 else 32)) ------------------
    def adopt_trainer_precision(self):
        """Lightning precision 16 / "16-mixed" / "bf16" selects the bf16 MFMA engines (the throughput mode), 32 the fp32 parity
        engines -- unless the option file names `hip_precision` itself.  Lightning's AMP plugin may still wrap the step in
        autocast and scale the loss: all arithmetic here is in libtcvn_hip.so with explicit types, the scaled loss gradient
        scales the arena linearly, and GradScaler.unscale_ works on the arena views (tests/test_optimizer_gpu.py)."""
        if "hip_precision" in vars(self.options):
            return
        try:
            prec = getattr(self, "trainer", None)
            prec = None if prec is None else getattr(prec, "precision", None)
        except Exception:                                   # pl.LightningModule.trainer raises while unattached
            prec = None
        if prec is None:
            return
        want = "bf16" if str(prec).replace("-mixed", "").replace("-true", "") in ("16", "bf16") else "fp32"
        self.set_precision(want, from_options=False)

    def set_precision(self, precision: str, from_options: bool = True):
        from transformercvn.hip.runtime import PRECISIONS
        mode = PRECISIONS[str(precision).lower()]
        net = self.network
        if from_options:
            self.options.hip_precision = precision
        net._precision_override = precision
        if net._runtime is not None and net._runtime.mode != mode:
            if getattr(self, "_flat_optimizer", None) is not None:
                raise RuntimeError("precision change after configure_optimizers(): the parameter arena would be rebuilt under the optimizer")
            net._runtime = None                              # engines are rebuilt (and the arenas re-bound) on the next call

    def setup(self, stage=None):
        self.adopt_trainer_precision()

    def on_fit_start(self):
        self.adopt_trainer_precision()
        self.enable_data_parallel()

    def on_train_batch_start(self, batch, batch_idx, *args):
        if self._reducer is not None:                       # DDP's broadcast_buffers=True semantics: rank 0's BN statistics
            from transformercvn.hip.distributed import broadcast_buffers
            broadcast_buffers(self.network.hip_runtime().flat_buf, getattr(self, "_dp_group", None))

    def on_after_backward(self):
        if self._reducer is not None:
            self._reducer.finish()

    @property
    def dataset(self):
        return MinkowskiDataset

    @property
    def dataloader_options(self):
        return {"drop_last": True, "batch_size": self.options.batch_size, "num_workers": self.options.num_dataloader_workers,
                "collate_fn": MinkowskiCollection()}

    # ---- forward ---------------------------------------------------------------------------------------------------
    def forward(self, features: Tensor, extra: Tensor, event_coords: Tensor, event_values: Tensor, event_mask: Tensor,
                prong_coords: Tensor, prong_values: Tensor, prong_mask: Tensor, counts=None) -> Tuple[Tensor, Tensor]:
        """-> (event_logits [B, Ce], prong_logits [B, P, Cp]) (reference :90-116).  Inputs are borrowed, never mutated."""
        dev = event_values.device
        if torch.is_tensor(self.mean) and features.numel() and not self.options.disable_smart_features:
            features = features.clone()
            features[prong_mask] = (features[prong_mask] - self.mean) / self.std
            extra = (extra - self.extra_mean) / self.extra_std
        shape = self.training_dataset.pixel_shape
        self.network.hip_runtime().anchor_param = self._ddp_anchor       # the autograd anchor of the fused step (see above)
        event_pixels = self.preprocess_pixels(event_coords, event_values, shape)
        prong_pixels = self.preprocess_pixels(prong_coords, prong_values, shape)
        return self.network(features, extra, event_pixels, event_mask, prong_pixels, prong_mask, counts)

    def shared_step(self, batch):
        (features, extra, ev_c, ev_v, ev_m, pr_c, pr_v, pr_m, ev_t, pr_t) = batch[:10]
        counts = batch[10] if len(batch) > 10 else None               # optional host-side (max_prongs, n_prongs): avoids syncs
        if counts is not None:
            width, n_prongs = int(counts[0]), int(counts[1])
        else:
            per_event = pr_m.sum(1)
            width, n_prongs = int(per_event.max()), int(per_event.sum())
        features = features[:, :width].contiguous()
        pr_m = pr_m[:, :width].contiguous()
        pr_t = pr_t[:, :width].contiguous()
        return (ev_t, pr_t, *self.forward(features, extra, ev_c, ev_v, ev_m, pr_c, pr_v, pr_m, (features.shape[0], n_prongs)))

    def loss(self, logits: Tensor, targets: Tensor) -> Tensor:
        """Softmax focal loss mean_i(-log p_t (1 - p_t)^gamma), cross-entropy for gamma == 0 (reference :148-160) for one
        logit matrix, on the HIP focal kernel."""
        rt = self.network.hip_runtime()
        from transformercvn.hip.engine import focal_rows
        return focal_rows(logits, targets, float(self.gamma))

    def losses(self, event_logits, prong_logits, event_targets, prong_targets):
        """(total, event_loss, prong_loss, event_accuracy, prong_accuracy) from the fused loss kernel (:162-183)."""
        return self.network.hip_runtime().loss(event_logits, prong_logits, event_targets, prong_targets)

    def training_step(self, batch, batch_idx):
        event_targets, prong_targets, event_logits, prong_logits = self.shared_step(batch)
        total, event_loss, prong_loss, event_acc, prong_acc = self.losses(event_logits, prong_logits, event_targets, prong_targets)
        self.log("prong_loss", prong_loss)
        self.log("event_loss", event_loss)
        self.log("train_loss", total)
        self.log("train_event_accuracy", event_acc)
        self.log("train_prong_accuracy", prong_acc)
        return total

    def validation_step(self, batch, batch_idx):
        event_targets, prong_targets, event_logits, prong_logits = self.shared_step(batch)
        valid = prong_targets.to(prong_logits.device) >= 0
        prong_probs = torch.softmax(prong_logits[valid], dim=-1)
        event_probs = torch.softmax(event_logits, dim=-1)
        pt = prong_targets.to(prong_logits.device)[valid].long()
        et = event_targets.to(event_logits.device)
        self.prong_accuracy.update(prong_probs, pt)
        self.event_accuracy.update(event_probs, et)
        self.prong_auc.update(prong_probs, pt)
        self.event_auc.update(event_probs, et)

    def validation_epoch_end(self, outputs) -> None:
        ea, pa = self.event_accuracy.compute(), self.prong_accuracy.compute()
        eu, pu = self.event_auc.compute(), self.prong_auc.compute()
        for name, value in (("val_epoch_accuracy", (pa + ea) / 2), ("event_epoch_accuracy", ea), ("prong_epoch_accuracy", pa),
                            ("val_epoch_AUC", (eu + pu) / 2), ("event_epoch_AUC", eu), ("prong_epoch_AUC", pu)):
            self.log(name, value, sync_dist=True)
        for m in (self.event_accuracy, self.prong_accuracy, self.event_auc, self.prong_auc):
            m.reset()

    # Lightning hook: zero the flat gradient arena with one memset instead of per-parameter set_to_none
    def optimizer_zero_grad(self, *args, **kwargs):
        self.network.hip_runtime().zero_grad()
