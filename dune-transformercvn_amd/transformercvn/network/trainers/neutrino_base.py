"""Training harness base class (reference: transformercvn/network/trainers/neutrino_base.py:13-164).

Subclasses ``pytorch_lightning.LightningModule`` when Lightning is installed, otherwise a minimal stand-in with the same
hooks (``log`` etc.) so the module -- and `fit_loop` in this package -- work without it.
"""
from __future__ import annotations

import torch
from torch.utils.data import DataLoader

from transformercvn.options import Options
from transformercvn.network.networks.learning_rate_schedules import (get_linear_schedule_with_warmup,
                                                                    get_cosine_with_hard_restarts_schedule_with_warmup)

try:                                                                     # pragma: no cover - depends on the environment
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except ImportError:
    class _Base(torch.nn.Module):
        """Hook-compatible stand-in for pl.LightningModule."""

        def __init__(self):
            super().__init__()
            self.logged = {}

        def log(self, name, value, **kwargs):
            self.logged[name] = value

SYNTHETIC_PREFIX = "synthetic"


class NeutrinoBase(_Base):
    def __init__(self, options: Options):
        super().__init__()
        self.options = options
        self.training_dataset, self.validation_dataset, self.testing_dataset = self.create_datasets()
        self.mean, self.std, self.extra_mean, self.extra_std, self.pixel_mean, self.pixel_std = 0, 1, 0, 1, 0, 1
        if self.options.normalize_features:                      # statistics become non-trainable parameters (:37-45)
            stats = self.training_dataset.compute_statistics()
            self.mean, self.std, self.extra_mean, self.extra_std = (torch.nn.Parameter(torch.as_tensor(s), requires_grad=False)
                                                                    for s in stats[:4])
            if self.training_dataset.pixels is not None:
                self.pixel_mean = torch.nn.Parameter(stats[4], requires_grad=False)
                self.pixel_std = torch.nn.Parameter(stats[5], requires_grad=False)
        per_step = self.options.batch_size * max(1, self.options.num_gpu)
        self.steps_per_epoch = len(self.training_dataset) // per_step
        self.total_steps = self.steps_per_epoch * self.options.epochs
        self.warmup_steps = int(round(self.steps_per_epoch * self.options.learning_rate_warmup_epochs))

    # ---- data ----------------------------------------------------------------------------------------------------
    @property
    def dataset(self):
        raise NotImplementedError()

    @property
    def dataloader(self):
        return DataLoader

    @property
    def dataloader_options(self):
        return {"drop_last": True, "batch_size": self.options.batch_size, "pin_memory": self.options.num_gpu > 0,
                "num_workers": self.options.num_dataloader_workers}

    def _open(self, path, *limit):
        return self.dataset(path, *limit, event_current_targets=self.options.event_current_targets,
                            load_full_dataset=self.options.load_full_dataset)

    def create_datasets(self):
        """Train/validation split by fraction of one file, or separate files (:68-86).  A training_file of the form
        ``synthetic[:N[:P]]`` (or an empty one) selects the seeded synthetic dataset instead of an HDF5 file."""
        o = self.options
        if not o.training_file or str(o.training_file).startswith(SYNTHETIC_PREFIX):
            from transformercvn.dataset.minkowski_dataset import SyntheticDataset
            parts = str(o.training_file).split(":")
            n = int(parts[1]) if len(parts) > 1 and parts[1] else 1024
            p = int(parts[2]) if len(parts) > 2 and parts[2] else 8
            return SyntheticDataset(n, p, seed=1234), SyntheticDataset(max(n // 8, 8), p, seed=4321), None
        if len(o.validation_file) > 0:
            train, val = self._open(o.training_file), self._open(o.validation_file)
        else:
            cut = o.dataset_limit * o.train_validation_split
            train, val = self._open(o.training_file, (0.0, cut)), self._open(o.training_file, (cut, o.dataset_limit))
        test = self._open(o.testing_file) if len(o.testing_file) > 0 else None
        return train, val, test

    def train_dataloader(self) -> DataLoader:
        return self.dataloader(self.training_dataset, shuffle=True, **self.dataloader_options)

    def val_dataloader(self) -> DataLoader:
        return self.dataloader(self.validation_dataset, **self.dataloader_options)

    def test_dataloader(self) -> DataLoader:
        if self.testing_dataset is None:
            raise ValueError("Testing dataset not provided.")
        return self.dataloader(self.testing_dataset, **self.dataloader_options)

    # ---- optimisation ----------------------------------------------------------------------------------------------
    def configure_gradient_clipping(self, optimizer, *args, **kwargs):
        """Lightning hook (train.py:140 passes gradient_clip_val to the Trainer): with the fused optimizer the global-norm clip is one
        reduction over the gradient arena inside FlatAdamW.step (same `gradient_clip` value), so Lightning's 782-tensor
        clip_grad_norm_ in front of it is skipped; any other optimizer keeps Lightning's clipping."""
        flat = getattr(self, "_flat_optimizer", None)
        if flat is not None and optimizer is flat and flat.clip > 0:
            return
        parent = getattr(super(), "configure_gradient_clipping", None)
        if parent is not None:
            return parent(optimizer, *args, **kwargs)

    def configure_optimizers(self):
        """AdamW-style optimizer with two parameter groups and a per-step LambdaLR (:88-152).  The no-decay group is
        selected by the substrings 'bias' / 'LayerNorm.weight' in the parameter name -- as in the reference the second
        string never matches torch's norm names, so norm and PReLU weights are decayed."""
        o = self.options
        if hasattr(self, "adopt_trainer_precision"):
            self.adopt_trainer_precision()                   # before the optimizer binds the parameter arena
        opt_cls = None
        if "apex" in o.optimizer:
            try:
                import apex.optimizers as ao
                opt_cls = {"apex_adam": ao.FusedAdam, "apex_lamb": ao.FusedLAMB}.get(o.optimizer, ao.FusedSGD)
            except ImportError:
                pass
        else:
            opt_cls = getattr(torch.optim, o.optimizer)
        if opt_cls is None:
            print(f"Unable to load desired optimizer: {o.optimizer}.\nUsing pytorch AdamW as a default.")
            opt_cls = torch.optim.AdamW
        no_decay = ("bias", "LayerNorm.weight")
        named = [(n, p) for n, p in self.named_parameters() if n != "_ddp_anchor"]     # the DDP anchor is not a model parameter
        groups = [{"params": [p for n, p in named if not any(s in n for s in no_decay)], "weight_decay": o.l2_penalty},
                  {"params": [p for n, p in named if any(s in n for s in no_decay)], "weight_decay": 0.0}]
        optimizer = None
        rt = self.network.hip_runtime() if hasattr(self.network, "hip_runtime") else None
        first = next(self.parameters())
        if opt_cls is torch.optim.AdamW and rt is not None and first.is_cuda and getattr(o, "hip_fused_optimizer", True):
            # one fused launch over the flat arenas instead of 782 tensor updates.  Parameters that never get a
            # gradient in the reference (grad None -> skipped by AdamW, no decay either) are frozen here as well.
            from transformercvn.hip.optimizer import FlatAdamW
            names = {id(p): n[len("network."):] for n, p in named if n.startswith("network.")}
            frozen = ["prong_position_embedding"] + (["feature_embedding."] if o.disable_smart_features else [])
            optimizer = FlatAdamW(groups, rt, names, lr=o.learning_rate, clip=o.gradient_clip, frozen=frozen)
            self._flat_optimizer = optimizer           # enable_data_parallel() broadcasts its moments from rank 0 (resume)
        if optimizer is None:
            optimizer = opt_cls(groups, lr=o.learning_rate)
        if o.learning_rate_cycles < 1:
            sched = get_linear_schedule_with_warmup(optimizer, self.warmup_steps, self.total_steps)
        else:
            sched = get_cosine_with_hard_restarts_schedule_with_warmup(optimizer, self.warmup_steps, self.total_steps,
                                                                       o.learning_rate_cycles)
        return [optimizer], [{"scheduler": sched, "interval": "step", "frequency": 1}]
