"""Validation metrics: torchmetrics when installed (reference: neutrino_full_base_trainer.py:70-74), otherwise small
accumulators with the same update/compute/reset surface (AUROC via scikit-learn, one-vs-rest macro average)."""
from __future__ import annotations

import torch


class _Accuracy:
    def __init__(self):
        self.reset()

    def reset(self):
        self.correct, self.total = 0, 0

    def update(self, probs, target):
        self.correct += int((probs.argmax(1) == target).sum())
        self.total += int(target.numel())

    def compute(self):
        return torch.tensor(self.correct / max(self.total, 1))


class _Auroc:
    def __init__(self, num_classes):
        self.num_classes = num_classes
        self.reset()

    def reset(self):
        self.p, self.t = [], []

    def update(self, probs, target):
        self.p.append(probs.detach().float().cpu())
        self.t.append(target.detach().cpu())

    def compute(self):
        from sklearn.metrics import roc_auc_score
        p, t = torch.cat(self.p).numpy(), torch.cat(self.t).numpy()
        scores = [roc_auc_score(t == c, p[:, c]) for c in range(self.num_classes) if 0 < (t == c).sum() < len(t)]
        return torch.tensor(float(sum(scores) / max(len(scores), 1)))


def make_metrics(num_event_classes: int, num_prong_classes: int):
    try:                                                                # pragma: no cover
        from torchmetrics import Accuracy, AUROC
        return (Accuracy(task="multiclass", num_classes=num_event_classes), Accuracy(task="multiclass", num_classes=num_prong_classes),
                AUROC(task="multiclass", num_classes=num_event_classes), AUROC(task="multiclass", num_classes=num_prong_classes))
    except ImportError:
        return _Accuracy(), _Accuracy(), _Auroc(num_event_classes), _Auroc(num_prong_classes)
