"""validation_step / validation_epoch_end (reference: trainers/neutrino_full_base_trainer.py:194-230) on the
GPU against accuracy / one-vs-rest AUROC computed with scikit-learn from the CPU oracle's probabilities."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case
from model_utils import build_trainer, to_device

pytestmark = pytest.mark.gpu


def _auc(t, p):
    from sklearn.metrics import roc_auc_score
    cs = [c for c in range(p.shape[1]) if 0 < (t == c).sum() < len(t)]
    return float(np.mean([roc_auc_score(t == c, p[:, c]) for c in cs]))


def test_validation_epoch_metrics_match_sklearn_on_oracle_probabilities():
    cfg, over, _, g = load_case("small_b3")
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    model = build_trainer(cfg, sd)
    model.eval()
    batches = [O.synthetic_batch([3, 1, 4, 2, 2, 3], 100 + i, cfg) for i in range(2)]
    ev_p, ev_t, pr_p, pr_t = [], [], [], []
    with torch.no_grad():
        for i, b in enumerate(batches):
            model.validation_step(to_device(b), i)
            et, pt, ev, pr, _ = O.shared_step(sd, cfg, b, training=False)
            valid = pt >= 0
            ev_p.append(torch.softmax(ev, -1).numpy()); ev_t.append(et.numpy())
            pr_p.append(torch.softmax(pr[valid], -1).numpy()); pr_t.append(pt[valid].long().numpy())
    model.validation_epoch_end(None)
    EP, ET, PP, PT = np.concatenate(ev_p), np.concatenate(ev_t), np.concatenate(pr_p), np.concatenate(pr_t)
    ea, pa = float((EP.argmax(1) == ET).mean()), float((PP.argmax(1) == PT).mean())
    eu, pu = _auc(ET, EP), _auc(PT, PP)
    got = {k: float(v) for k, v in model.logged.items()}
    print(got)
    assert abs(got["event_epoch_accuracy"] - ea) < 1e-6 and abs(got["prong_epoch_accuracy"] - pa) < 1e-6
    assert abs(got["val_epoch_accuracy"] - (ea + pa) / 2) < 1e-6
    assert abs(got["event_epoch_AUC"] - eu) < 1e-5 and abs(got["prong_epoch_AUC"] - pu) < 1e-5
    assert abs(got["val_epoch_AUC"] - (eu + pu) / 2) < 1e-5
    # metrics are reset for the next epoch
    model.validation_step(to_device(batches[0]), 0)
    model.validation_epoch_end(None)
    assert abs(float(model.logged["event_epoch_accuracy"]) - float((ev_p[0].argmax(1) == ev_t[0]).mean())) < 1e-6
