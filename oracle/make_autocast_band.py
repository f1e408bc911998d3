"""The REFERENCE's own bf16 behaviour on the golden cases (run in the build container only, like make_golden.py).

    python oracle/make_autocast_band.py        # writes tests/golden/autocast_bf16_band.npz

The real reference module (imported from /root/reference through make_golden.build_reference) is run under
``torch.autocast("cpu", dtype=torch.bfloat16)`` -- what ``train.py -fp16`` selects up to the half type (reference train.py:141,172) --
on the inputs and weights of every golden case, eval and train mode, forward and (train) backward.  Stored per case: its logits, its
max-norm relative logit error against the case's own fp32 golden, and the cosine of its sentinel gradients with the fp32 golden
gradients.  The GPU tests gate the bf16 throughput mode of this repo against that band: a bf16 mode cannot be asked to sit closer to
the fp32 reference than the reference's own bf16 mode does on identical inputs.  Data only; nothing of the reference travels."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import make_golden as MG          # noqa: E402
from oracle import tcvn_oracle as O           # noqa: E402
from golden_utils import load_case, rel_err, train_cfg  # noqa: E402

CASES = ["small_b3", "tutorial_b2p4", "tutorial_ragged", "tutorial_b2p8", "tutorial_b2p12", "tutorial_b32p8"]


def main():
    torch.set_num_threads(8)
    out = {}
    for name in CASES:
        cfg, over, batch, g = load_case(name)
        ref = MG.build_reference(cfg)
        ref.load_state_dict(O.fill_state(cfg, int(g["weight_seed"])), strict=True)
        ref.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, ev, pr = ref.shared_step(batch)
        out[f"{name}:eval_event_logits"], out[f"{name}:eval_prong_logits"] = ev.float().numpy(), pr.float().numpy()
        e = (rel_err(ev.float(), g["eval_event_logits"]), rel_err(pr.float(), g["eval_prong_logits"]))
        cfgt = train_cfg(over)
        reft = MG.build_reference(cfgt)
        reft.load_state_dict(O.fill_state(cfgt, int(g["weight_seed"])), strict=True)
        reft.train()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            loss = reft.training_step(batch, 0)
        loss.backward()
        named = dict(reft.named_parameters())
        for k in [k for k in g if k.startswith("grad:")]:
            mine, r = named[k[5:]].grad.float().numpy().ravel(), g[k].ravel()
            out[f"{name}:gradcos:{k[5:]}"] = np.array(float((mine * r).sum() / (np.linalg.norm(mine) * np.linalg.norm(r) + 1e-30)))
        out[f"{name}:train_total_loss"] = np.array(float(loss))
        reft.zero_grad()
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            _, _, ev, pr = reft.shared_step(batch)
        out[f"{name}:train_event_logits"], out[f"{name}:train_prong_logits"] = ev.float().numpy(), pr.float().numpy()
        t = (rel_err(ev.float(), g["train_event_logits"]), rel_err(pr.float(), g["train_prong_logits"]))
        out[f"{name}:logit_err"] = np.array([e[0], e[1], t[0], t[1]])
        print(f"reference under bf16 autocast, {name}: eval event {e[0]:.3e} prong {e[1]:.3e}; train event {t[0]:.3e} prong {t[1]:.3e}; "
              f"train loss {float(loss):.5f} (fp32 golden {float(g['train_total_loss']):.5f})", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "autocast_bf16_band.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
