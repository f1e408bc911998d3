"""Run-to-run reproducibility of the bf16 DenseNet forward (train mode) on the GPU.

The forward has no atomics: every repetition of the same step must be bit-identical.  This is the regression test for a
hazard inside the hand-scheduled MFMA chain of the 3x3 tile kernel (a VALU copy landing directly in front of an inline-asm
MFMA), which corrupted about one wave-tile in 10^4 and only showed as a rare parity failure."""
import pytest
import torch

from golden_utils import load_case, train_cfg
from oracle import tcvn_oracle as O

pytestmark = pytest.mark.gpu


def test_bf16_forward_is_bit_reproducible():
    import test_densenet_gpu as T
    cfg, over, batch, g = load_case("tutorial_b2p4")
    cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    n_img = int(batch[7].sum())
    eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=False)
    coords, values = batch[5].cuda(), batch[6].cuda()
    out = torch.empty(n_img, eng.out_dim, device="cuda")
    names = [f"dense{i + 1}" for i in range(len(cfg.densenet_structure))]
    first = None
    for rep in range(60):
        eng.forward(coords, values, n_img, out, train=True, seed=1)
        cur = [eng.tap(n).clone().view(torch.int16) for n in names] + [out.clone().view(torch.int32)]
        if first is None:
            first = cur
            continue
        for n, a, b in zip(names + ["out"], cur, first):
            assert torch.equal(a, b), f"repetition {rep}: {n} differs in {(a != b).sum().item()} elements"
