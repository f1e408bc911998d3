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


def test_bf16_dense_layer_weight_gradients_are_bit_reproducible():
    """Round 5: the per-workgroup weight-gradient slabs are reduced in a fixed order (one y-slice per job, no fp32 atomics between slices:
    csrc/elementwise_bwd.hip, slab_reduce_body), and the forward statistics are integer sums (csrc/bn_lf.h): repeating the same train
    step must reproduce the dense layers' convolution weight gradients bit for bit.  (Not claimed for the stem's conv0 gradient and the
    exact-zero convolution biases: those still end in LDS / global fp32 atomics.)"""
    import test_densenet_gpu as T
    cfg, over, batch, g = load_case("tutorial_b2p4")
    cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    n_img = int(batch[7].sum())
    eng, data, grads = T._engine(cfg, sd, mode=1, with_grad=True)
    coords, values = batch[5].cuda(), batch[6].cuda()
    out = torch.empty(n_img, eng.out_dim, device="cuda")
    d_out = torch.randn(n_img, eng.out_dim, generator=torch.Generator().manual_seed(3)).cuda()
    keys = [k for k in grads if ".dense" in k and k.endswith(("conv1.weight", "conv2.weight"))]
    assert len(keys) == 2 * sum(cfg.densenet_structure)
    first = None
    for rep in range(6):
        for v in grads.values():
            v.zero_()
        eng.forward(coords, values, n_img, out, train=True, seed=1)
        eng.backward(d_out)
        torch.cuda.synchronize()
        cur = {k: grads[k].clone().view(torch.int32) for k in keys}
        assert all(torch.isfinite(grads[k]).all() and grads[k].abs().max() > 0 for k in keys)
        if first is None:
            first = cur
            continue
        diff = [k for k in keys if not torch.equal(cur[k], first[k])]
        assert not diff, f"repetition {rep}: {len(diff)} of {len(keys)} weight gradients differ, e.g. {diff[:3]}"
