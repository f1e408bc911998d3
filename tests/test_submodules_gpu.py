"""The reference's sub-module call surface (CreateCompiled.ipynb cells 7-8 call network.prong_embedding,
network.encoder, network.event_decoder, network.prong_decoder one after the other): each stage's own forward() runs the
matching C-ABI stage and must reproduce the reference's intermediate taps and logits stored in the goldens."""
import numpy as np
import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, train_cfg, rel_err, tap_sample
from model_utils import build_trainer, to_device

pytestmark = pytest.mark.gpu


# fp32 summation-order noise of the train-mode taps (BatchNorm over a handful of maps): measured 1.3e-5 .. 2.3e-5 on the tokens and
# 3.2e-5 on the hidden states depending on the kernel generation; eval-mode taps agree to 4e-6.  The gate on the logits is 1e-3.
def _close(t, g, key, tol=5e-5):
    stat, samp = tap_sample(t.cpu())
    ref_stat, ref_samp = g[key.replace("tap:", "tap_stat:")], g[key.replace("tap:", "tap_samp:")]
    scale = max(float(np.abs(ref_samp).max()), 1e-12)
    err = max(float(np.abs(samp - ref_samp).max()) / scale, float(np.abs(stat - ref_stat).max()) / max(float(np.abs(ref_stat).max()), 1e-12))
    print(key, "max-norm error vs reference tap", err)
    return err < tol


@pytest.mark.parametrize("name", ["small_b3", "tutorial_b2p4", "tutorial_ragged", "tutorial_b2p8", "tutorial_b2p12"])
@pytest.mark.parametrize("training", [False, True])
def test_stage_by_stage_forward_matches_reference(name, training):
    cfg, over, batch, g = load_case(name)
    if training:
        cfg = train_cfg(over)
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])))
    model.train(training)
    f, x, ec, ev, em, pc, pv, pm, et, pt = to_device(batch)
    width = int(pm.sum(1).max())
    f, pm = f[:, :width].contiguous(), pm[:, :width].contiguous()
    net = model.network
    shape = model.training_dataset.pixel_shape
    event_pixels = model.preprocess_pixels(ec, ev, shape)
    prong_pixels = model.preprocess_pixels(pc, pv, shape)
    kind = "traintap:" if training else "evaltap:"
    with torch.no_grad():
        tokens, mask = net.prong_embedding(f, x, event_pixels, em, prong_pixels, pm)
        assert tokens.shape == (f.shape[0], 1 + width, cfg.hidden_dim) and mask.shape == (f.shape[0], 1 + width)
        assert mask.dtype == torch.bool and bool(mask[:, 0].all())
        assert _close(tokens, g, kind + "tokens"), "tokens"
        hidden, padding_mask, sequence_mask = net.encoder(tokens, mask)
        assert hidden.shape == (1 + width, f.shape[0], cfg.hidden_dim)
        assert torch.equal(padding_mask, ~mask) and sequence_mask.shape == (1 + width, f.shape[0], 1)
        assert _close(hidden, g, kind + "hidden", 5e-5), "hidden"
        ev_logits = net.event_decoder(hidden[0])
        pr_logits = net.prong_decoder(hidden[1:]).transpose(0, 1)
    key = "train" if training else "eval"
    e1, e2 = rel_err(ev_logits.cpu(), g[key + "_event_logits"]), rel_err(pr_logits.cpu(), g[key + "_prong_logits"])
    print(name, training, "stage-by-stage logits vs reference", e1, e2)
    assert e1 < 1e-4 and e2 < 1e-4


def test_linear_block_and_dense_net_forward_alone():
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])))
    model.eval()
    pe = model.network.prong_embedding
    x = torch.randn(7, pe.combined_embedding.linear.in_features, device="cuda")
    y = pe.combined_embedding(x)
    lb = pe.combined_embedding
    z = torch.nn.functional.linear(x, lb.linear.weight)
    ref = torch.nn.functional.prelu(torch.nn.functional.batch_norm(z, lb.norm.running_mean, lb.norm.running_var, lb.norm.weight,
                                                                  lb.norm.bias, False, 0.1, 1e-5), lb.activation.weight)
    assert rel_err(y.cpu(), ref.detach().cpu()) < 1e-5
    dense = model.preprocess_pixels(batch[5].cuda(), batch[6].cuda(), model.training_dataset.pixel_shape).to_dense()
    out = pe.prong_pixel_embedding(dense)                         # dense NCHW map in, like the reference's DenseNet.forward
    stat, samp = tap_sample(out.cpu())
    pfx = "evaltap_samp:network.prong_embedding.prong_pixel_embedding:out"
    assert float(np.abs(samp - g[pfx]).max()) / float(np.abs(g[pfx]).max()) < 1e-4


@pytest.mark.parametrize("bn,prelu", [(False, True), (True, False), (False, False)])
def test_linear_block_option_variants_forward_alone(bn, prelu):
    """The holder modules' own forward for the LinearBlock option variants (reference layers/prong_feature_embedding.py:11-21,
    layers/encoder.py:10-24: linear_batch_norm / linear_prelu_activation False) on the HIP row kernels against the same modules through
    ATen on the CPU, eval mode."""
    cfg = O.tutorial_config(densenet_structure=[1, 1], densenet_growth_rate=8, initial_pixel_dim=16, num_encoder_layers=1, pixel_embedding_dim=64,
                            hidden_dim=64, num_prong_decoder_layers=3, dropout=0.1, pixel_noise_std=0.0, linear_batch_norm=bn,
                            linear_prelu_activation=prelu)
    model = build_trainer(cfg, O.fill_state(cfg, 5))
    model.eval()
    lb, dec = model.network.prong_embedding.combined_embedding, model.network.prong_decoder
    assert isinstance(lb.norm, torch.nn.BatchNorm1d) == bn and isinstance(lb.activation, torch.nn.PReLU) == prelu
    assert (lb.linear.bias is None) == bn
    x = torch.randn(9, lb.linear.in_features, device="cuda")
    h = torch.randn(3, 4, cfg.hidden_dim, device="cuda")
    y, z = lb(x), dec(h)
    import copy
    lb_c, dec_c = copy.deepcopy(lb).cpu().eval(), copy.deepcopy(dec).cpu().eval()
    with torch.no_grad():
        ref_y = lb_c.dropout(lb_c.activation(lb_c.norm(lb_c.linear(x.cpu()))))
        ref_z = dec_c.output_layer(dec_c.hidden_layers(h.cpu().reshape(12, -1))).reshape(3, 4, -1)
    assert rel_err(y.cpu(), ref_y) < 1e-5 and rel_err(z.cpu(), ref_z) < 1e-5


def test_stage_calls_on_cpu_fail_loudly():
    cfg, over, batch, g = load_case("small_b3")
    model = build_trainer(cfg, None, device=None)
    with pytest.raises(RuntimeError):
        model.network.event_decoder(torch.zeros(2, cfg.hidden_dim))


@pytest.mark.parametrize("case", ["tutorial_ragged", "tutorial_b2p8", "tutorial_b2p12", "tutorial_b2p4"])
@pytest.mark.parametrize("training", [False, True])
def test_fused_encoder_matches_layer_by_layer_kernels(training, case):
    """csrc/encoder_fused.hip (one launch, one workgroup per event) against the row kernels it replaces, same inputs, incl.
    dropout (same stateless masks) and a ragged key-padding mask: hidden states to fp32 summation-order level, and the saved
    tensors the backward reads (it runs on the unfused kernels either way) give the same parameter gradients."""
    from transformercvn.hip._lib import lib
    # ragged: prongs 1 / 16 / 5, S = 17 with padding (<16,11>); b2p8: S = 9 (<16,5>, BASELINE config 2); b2p12: S = 13 (<16,8>);
    # b2p4: S = 5 (<16,3>, 2-layer encoder)
    cfg, over, batch, g = load_case(case)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    res = {}
    for fused in (1, 0):
        model = build_trainer(cfg, sd)
        model.train(training)
        rt = model.network.hip_runtime()
        rt.ensure_bound()
        lib.tcvn_head_set_fused_encoder(rt.head.handle, fused)
        rt.zero_grad()
        if training:
            loss = model.training_step(to_device(batch), 0)
            loss.backward()
            torch.cuda.synchronize()
            res[fused] = (loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if "network.encoder." in k})
        else:
            with torch.no_grad():
                _, _, ev, pr = model.shared_step(to_device(batch))
            res[fused] = (ev.clone(), pr.clone())
    if training:
        assert abs(res[1][0] - res[0][0]) < 2e-6 * abs(res[0][0]), (res[1][0], res[0][0])
        scale = max(v.norm().item() for v in res[0][1].values())
        errs = sorted((((res[1][1][k] - res[0][1][k]).norm() / (res[0][1][k].norm() + 1e-4 * scale)).item(), k) for k in res[0][1])
        print("largest differences:", errs[-3:])
        worst = errs[-1][0]
        print("fused vs unfused encoder, train (dropout 0.1): loss", res[1][0], res[0][0], "worst grad rel L2", worst)
        assert worst < 2e-4
    else:
        e = max(rel_err(res[1][0].cpu(), res[0][0].cpu()), rel_err(res[1][1].cpu(), res[0][1].cpu()))
        print("fused vs unfused encoder, eval: logit rel err", e)
        assert e < 1e-5
