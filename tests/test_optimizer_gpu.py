"""Fused flat AdamW + gradient clip against what the reference runs: torch.optim.AdamW with the two
parameter groups of NeutrinoBase.configure_optimizers (trainers/neutrino_base.py:88-152) after torch.nn.utils.clip_grad_norm_
(Lightning's gradient_clip_val, train.py:140)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def test_adamw_kernel_matches_torch_adamw_with_clip():
    from transformercvn.hip._lib import lib, check
    gen = torch.Generator().manual_seed(3)
    sizes = [70001, 4096, 333]                 # decayed group, no-decay group, never-touched parameter
    decays = [0.0213, 0.0, -1.0]
    ps = [torch.randn(s, generator=gen).cuda() for s in sizes]
    flat_p = torch.cat(ps).clone()
    ref = [p.clone().requires_grad_(True) for p in ps[:2]]
    frozen_before = ps[2].clone()
    opt = torch.optim.AdamW([{"params": [ref[0]], "weight_decay": decays[0]}, {"params": [ref[1]], "weight_decay": 0.0}], lr=3e-3)
    n = flat_p.numel()
    wd = torch.cat([torch.full((s,), d) for s, d in zip(sizes, decays)]).cuda()
    m, v = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    partials, ss = torch.zeros(1024, dtype=torch.float64, device="cuda"), torch.zeros(1, device="cuda")
    clip = 5.0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for step in range(1, 6):
        scale = 10.0 if step % 2 else 0.01       # one clipped and one unclipped regime
        gs = [torch.randn(s, generator=gen).cuda() * scale for s in sizes]
        flat_g = torch.cat(gs).contiguous()
        # the reference clips over the gradients that exist (the frozen parameter has none)
        flat_g[sizes[0] + sizes[1]:] = 0
        for r, g in zip(ref, gs):
            r.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, clip)
        opt.step()
        check(lib.tcvn_grad_sumsq(_ptr(flat_g), n, _ptr(partials), 1024, _ptr(ss), st), "sumsq")
        check(lib.tcvn_adamw_step(_ptr(flat_p), _ptr(flat_g), _ptr(m), _ptr(v), _ptr(wd), n, 3e-3, 0.9, 0.999, 1e-8, step,
                                  _ptr(ss), clip, st), "adamw")
        torch.cuda.synchronize()
        assert abs(ss.sqrt().item() - norm.item()) <= 1e-5 * norm.item()
        off = 0
        for r, s in zip(ref, sizes[:2]):
            mine = flat_p[off:off + s]
            err = (mine - r.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, r.detach().abs().max().item()), (step, err)
            off += s
    assert torch.equal(flat_p[sizes[0] + sizes[1]:], frozen_before)


def test_trainer_builds_the_fused_optimizer_and_steps():
    from model_utils import build_trainer
    from golden_utils import load_case
    cfg, over, batch, g = load_case("small_b3")
    trainer = build_trainer(cfg, None, "fp32")
    trainer.train()
    (opt,), (sched,) = trainer.configure_optimizers()
    from transformercvn.hip.optimizer import FlatAdamW
    assert isinstance(opt, FlatAdamW) and len(opt.param_groups) == 2
    rt = trainer.network.hip_runtime()
    before = rt.flat_param.clone()
    dead = dict(trainer.network.named_parameters())["prong_embedding.prong_position_embedding"].detach().clone()
    rt.zero_grad()
    rt.flat_grad.normal_(generator=torch.Generator(device="cuda").manual_seed(1))
    for grp in opt.param_groups:                           # the warm-up schedule starts at lr = 0
        grp["lr"] = 1e-3
    opt.step()
    sched["scheduler"].step()
    torch.cuda.synchronize()
    changed = (rt.flat_param != before).float().mean().item()
    assert changed > 0.9                                   # everything with a gradient moved ...
    assert torch.equal(dict(trainer.network.named_parameters())["prong_embedding.prong_position_embedding"].detach(), dead)
    assert torch.isfinite(rt.flat_param).all()
    sd = opt.state_dict()
    assert sd["flat"]["step"] == 1 and sd["flat"]["exp_avg"].numel() == rt.flat_param.numel()


def test_amp_grad_scaler_flow_on_the_arena_views():
    """`train.py -fp16` runs the step under Lightning's AMP plugin: autocast around training_step, the loss scaled by a GradScaler,
    unscale_ + inf check on every parameter's .grad, then optimizer.step().  Gradients here live in one arena and never pass through
    autograd: the scaled loss gradient must scale the arena linearly, unscale_ must bring it back, and the scaler must step the
    fused optimizer."""
    from model_utils import build_trainer, to_device
    from golden_utils import load_case, train_cfg
    from oracle import tcvn_oracle as O
    cfg, over, batch, g = load_case("small_b3")
    cfg = train_cfg(over)
    sd = O.fill_state(cfg, int(g["weight_seed"]))
    dbatch = to_device(batch)
    ref = build_trainer(cfg, sd, "bf16")
    ref.train()
    rt0 = ref.network.hip_runtime()
    rt0.zero_grad()
    ref.training_step(dbatch, 0).backward()
    torch.cuda.synchronize()
    plain = rt0.flat_grad.clone()

    model = build_trainer(cfg, sd, "bf16")
    model.train()
    (opt,), _ = model.configure_optimizers()
    for grp in opt.param_groups:
        grp["lr"] = 1e-3
    rt = model.network.hip_runtime()
    scaler = torch.cuda.amp.GradScaler(init_scale=1024.0)
    rt.zero_grad()
    with torch.autocast("cuda", dtype=torch.float16):
        loss = model.training_step(dbatch, 0)
    scaler.scale(loss).backward()
    torch.cuda.synchronize()
    e = ((rt.flat_grad / 1024.0 - plain).norm() / plain.norm()).item()
    print("scaled backward / 1024 vs plain backward: rel L2", e)
    assert e < 1e-5
    scaler.unscale_(opt)
    assert ((rt.flat_grad - plain).norm() / plain.norm()).item() < 1e-5
    before = rt.flat_param.clone()
    scaler.step(opt)
    scaler.update()
    torch.cuda.synchronize()
    assert (rt.flat_param != before).float().mean().item() > 0.9 and torch.isfinite(rt.flat_param).all()
