"""SURVEY.md 8f-3 on the GPU: a module scripted with the registered operator (network.prepare_export(use_ops=True)) and moved to the
MI355X dispatches its embedders to tcvn::densenet_embed's HIP kernel (libtcvn_hip.so) -- checked by the launches the library's
profiler records while the SCRIPTED module runs -- and reproduces the reference's golden eval logits; the same file run on CPU
tensors takes the operator's ATen kernel."""
import io

import pytest
import torch

from oracle import tcvn_oracle as O
from golden_utils import load_case, rel_err
from model_utils import build_trainer
from test_export_cpu import _dense_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,precision,gate", [("small_b3", "fp32", 1e-4), ("tutorial_b2p4", "bf16", 2e-2)])
def test_scripted_module_reaches_the_hip_kernels(name, precision, gate):
    from transformercvn.hip import _lib
    cfg, over, batch, g = load_case(name)
    model = build_trainer(cfg, O.fill_state(cfg, int(g["weight_seed"])), precision=precision)
    model.eval()
    scripted = torch.jit.script(model.network.prepare_export(use_ops=True))
    buf = io.BytesIO()
    torch.jit.save(scripted, buf)
    buf.seek(0)
    loaded = torch.jit.load(buf, map_location="cuda")
    inputs = tuple(t.cuda() for t in _dense_inputs(cfg, batch))
    _lib.lib.tcvn_profile_filter(None); _lib.lib.tcvn_profile_reset(); _lib.lib.tcvn_profile_enable(1)
    with torch.no_grad():
        ev, pr = loaded(*inputs)
    torch.cuda.synchronize()
    _lib.lib.tcvn_profile_enable(0)
    launches = [r[0] for r in _lib.profile_records()]
    _lib.lib.tcvn_profile_reset()
    assert any("conv" in k or "gemm" in k or "stem" in k for k in launches), launches[:5]       # the scripted graph ran libtcvn_hip.so kernels
    e1, e2 = rel_err(ev.cpu(), g["eval_event_logits"]), rel_err(pr.cpu(), g["eval_prong_logits"])
    print(name, precision, "scripted module on the GPU:", len(launches), "HIP convolution-class launches; logit error", e1, e2)
    assert e1 < gate and e2 < gate
    # the same scripted file on CPU tensors: the operator's ATen kernel
    cpu = torch.jit.load(io.BytesIO(buf.getvalue()), map_location="cpu")
    with torch.no_grad():
        ev_c, pr_c = cpu(*_dense_inputs(cfg, batch))
    assert rel_err(ev_c, g["eval_event_logits"]) < 1e-4 and rel_err(pr_c, g["eval_prong_logits"]) < 1e-4
